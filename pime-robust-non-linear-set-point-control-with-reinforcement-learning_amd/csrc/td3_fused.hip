// One TD3 optimizer step as FOUR launches: critic gradients, critic apply, actor gradients, actor apply.
//
// replaces (reference, /root/reference/elegantrl/agent.py): AgentTD3.update_net's loop body (:314-331) -- get_obj_critic_raw
// (:361-370: minibatch gather, target actor + clamped smoothing noise (net.py:107-110), twin target heads + min, online twin
// forward, SmoothL1 x 2), obj_critic.backward(), cri_optimizer.step(), the delayed soft update of cri_target (:116-124),
// obj_actor = -cri_target(state, act(state)).mean() (:323-324; the reference differentiates through the TARGET critic's first head),
// obj_actor.backward(), act_optimizer.step(), the delayed soft update of act_target -- ~150 PyTorch / rocBLAS launches per
// optimizer step in round 3 (0.76 ms, 0.016 of the f32 matrix peak at batch 4 096).
//
// Shape of the problem: batch 4 096, nets of width 128 -- 0.48 MFLOP per sample, 2 GFLOP per step.  A sample tile per wave for the
// whole net (the PPO kernels' decomposition) would occupy 128 of 1 024 SIMDs.  Here a WORKGROUP owns one 16-sample tile and its
// NW waves (8 at width 128: two per SIMD; 4 at width 64) split every layer's OUTPUT features (v_mfma_f32_16x16x4_f32; wave w computes
// output tiles [PER w, PER w + PER) of md / 16, PER = md / 16 / NW): batch 4 096 = 256 workgroups = every SIMD of the chip busy.  Consequences:
//   * a weight element is used by exactly ONE wave of a workgroup, once: weights go global -> registers (16-byte row pieces of the
//     nn.Linear tensors themselves: a lane's four consecutive k values of an output row are one global_load_dwordx4), a layer ahead
//     of their use.  No packed images, no LDS staging, nothing to re-pack after an optimizer step.
//   * the layer's activation meets in LDS ("chain layout": tile t, lane group q, sample j, 4 registers = features 16 t + 4 q + r;
//     lane groups 72 floats apart): written as the accumulators stand (one ds_write_b128 per tile), read back by every wave as the
//     next layer's B operands (one ds_read_b128 per tile), and read again -- the same image, conflict-free ds_read_b32 -- as BOTH
//     operands of the weight gradients dW = dZ^T H, which contract over the tile's 16 samples (4 k-steps per 16 x 16 block).
//   * dX = W^T dZ reads W by columns (four dword loads where the forward has one 16-byte load).
//   * every workgroup leaves one partial gradient (slab, block-major in accumulator order); td3_apply_kernel sums the slabs in slab
//     order (bit-reproducible), applies torch.optim.Adam to the element it has just reduced and, on delayed steps, the soft target
//     update -- the parameters are the only copy of the weights, so that is all an optimizer step has to write.
// The minibatch rows come from an index table [steps][B]; the row of a step is a launch argument (a captured graph of an update's
// steps bakes each row into its nodes: no launch advances shared state); the smoothing noise comes from a table of normals (parity
// tests inject the reference's draws) or from Philox stream 3 in the kernel.
#include <cstdlib>
#include "td3.hpp"
#include "pime_common.hpp"

namespace pime {

typedef float f32x4_t __attribute__((ext_vector_type(4)));

constexpr int kTd3Tile = 16;
constexpr int kTd3DefaultWaves = 8;   // measured: 63.1 -> 60.4 us per optimizer step (profiles/r04_u_td3_waves_ab.txt)
constexpr int kQP = 72, kTP = 4 * kQP;   // chain layout: floats between lane groups / tiles (72 = 16 samples x 4 + 8: the operand reads of the weight gradients hit 32 banks; pitches 68 .. 88 swept at the end of round 4: the launches take the same 20.0 / 21.7 us)
constexpr uint32_t STREAM_TD3_SMOOTH = 3;

#define TD3_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#define TD3_NO_HOIST() asm volatile("" ::: "memory")
// PIME_TD3_TRACE=1: 100 MHz wall-clock marks of workgroup 0 (tuning aid; the pointer is NULL in production)
#define TD3_MARK(i)                                                                        \
    do {                                                                                   \
        if (a.trace && blockIdx.x == 0 && threadIdx.x == 0) a.trace[i] = wall_clock64();   \
    } while (0)

__device__ __forceinline__ f32x4_t mfma16(float a, float b, f32x4_t c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x4_t ld4(const float* p) { return *reinterpret_cast<const f32x4_t*>(p); }
__device__ __forceinline__ void st4(float* p, const f32x4_t& v) { *reinterpret_cast<f32x4_t*>(p) = v; }
__device__ __forceinline__ f32x4_t relu4(f32x4_t v) {   // one v_med3_f32 per element (`v > 0 ? v : 0` compiles to a canonicalising max + a max)
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = __builtin_amdgcn_fmed3f(v[r], 0.f, __builtin_inff());
    return v;
}
// d * [h > 0] (torch's threshold_backward)
__device__ __forceinline__ f32x4_t gate4(f32x4_t d, const f32x4_t& h) {
#pragma unroll
    for (int r = 0; r < 4; ++r) d[r] = h[r] > 0.f ? d[r] : 0.f;
    return d;
}
template <int CTRL>
__device__ __forceinline__ float dpp_add16(float v) {
    return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
// sum over the 16 lanes of a row (= the tile's 16 samples), result in every lane of the row; fixed order
__device__ __forceinline__ float row_sum16(float v) {
    v = dpp_add16<0xb1>(v);    // quad_perm [1,0,3,2]
    v = dpp_add16<0x4e>(v);    // quad_perm [2,3,0,1]
    v = dpp_add16<0x141>(v);   // row_half_mirror
    return dpp_add16<0x140>(v);   // row_mirror
}
__device__ __forceinline__ f32x4_t row_sum16(f32x4_t v) {
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = row_sum16(v[r]);
    return v;
}

// ---- chain-layout activation images in LDS ---------------------------------------------------------------------------------------
__device__ __forceinline__ void chain_put(float* __restrict__ buf, int lane, int tile, const f32x4_t& v) {
    st4(buf + tile * kTP + (lane >> 4) * kQP + (lane & 15) * 4, v);
}
template <int NT>
__device__ __forceinline__ void chain_get(const float* __restrict__ buf, int lane, f32x4_t (&v)[NT]) {
    const float* p = buf + (lane >> 4) * kQP + (lane & 15) * 4;
#pragma unroll
    for (int t = 0; t < NT; ++t) v[t] = ld4(p + t * kTP);
    // all NT reads are issued before the first MFMA that consumes one (left alone, hipcc re-uses the consumed weight registers as
    // destinations and issues the reads two at a time between the MFMAs: four exposed LDS round trips per layer); the counted
    // lgkmcnt waits are inserted after scheduling, so the MFMAs of k-tile t still wait for read t only
    __builtin_amdgcn_sched_barrier(0);
}
// element (feature 16 t + i, sample 4 s + q) of an image, for the lane (q, i): the A / B operand of a weight-gradient k-step
__device__ __forceinline__ float chain_elem(const float* __restrict__ buf, int lane, int t, int s) {
    const int i = lane & 15, q = lane >> 4;
    return buf[t * kTP + (i >> 2) * kQP + (4 * s + q) * 4 + (i & 3)];
}

// Timing ablation only (-DPIME_TD3_ABLATE_W: every weight load of the md x md matrices folded into the tensor's first 4 KB -- wrong
// results, the same instructions; what is left is the step without the L2 -> compute-unit weight stream)
#ifdef PIME_TD3_ABLATE_W
#define PIME_TD3_WOFF(x) ((x) & 1023)
#else
#define PIME_TD3_WOFF(x) (x)
#endif
// ---- weights: global -> registers --------------------------------------------------------------------------------------------------
// forward: A operand of output tile t0 + n, k-step (kt, r) = W[16 (t0 + n) + i][16 kt + 4 q + r]: component r of one 16-byte load
// (a wave-uniform base pointer + ONE 32-bit lane offset + compile-time offsets: with a 64-bit per-lane pointer hipcc spends two
// vector adds per load on the address; one wave per SIMD means every such instruction is exposed issue time)
template <int NT, int PER>
__device__ __forceinline__ void load_w(const float* __restrict__ W, int t0, int lane, f32x4_t (&w)[PER][NT]) {
    const int o = (16 * t0 + (lane & 15)) * (NT * 16) + 4 * (lane >> 4);
#pragma unroll
    for (int n = 0; n < PER; ++n)
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) w[n][kt] = ld4(W + PIME_TD3_WOFF(o + n * 16 * (NT * 16) + 16 * kt));
}
// transposed (dX = W^T dZ): A operand of output (= input-feature) tile t0 + n, k-step (kt, r) = W[16 kt + 4 q + r][16 (t0 + n) + i]
template <int NT, int PER>
__device__ __forceinline__ void load_wt(const float* __restrict__ W, int t0, int lane, f32x4_t (&w)[PER][NT]) {
    const int o = (4 * (lane >> 4)) * (NT * 16) + 16 * t0 + (lane & 15);
#pragma unroll
    for (int kt = 0; kt < NT; ++kt)
#pragma unroll
        for (int n = 0; n < PER; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) w[n][kt][r] = W[PIME_TD3_WOFF(o + (16 * kt + r) * (NT * 16) + 16 * n)];
}

// out[n] += W in (output tiles t0 .. t0 + PER - 1).  The caller initialises out: bias_get IN FRONT of the barrier that publishes `in`
// (the bias lives in the small-tensor image; read behind the barrier it was the youngest LDS read in front of the first MFMA, which
// then waited for all of the activation's reads -- lgkmcnt(0) -- instead of the first), or zero4 for the backward chain.
template <int PER>
__device__ __forceinline__ void bias_get(const float* __restrict__ bias, int t0, int lane, f32x4_t (&out)[PER]) {
#pragma unroll
    for (int n = 0; n < PER; ++n) out[n] = ld4(bias + 16 * (t0 + n) + 4 * (lane >> 4));
}
template <int PER>
__device__ __forceinline__ void zero4(f32x4_t (&out)[PER]) {
#pragma unroll
    for (int n = 0; n < PER; ++n) out[n] = f32x4_t{0.f, 0.f, 0.f, 0.f};
}
template <int NT, int PER>
__device__ __forceinline__ void layer(const f32x4_t (&w)[PER][NT], const f32x4_t (&in)[NT], f32x4_t (&out)[PER]) {
#pragma unroll
    for (int kt = 0; kt < NT; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int n = 0; n < PER; ++n) out[n] = mfma16(w[n][kt][r], in[kt][r], out[n]);
}

// first layer, fan-in Din <= 8: k-step 0 = input columns 0..3, k-step 1 = columns 4..7.  x0 / x1: this lane's B operands, input
// column q / 4 + q of sample j (0 beyond Din).
template <int PER>
__device__ __forceinline__ void layer_first(const float* __restrict__ W, const float* __restrict__ bias, int Din, int t0, int lane,
                                            float x0, float x1, f32x4_t (&out)[PER]) {
    const int i = lane & 15, q = lane >> 4;
#pragma unroll
    for (int n = 0; n < PER; ++n) {
        const float* row = W + (size_t)(16 * (t0 + n) + i) * Din;
        const float a0 = q < Din ? row[q] : 0.f;
        const float a1 = 4 + q < Din ? row[4 + q] : 0.f;
        out[n] = ld4(bias + 16 * (t0 + n) + 4 * q);
        out[n] = mfma16(a0, x0, out[n]);
        if (Din > 4) out[n] = mfma16(a1, x1, out[n]);
    }
}

// partial head: sum over this wave's PER * 16 features of w[f] h[j][f], for the lane's sample j (same value in the four lane groups)
template <int PER>
__device__ __forceinline__ float head_partial(const float* __restrict__ w, int t0, int lane, const f32x4_t (&h)[PER]) {
    float p = 0.f;
#pragma unroll
    for (int n = 0; n < PER; ++n) {
        const f32x4_t wv = ld4(w + 16 * (t0 + n) + 4 * (lane >> 4));
#pragma unroll
        for (int r = 0; r < 4; ++r) p = fmaf(h[n][r], wv[r], p);
    }
    p += __shfl_xor(p, 16);
    p += __shfl_xor(p, 32);
    return p;
}
// cross-wave sums through LDS: slot = kRedSlot floats [wave][sample] (up to eight waves), summed in wave order
constexpr int kRedSlot = 128;
__device__ __forceinline__ void red_put(float* __restrict__ red, int slot, int wave, int lane, float p) {
    if (lane < 16) red[slot * kRedSlot + wave * 16 + lane] = p;
}
template <int NW>
__device__ __forceinline__ float red_get(const float* __restrict__ red, int slot, int lane) {
    const float* p = red + slot * kRedSlot + (lane & 15);
    float s = p[0] + p[16];
#pragma unroll
    for (int w = 2; w < NW; ++w) s += p[16 * w];
    return s;
}

// ---- weight gradients ----------------------------------------------------------------------------------------------------------------
// acc[n][b] = sum over the tile's samples of dZ[s][16 (t0 + n) + .] (x) H[s][16 b + .]   (both operands from chain images)
template <int NT, int PER>
__device__ __forceinline__ void dw_blocks(const float* __restrict__ dz, const float* __restrict__ h, int t0, int lane,
                                          f32x4_t (&acc)[PER][NT]) {
#pragma unroll
    for (int n = 0; n < PER; ++n)
#pragma unroll
        for (int b = 0; b < NT; ++b) acc[n][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        float av[PER], bv[NT];
#pragma unroll
        for (int n = 0; n < PER; ++n) av[n] = chain_elem(dz, lane, t0 + n, s);
#pragma unroll
        for (int b = 0; b < NT; ++b) bv[b] = chain_elem(h, lane, b, s);
#pragma unroll
        for (int n = 0; n < PER; ++n)
#pragma unroll
            for (int b = 0; b < NT; ++b) acc[n][b] = mfma16(av[n], bv[b], acc[n][b]);
    }
}
// first-layer weight gradient: B = the tile's input rows [16 samples][16 columns, zero beyond Din] (xin)
template <int PER>
__device__ __forceinline__ void dw_first(const float* __restrict__ dz, const float* __restrict__ xin, int t0, int lane,
                                         f32x4_t (&acc)[PER]) {
#pragma unroll
    for (int n = 0; n < PER; ++n) acc[n] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const float bv = xin[(4 * s + (lane >> 4)) * 16 + (lane & 15)];
#pragma unroll
        for (int n = 0; n < PER; ++n) acc[n] = mfma16(chain_elem(dz, lane, t0 + n, s), bv, acc[n]);
    }
}
__device__ __forceinline__ void slab_put(float* __restrict__ p, f32x4_t v, bool accum) {
    if (accum) v += ld4(p);
    st4(p, v);
}
// a finished weight-gradient job -> the slab, block-major (one 16-byte store per lane and block).  The accumulate / overwrite decision
// (a later sample group of the same workgroup: batches beyond 512 tiles) is taken once per job, not per block.
template <int NT, int PER>
__device__ __forceinline__ void slab_blocks(float* __restrict__ seg, int t0, int lane, const f32x4_t (&acc)[PER][NT], bool accum) {
    float* const p = seg + (t0 * NT * 64 + lane) * 4;
    if (!accum) {
#pragma unroll
        for (int n = 0; n < PER; ++n)
#pragma unroll
            for (int b = 0; b < NT; ++b) st4(p + (n * NT + b) * 256, acc[n][b]);
    } else {
#pragma unroll
        for (int n = 0; n < PER; ++n)
#pragma unroll
            for (int b = 0; b < NT; ++b) st4(p + (n * NT + b) * 256, acc[n][b] + ld4(p + (n * NT + b) * 256));
    }
}
// a vector gradient (bias, head weights) of this wave's features: v = per-sample terms, summed over the tile's samples
template <int PER>
__device__ __forceinline__ void vec_grad(float* __restrict__ seg, int t0, int lane, const f32x4_t (&v)[PER], bool accum) {
#pragma unroll
    for (int n = 0; n < PER; ++n) {
        const f32x4_t s = row_sum16(v[n]);
        if ((lane & 15) == 0) slab_put(seg + 16 * (t0 + n) + 4 * (lane >> 4), s, accum);
    }
}

// The SMALL tensors of a net -- first-layer weights, biases, heads: everything but the md x md matrices -- are read by every wave at
// the moment a layer starts; from global memory each such read is an exposed L2 round trip on the workgroup's critical path (a dozen
// per kernel).  They are copied into LDS once per workgroup, behind the minibatch gather: the flat tensors minus the big matrices,
// in the same order (td3.hpp), so that a small-image offset is the flat offset minus the matrices in front of it.
struct Td3SmallActor { int W1, b1, b2, b3, w4, b4, total; };
struct Td3SmallCritic { int W1, b1, b2, q1w, q1b, q2w, q2b, total; };
__host__ __device__ inline Td3SmallActor td3_small_actor(int D, int md) {
    const Td3ActorOff P = td3_actor_off(D, md);
    const int mm = md * md;
    return Td3SmallActor{P.W1, P.b1, P.b2 - mm, P.b3 - 2 * mm, P.w4 - 2 * mm, P.b4 - 2 * mm, P.total - 2 * mm};
}
__host__ __device__ inline Td3SmallCritic td3_small_critic(int D, int md) {
    const Td3CriticOff P = td3_critic_off(D, md);
    const int mm = md * md;
    return Td3SmallCritic{P.W1, P.b1, P.b2 - mm, P.q1w - mm, P.q1b - mm, P.q2w - mm, P.q2b - mm, P.total - mm};
}

struct Td3Lds {
    int buf[4], xin, red, small[3], total;
};
__host__ __device__ constexpr int td3_buf_floats(int NT) { return NT * kTP; }
__host__ __device__ inline Td3Lds td3_lds(int NT, int D) {
    Td3Lds L{};
    int o = 0;
    for (int k = 0; k < 4; ++k) { L.buf[k] = o; o += td3_buf_floats(NT); }
    L.xin = o; o += 16 * 16;
    L.red = o; o += 8 * kRedSlot;
    const int md = NT * 16, sa = td3_small_actor(D, md).total, sc = td3_small_critic(D, md).total;
    L.small[0] = o; o += sa;                 // the launch's actor (critic launch: the target actor)
    L.small[1] = o; o += sc;                 // the launch's critic (critic launch: the online critic; actor launch: the target critic)
    L.small[2] = o; o += sc;                 // critic launch only: the target critic
    L.total = o;
    return L;
}

// up to 256 16-byte words of a flat tensor, this thread's share: loaded here, written to LDS by small_store (the caller puts other
// loads in between, so that one memory round trip covers them all)
__device__ __forceinline__ f32x4_t small_load(const float* __restrict__ src, int n4, int tid) {
    return tid < n4 ? ld4(src + 4 * tid) : f32x4_t{0.f, 0.f, 0.f, 0.f};
}
__device__ __forceinline__ void small_store(float* __restrict__ dst, int n4, int tid, const f32x4_t& v) {
    if (tid < n4) st4(dst + 4 * tid, v);
}

// this lane's smoothing-noise draw for batch position pos
__device__ __forceinline__ float td3_noise(const Td3Batch& b, long long trow, int pos) {
    if (b.noise) return b.noise[(size_t)trow * b.B + pos];
    double ua, ub;
    const uint32_t epoch = b.noise_epoch + (b.epoch ? (uint32_t)b.epoch[0] : 0u);   // bumped by the host per update
    philox_pair(b.noise_seed, (uint32_t)pos, epoch, (uint32_t)trow, STREAM_TD3_SMOOTH, ua, ub);
    // Box-Muller (cosine branch) in float32 on the hardware transcendentals: the float64 log / sqrt / cos of the exploration kernels
    // are several hundred instructions, exposed issue time in front of this kernel's first barrier (1 - ua in (0, 1]: log finite)
    const float rad = __builtin_sqrtf(-2.0f * __logf((float)(1.0 - ua)));
    return rad * __cosf(6.2831853071795864769f * (float)ub);
}

// ======================================================================================================== critic gradients
// DD: the state width as a compile-time constant (3: pH, 4: water tank Integrator), 0: read from the arguments.  With DD fixed every
// offset of the parameter / slab / LDS layouts folds into an immediate; as run-time values they are ~100 live scalars that hipcc
// spills through VGPR lanes (v_readlane / v_writelane) and re-derives with scalar arithmetic in every phase.
template <int MD, int DD, int NW>
__global__ __launch_bounds__(NW * 64, 1) void td3_critic_kernel(Td3GradArgs a) {
    constexpr int NT = MD / 16, PER = NT / NW;
    static_assert(PER >= 1 && PER * NW == NT, "the waves split a layer's output tiles evenly");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int D = DD ? DD : a.D, Dc = D + 1;
    const Td3Lds F = td3_lds(NT, D);
    float* const B0 = lds + F.buf[0];
    float* const B1 = lds + F.buf[1];
    float* const B2 = lds + F.buf[2];
    float* const xin = lds + F.xin;
    float* const red = lds + F.red;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int t0 = wave * PER;
    const Td3ActorOff PA = td3_actor_off(D, MD);
    const Td3CriticOff PC = td3_critic_off(D, MD);
    const Td3SmallActor SA = td3_small_actor(D, MD);
    const Td3SmallCritic SC = td3_small_critic(D, MD);
    const float* const at = lds + F.small[0];   // target actor, small tensors
    const float* const cr = lds + F.small[1];   // online critic
    const float* const ct = lds + F.small[2];   // target critic
    const Td3SlabLayout SL = td3_critic_slab(D, MD);
    const float invB = 1.0f / (float)a.b.B;
    const long long trow = a.b.row;
    float* const sl = a.slab + (size_t)blockIdx.x * a.stride;
    float loss_acc = 0.f;   // wave 0, lanes 0..15: this workgroup's loss terms
    TD3_MARK(0);
    const long long cyc0 = a.trace ? (long long)__builtin_readcyclecounter() : 0;   // shader clock (s_memtime) beside the 100 MHz marks

#pragma unroll 1
    for (int group = blockIdx.x; group < a.ngroups; group += gridDim.x) {
        int lane = tid & 63;
        asm volatile("" : "+v"(lane));
        const int j = lane & 15, q = lane >> 4;
        const bool accum = group != (int)blockIdx.x;
        const int pos = group * kTd3Tile + j;
        const bool valid = pos < a.b.B;
        const int p = valid ? pos : a.b.B - 1;
        const long long row = a.b.idx[(size_t)trow * a.b.B + p], nrow = a.b.nxt[(size_t)trow * a.b.B + p];
        f32x4_t wA[PER][NT], wB[PER][NT], in[NT];
        load_w<NT, PER>(a.act + PA.W2, t0, lane, wA);   // the first md x md weights: in flight behind the gather's two round trips
        // the nets' small tensors ride behind the index loads (first group only): seven 16-byte loads per thread, one round trip
        const bool stage = !accum;
        const int nA0 = (PA.W2 - PA.W1) / 4, nA1 = MD / 4, nA2 = (PA.total - PA.b3) / 4, nC0 = (PC.W2 - PC.W1) / 4, nC1 = (PC.total - PC.b2) / 4;
        f32x4_t sv[9];
        if (stage) {
            sv[0] = small_load(a.act + PA.W1, nA0, tid); sv[1] = small_load(a.act + PA.b2, nA1, tid); sv[2] = small_load(a.act + PA.b3, nA2, tid);
            sv[3] = small_load(a.cri + PC.W1, nC0, tid); sv[4] = small_load(a.cri + PC.b2, nC1, tid);
            sv[5] = small_load(a.cri_target + PC.W1, nC0, tid); sv[6] = small_load(a.cri_target + PC.b2, nC1, tid);
            sv[7] = small_load(a.cri + PC.W1 + 1024, nC0 - 256, tid); sv[8] = small_load(a.cri_target + PC.W1 + 1024, nC0 - 256, tid);   // D = 7 only
        }
        // first-layer B operands: input column q / 4 + q of sample j
        const float* srow = a.b.state + (size_t)row * D;
        const float* nsrow = a.b.state + (size_t)nrow * D;
        const float s0 = q < D ? srow[q] : 0.f, s1 = 4 + q < D ? srow[4 + q] : 0.f;
        const float n0 = q < D ? nsrow[q] : 0.f, n1 = 4 + q < D ? nsrow[4 + q] : 0.f;
        const float* orow = a.b.other + (size_t)row * 3;
        const float reward = orow[0], mask = orow[1], action = orow[2];
        const float eps = td3_noise(a.b, trow, p);
        if (stage) {
            float* const w = lds + F.small[0];
            small_store(w + SA.W1, nA0, tid, sv[0]); small_store(w + SA.b2, nA1, tid, sv[1]); small_store(w + SA.b3, nA2, tid, sv[2]);
            float* const c = lds + F.small[1];
            small_store(c + SC.W1, nC0, tid, sv[3]); small_store(c + SC.b2, nC1, tid, sv[4]);
            float* const t = lds + F.small[2];
            small_store(t + SC.W1, nC0, tid, sv[5]); small_store(t + SC.b2, nC1, tid, sv[6]);
            small_store(c + SC.W1 + 1024, nC0 - 256, tid, sv[7]); small_store(t + SC.W1 + 1024, nC0 - 256, tid, sv[8]);
        }
        TD3_BARRIER();   // the previous group is done with the LDS images; the small tensors are in
        TD3_MARK(1);   // gather + small tensors
        // the online critic's input [s, a, 0 ..]: column q / 4 + q of sample j (this lane's first-layer B operands)
        const float xs0 = q < D ? s0 : (q == D ? action : 0.f), xs1 = 4 + q < D ? s1 : (4 + q == D ? action : 0.f);
        if (wave == 0) {   // ... as rows [16 samples][16 columns] for its first-layer weight gradient, and for the actor launch
            xin[j * 16 + q] = xs0; xin[j * 16 + 4 + q] = xs1; xin[j * 16 + 8 + q] = 0.f; xin[j * 16 + 12 + q] = 0.f;
            if (valid) { a.xg[(size_t)pos * 8 + q] = xs0; a.xg[(size_t)pos * 8 + 4 + q] = xs1; }
        }

        // ------------------------------------------------------------------ next_a = clamp(tanh(act_target(s')) + clamp(noise))
        {
            f32x4_t h[PER];
            layer_first<PER>(at + SA.W1, at + SA.b1, D, t0, lane, n0, n1, h);
#pragma unroll
            for (int n = 0; n < PER; ++n) chain_put(B0, lane, t0 + n, relu4(h[n]));
        }
        load_w<NT, PER>(a.act + PA.W3, t0, lane, wB);
        f32x4_t hb[PER];   // the next layer's accumulators, initialised with its bias in front of the barrier
        bias_get<PER>(at + SA.b2, t0, lane, hb);
        TD3_BARRIER();
        TD3_MARK(2);   // target actor layer 1
        chain_get<NT>(B0, lane, in);
        {
            layer<NT, PER>(wA, in, hb);
#pragma unroll
            for (int n = 0; n < PER; ++n) chain_put(B1, lane, t0 + n, relu4(hb[n]));
        }
        load_w<NT, PER>(a.cri_target + PC.W2, t0, lane, wA);
        bias_get<PER>(at + SA.b3, t0, lane, hb);
        TD3_BARRIER();
        TD3_MARK(3);   // layer 2
        chain_get<NT>(B1, lane, in);
        float next_a;
        {
            f32x4_t (&h)[PER] = hb;
            layer<NT, PER>(wB, in, h);
#pragma unroll
            for (int n = 0; n < PER; ++n) h[n] = relu4(h[n]);
            red_put(red, 0, wave, lane, head_partial<PER>(at + SA.w4, t0, lane, h));
            TD3_BARRIER();
            const float pre = red_get<NW>(red, 0, lane) + at[SA.b4];
            const float nz = fminf(fmaxf(eps * a.b.policy_noise, -a.b.noise_clip), a.b.noise_clip);   // net.py:109
            next_a = fminf(fmaxf(tanhf(pre) + nz, -1.0f), 1.0f);
        }
        TD3_MARK(4);   // layer 3 + head: next_a
        // ------------------------------------------------------------------ q_label = r + mask * min(cri_target twin heads)(s', next_a)
        load_w<NT, PER>(a.cri + PC.W2, t0, lane, wB);
        float label;
        {
            const float x0 = q < D ? n0 : (q == D ? next_a : 0.f);
            const float x1 = 4 + q < D ? n1 : (4 + q == D ? next_a : 0.f);
            f32x4_t h[PER];
            layer_first<PER>(ct + SC.W1, ct + SC.b1, Dc, t0, lane, x0, x1, h);
#pragma unroll
            for (int n = 0; n < PER; ++n) chain_put(B0, lane, t0 + n, relu4(h[n]));
            bias_get<PER>(ct + SC.b2, t0, lane, h);
            TD3_BARRIER();
            chain_get<NT>(B0, lane, in);
            layer<NT, PER>(wA, in, h);
#pragma unroll
            for (int n = 0; n < PER; ++n) h[n] = relu4(h[n]);
            red_put(red, 1, wave, lane, head_partial<PER>(ct + SC.q1w, t0, lane, h));
            red_put(red, 2, wave, lane, head_partial<PER>(ct + SC.q2w, t0, lane, h));
            TD3_BARRIER();
            const float tq1 = red_get<NW>(red, 1, lane) + ct[SC.q1b], tq2 = red_get<NW>(red, 2, lane) + ct[SC.q2b];
            label = reward + mask * fminf(tq1, tq2);
        }
        TD3_MARK(5);   // target critic: label
        // ------------------------------------------------------------------ online twin critic on (s, a): forward
        f32x4_t h1[PER], h2[PER];
        {
            layer_first<PER>(cr + SC.W1, cr + SC.b1, Dc, t0, lane, xs0, xs1, h1);
#pragma unroll
            for (int n = 0; n < PER; ++n) { h1[n] = relu4(h1[n]); chain_put(B1, lane, t0 + n, h1[n]); }
        }
        load_wt<NT, PER>(a.cri + PC.W2, t0, lane, wA);   // for dH1 = W2^T dZ2
        bias_get<PER>(cr + SC.b2, t0, lane, h2);
        TD3_BARRIER();
        chain_get<NT>(B1, lane, in);
        layer<NT, PER>(wB, in, h2);
#pragma unroll
        for (int n = 0; n < PER; ++n) h2[n] = relu4(h2[n]);
        red_put(red, 3, wave, lane, head_partial<PER>(cr + SC.q1w, t0, lane, h2));
        red_put(red, 4, wave, lane, head_partial<PER>(cr + SC.q2w, t0, lane, h2));
        TD3_BARRIER();
        TD3_MARK(6);   // online critic forward
        // ------------------------------------------------------------------ SmoothL1 x 2 (beta = 1, mean) and its gradient
        float g1 = 0.f, g2 = 0.f;
        {
            const float d1 = red_get<NW>(red, 3, lane) + cr[SC.q1b] - label, d2 = red_get<NW>(red, 4, lane) + cr[SC.q2b] - label;
            const float a1 = fabsf(d1), a2 = fabsf(d2);
            if (valid) {
                g1 = (a1 < 1.f ? d1 : (d1 > 0.f ? 1.f : -1.f)) * invB;
                g2 = (a2 < 1.f ? d2 : (d2 > 0.f ? 1.f : -1.f)) * invB;
                if (wave == 0 && q == 0) loss_acc += (a1 < 1.f ? 0.5f * d1 * d1 : a1 - 0.5f) + (a2 < 1.f ? 0.5f * d2 * d2 : a2 - 0.5f);
            }
        }
        // heads: weight / bias gradients, dZ2 = (g1 wq1 + g2 wq2) [h2 > 0]
        {
            f32x4_t v1[PER], v2[PER], dz[PER];
#pragma unroll
            for (int n = 0; n < PER; ++n) {
                const f32x4_t w1 = ld4(cr + SC.q1w + 16 * (t0 + n) + 4 * q), w2 = ld4(cr + SC.q2w + 16 * (t0 + n) + 4 * q);
                v1[n] = h2[n] * g1;
                v2[n] = h2[n] * g2;
                dz[n] = gate4(w1 * g1 + w2 * g2, h2[n]);
                chain_put(B2, lane, t0 + n, dz[n]);
            }
            vec_grad<PER>(sl + SL.seg[4].slab_off, t0, lane, v1, accum);
            vec_grad<PER>(sl + SL.seg[6].slab_off, t0, lane, v2, accum);
            vec_grad<PER>(sl + SL.seg[3].slab_off, t0, lane, dz, accum);   // net_sa.2 bias
            if (wave == 0) {
                const float b1 = row_sum16(g1), b2 = row_sum16(g2);
                if (lane == 0) {
                    float* p1 = sl + SL.seg[5].slab_off;
                    float* p2 = sl + SL.seg[7].slab_off;
                    p1[0] = accum ? p1[0] + b1 : b1;
                    p2[0] = accum ? p2[0] + b2 : b2;
                }
            }
        }
        TD3_BARRIER();   // dZ2 published
        TD3_MARK(7);   // loss, head gradients, dZ2
        {   // net_sa.2 weight gradient
            f32x4_t acc[PER][NT];
            dw_blocks<NT, PER>(B2, B1, t0, lane, acc);
            slab_blocks<NT, PER>(sl + SL.seg[2].slab_off, t0, lane, acc, accum);
        }
        TD3_NO_HOIST();
        TD3_MARK(8);   // dW2
        chain_get<NT>(B2, lane, in);
        {
            f32x4_t d1[PER];
            zero4<PER>(d1);
            layer<NT, PER>(wA, in, d1);
#pragma unroll
            for (int n = 0; n < PER; ++n) { d1[n] = gate4(d1[n], h1[n]); chain_put(B0, lane, t0 + n, d1[n]); }
            vec_grad<PER>(sl + SL.seg[1].slab_off, t0, lane, d1, accum);   // net_sa.0 bias
        }
        TD3_BARRIER();   // dZ1 published
        TD3_MARK(9);   // dH1, dZ1
        {
            f32x4_t acc[PER];
            dw_first<PER>(B0, xin, t0, lane, acc);
            float* seg = sl + SL.seg[0].slab_off;
#pragma unroll
            for (int n = 0; n < PER; ++n) slab_put(seg + ((t0 + n) * 64 + lane) * 4, acc[n], accum);
        }
    }
    TD3_MARK(10);   // dW1
    if (a.trace && blockIdx.x == 0 && threadIdx.x == 0) a.trace[30] = (long long)__builtin_readcyclecounter() - cyc0;
    if (wave == 0) {
        const float t = row_sum16(loss_acc);
        if (tid == 0) st4(sl + SL.scalar_off, f32x4_t{t, 0.f, 0.f, 0.f});
    }
}

// ======================================================================================================== actor gradients
// obj_actor = -mean(cri_target.q1(s, tanh(act(s))))  (agent.py:323-324), differentiated down to the actor's parameters
template <int MD, int DD, int NW>
__global__ __launch_bounds__(NW * 64, 1) void td3_actor_kernel(Td3GradArgs a) {
    constexpr int NT = MD / 16, PER = NT / NW;
    static_assert(PER >= 1 && PER * NW == NT, "the waves split a layer's output tiles evenly");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int D = DD ? DD : a.D, Dc = D + 1;
    const Td3Lds F = td3_lds(NT, D);
    float* const B0 = lds + F.buf[0];
    float* const B1 = lds + F.buf[1];
    float* const B2 = lds + F.buf[2];
    float* const B3 = lds + F.buf[3];
    float* const xin = lds + F.xin;
    float* const red = lds + F.red;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int t0 = wave * PER;
    const Td3ActorOff PA = td3_actor_off(D, MD);
    const Td3CriticOff PC = td3_critic_off(D, MD);
    const Td3SmallActor SA = td3_small_actor(D, MD);
    const Td3SmallCritic SC = td3_small_critic(D, MD);
    const float* const ac = lds + F.small[0];   // online actor, small tensors
    const float* const ct = lds + F.small[1];   // target critic
    const Td3SlabLayout SL = td3_actor_slab(D, MD);
    const float invB = 1.0f / (float)a.b.B;
    float* const sl = a.slab + (size_t)blockIdx.x * a.stride;
    float q_acc = 0.f;
    TD3_MARK(0);
    const long long cyc0 = a.trace ? (long long)__builtin_readcyclecounter() : 0;

#pragma unroll 1
    for (int group = blockIdx.x; group < a.ngroups; group += gridDim.x) {
        int lane = tid & 63;
        asm volatile("" : "+v"(lane));
        const int j = lane & 15, q = lane >> 4;
        const bool accum = group != (int)blockIdx.x;
        const int pos = group * kTd3Tile + j;
        const bool valid = pos < a.b.B;
        const int p = valid ? pos : a.b.B - 1;
        f32x4_t wA[PER][NT], wB[PER][NT], in[NT];
        load_w<NT, PER>(a.act + PA.W2, t0, lane, wA);
        // the minibatch's state rows as the critic launch of this step gathered them (one round trip instead of index -> row)
        const float s0 = q < D ? a.xg[(size_t)p * 8 + q] : 0.f, s1 = 4 + q < D ? a.xg[(size_t)p * 8 + 4 + q] : 0.f;
        const bool stage = !accum;   // the small tensors ride along (first group only)
        const int nA0 = (PA.W2 - PA.W1) / 4, nA1 = MD / 4, nA2 = (PA.total - PA.b3) / 4, nC0 = (PC.W2 - PC.W1) / 4, nC1 = (PC.total - PC.b2) / 4;
        f32x4_t sv[6];
        if (stage) {
            sv[0] = small_load(a.act + PA.W1, nA0, tid); sv[1] = small_load(a.act + PA.b2, nA1, tid); sv[2] = small_load(a.act + PA.b3, nA2, tid);
            sv[3] = small_load(a.cri + PC.W1, nC0, tid); sv[4] = small_load(a.cri + PC.b2, nC1, tid);
            sv[5] = small_load(a.cri + PC.W1 + 1024, nC0 - 256, tid);   // D = 7 only
        }
        if (stage) {
            float* const w = lds + F.small[0];
            small_store(w + SA.W1, nA0, tid, sv[0]); small_store(w + SA.b2, nA1, tid, sv[1]); small_store(w + SA.b3, nA2, tid, sv[2]);
            float* const c = lds + F.small[1];
            small_store(c + SC.W1, nC0, tid, sv[3]); small_store(c + SC.b2, nC1, tid, sv[4]);
            small_store(c + SC.W1 + 1024, nC0 - 256, tid, sv[5]);
        }
        TD3_BARRIER();   // the previous group is done with the LDS images; the small tensors are in
        TD3_MARK(1);
        if (wave == 0) {   // the actor's input rows [16 samples][16 columns, zero beyond D] for its first-layer weight gradient
            xin[j * 16 + q] = s0; xin[j * 16 + 4 + q] = s1; xin[j * 16 + 8 + q] = 0.f; xin[j * 16 + 12 + q] = 0.f;
        }
        f32x4_t a1[PER], a2[PER], a3[PER], c1[PER], c2[PER];

        // ------------------------------------------------------------------ action = tanh(act(s))
        layer_first<PER>(ac + SA.W1, ac + SA.b1, D, t0, lane, s0, s1, a1);
#pragma unroll
        for (int n = 0; n < PER; ++n) { a1[n] = relu4(a1[n]); chain_put(B0, lane, t0 + n, a1[n]); }
        load_w<NT, PER>(a.act + PA.W3, t0, lane, wB);
        bias_get<PER>(ac + SA.b2, t0, lane, a2);   // a layer's accumulators start as its bias, read in front of the barrier
        TD3_BARRIER();
        chain_get<NT>(B0, lane, in);
        layer<NT, PER>(wA, in, a2);
#pragma unroll
        for (int n = 0; n < PER; ++n) { a2[n] = relu4(a2[n]); chain_put(B1, lane, t0 + n, a2[n]); }
        load_w<NT, PER>(a.cri + PC.W2, t0, lane, wA);
        bias_get<PER>(ac + SA.b3, t0, lane, a3);
        TD3_BARRIER();
        chain_get<NT>(B1, lane, in);
        layer<NT, PER>(wB, in, a3);
#pragma unroll
        for (int n = 0; n < PER; ++n) a3[n] = relu4(a3[n]);
        red_put(red, 0, wave, lane, head_partial<PER>(ac + SA.w4, t0, lane, a3));
        TD3_BARRIER();
        const float act = tanhf(red_get<NW>(red, 0, lane) + ac[SA.b4]);
        TD3_MARK(2);   // actor forward
        // ------------------------------------------------------------------ q1 = cri_target.q1(s, action)
        {
            const float x0 = q < D ? s0 : (q == D ? act : 0.f);
            const float x1 = 4 + q < D ? s1 : (4 + q == D ? act : 0.f);
            layer_first<PER>(ct + SC.W1, ct + SC.b1, Dc, t0, lane, x0, x1, c1);
#pragma unroll
            for (int n = 0; n < PER; ++n) { c1[n] = relu4(c1[n]); chain_put(B2, lane, t0 + n, c1[n]); }
        }
        load_wt<NT, PER>(a.cri + PC.W2, t0, lane, wB);   // dC1 = W2^T dZc2
        bias_get<PER>(ct + SC.b2, t0, lane, c2);
        TD3_BARRIER();
        chain_get<NT>(B2, lane, in);
        layer<NT, PER>(wA, in, c2);
#pragma unroll
        for (int n = 0; n < PER; ++n) c2[n] = relu4(c2[n]);
        red_put(red, 1, wave, lane, head_partial<PER>(ct + SC.q1w, t0, lane, c2));   // (the value itself only feeds the logged objective)
        // ------------------------------------------------------------------ backward through the critic to the action
        const float g = valid ? -invB : 0.f;   // d(-mean q1) / d q1
#pragma unroll
        for (int n = 0; n < PER; ++n) {
            const f32x4_t wq = ld4(ct + SC.q1w + 16 * (t0 + n) + 4 * q);
            chain_put(B3, lane, t0 + n, gate4(wq * g, c2[n]));
        }
        load_wt<NT, PER>(a.act + PA.W3, t0, lane, wA);   // dA2 = W3^T dZ3
        TD3_BARRIER();
        TD3_MARK(3);   // target critic forward, dZc2
        if (valid && wave == 0 && q == 0) q_acc += red_get<NW>(red, 1, lane) + ct[SC.q1b];
        chain_get<NT>(B3, lane, in);
        float dpre;
        {
            f32x4_t d[PER];
            zero4<PER>(d);
            layer<NT, PER>(wB, in, d);
            float pa = 0.f;   // d obj / d action = sum_f W1[f][D] dZc1[f]
#pragma unroll
            for (int n = 0; n < PER; ++n) {
                d[n] = gate4(d[n], c1[n]);
#pragma unroll
                for (int r = 0; r < 4; ++r) pa = fmaf(d[n][r], ct[SC.W1 + (16 * (t0 + n) + 4 * q + r) * Dc + D], pa);
            }
            pa += __shfl_xor(pa, 16);
            pa += __shfl_xor(pa, 32);
            red_put(red, 2, wave, lane, pa);
            TD3_BARRIER();
            dpre = red_get<NW>(red, 2, lane) * (1.0f - act * act);   // tanh'
        }
        TD3_MARK(4);   // critic backward to the action
        // ------------------------------------------------------------------ actor backward + weight gradients
        {
            f32x4_t v[PER], dz[PER];
#pragma unroll
            for (int n = 0; n < PER; ++n) {
                const f32x4_t w4 = ld4(ac + SA.w4 + 16 * (t0 + n) + 4 * q);
                v[n] = a3[n] * dpre;
                dz[n] = gate4(w4 * dpre, a3[n]);
                chain_put(B2, lane, t0 + n, dz[n]);
            }
            vec_grad<PER>(sl + SL.seg[6].slab_off, t0, lane, v, accum);    // net.6 weight
            vec_grad<PER>(sl + SL.seg[5].slab_off, t0, lane, dz, accum);   // net.4 bias
            if (wave == 0) {
                const float bg = row_sum16(dpre);
                if (lane == 0) {
                    float* pb = sl + SL.seg[7].slab_off;
                    pb[0] = accum ? pb[0] + bg : bg;
                }
            }
        }
        load_wt<NT, PER>(a.act + PA.W2, t0, lane, wB);   // dA1 = W2^T dZ2
        TD3_BARRIER();   // dZ3 published
        TD3_MARK(5);
        {
            f32x4_t acc[PER][NT];
            dw_blocks<NT, PER>(B2, B1, t0, lane, acc);   // net.4: dZ3^T A2
            slab_blocks<NT, PER>(sl + SL.seg[4].slab_off, t0, lane, acc, accum);
        }
        TD3_NO_HOIST();
        chain_get<NT>(B2, lane, in);
        {
            f32x4_t d[PER];
            zero4<PER>(d);
            layer<NT, PER>(wA, in, d);
#pragma unroll
            for (int n = 0; n < PER; ++n) { d[n] = gate4(d[n], a2[n]); chain_put(B3, lane, t0 + n, d[n]); }
            vec_grad<PER>(sl + SL.seg[3].slab_off, t0, lane, d, accum);    // net.2 bias
        }
        TD3_BARRIER();   // dZ2 published
        TD3_MARK(6);   // dW3, dA2
        {
            f32x4_t acc[PER][NT];
            dw_blocks<NT, PER>(B3, B0, t0, lane, acc);   // net.2: dZ2^T A1
            slab_blocks<NT, PER>(sl + SL.seg[2].slab_off, t0, lane, acc, accum);
        }
        TD3_NO_HOIST();
        chain_get<NT>(B3, lane, in);
        {
            f32x4_t d[PER];
            zero4<PER>(d);
            layer<NT, PER>(wB, in, d);
#pragma unroll
            for (int n = 0; n < PER; ++n) { d[n] = gate4(d[n], a1[n]); chain_put(B1, lane, t0 + n, d[n]); }
            vec_grad<PER>(sl + SL.seg[1].slab_off, t0, lane, d, accum);    // net.0 bias
        }
        TD3_BARRIER();   // dZ1 published
        TD3_MARK(7);   // dW2, dA1
        {
            f32x4_t acc[PER];
            dw_first<PER>(B1, xin, t0, lane, acc);
            float* seg = sl + SL.seg[0].slab_off;
#pragma unroll
            for (int n = 0; n < PER; ++n) slab_put(seg + ((t0 + n) * 64 + lane) * 4, acc[n], accum);
        }
    }
    TD3_MARK(8);   // dW1
    if (a.trace && blockIdx.x == 0 && threadIdx.x == 0) a.trace[30] = (long long)__builtin_readcyclecounter() - cyc0;
    if (wave == 0) {
        const float t = row_sum16(q_acc);
        if (tid == 0) st4(sl + SL.scalar_off, f32x4_t{t, 0.f, 0.f, 0.f});
    }
}

// ======================================================================================================== slab reduction + Adam + soft update
// Workgroup = kApplyWords (64) consecutive 16-byte words of the slab layout x kApplyGroups (8) slab groups (the first version: 16 x 16;
// 1 KB contiguous per slab and 32 loads per thread in flight read the freshly written slabs 23 % faster, see the sweep below):
// thread (g, l) sums word l of slabs g, g + 8, ... with
// ALL of them in flight at once (the slabs were written by other compute units a kernel ago: every load is an Infinity-Cache / HBM
// round trip, and the launch is a few hundred workgroups of one such round trip each -- the first version, 64 words x 4 groups with
// four loads in flight, took 11.5 us of a 78 us optimizer step); the 16 partial sums meet in LDS and are combined in group order
// (bit-reproducible); the threads of group 0 then own four gradient elements each: they write them, apply torch.optim.Adam (defaults:
// no weight decay, no amsgrad) to their parameters and, on a delayed step, target = tau * param + (1 - tau) * target (agent.py:116-124,
// the reference's operand order).
#ifndef PIME_APPLY_WORDS
#define PIME_APPLY_WORDS 64   // swept at the end of round 4 (words x threads, us per launch): 16x256 9.6, 8x256 11.6, 32x256 8.7, 64x256 8.2, 128x256 10.6,
                              // 32x512 8.6, 64x512 7.4, 128x512 9.1, 256x512 14.8, 64x1024 7.5, 32x1024 12.8 (profiles/r04_w_td3_apply_shape_sweep.txt)
#endif
#ifndef PIME_APPLY_THREADS
#define PIME_APPLY_THREADS 512
#endif
constexpr int kApplyThreads = PIME_APPLY_THREADS;
constexpr int kApplyWords = PIME_APPLY_WORDS, kApplyGroups = kApplyThreads / kApplyWords;
constexpr int kApplyBatch = 256 / kApplyGroups > 32 ? 32 : 256 / kApplyGroups;   // slab loads a thread keeps in flight: one batch covers 256 slabs
__global__ __launch_bounds__(kApplyThreads) void td3_apply_kernel(Td3ApplyArgs a) {
    __shared__ float4 part[kApplyGroups][kApplyWords];
    __shared__ float adam_sh[3];   // [1] step size, [2] sqrt of the second bias correction
    const int tid = threadIdx.x, l = tid & (kApplyWords - 1), g = tid / kApplyWords;
    const int nwords = a.L.stride / 4;
    const int unit = blockIdx.x * kApplyWords + l;
    const bool finisher = g == 0 && unit < nwords;
    if (tid == 64) {   // Adam's bias corrections, off the finishers' critical path.  The step number is a launch argument + a base the
        // host moves once per update: no workgroup writes shared state, so the launch needs no arrival counter (the first version's
        // atomic on one word cost ~12 ns per workgroup: 2.5 us of its 11.5 with 200 workgroups, 10 us with 850)
        const double t = (double)a.step[0] + (double)a.row + 1.0;
        adam_sh[1] = a.lr / (float)(1.0 - pow((double)a.b1, t));
        adam_sh[2] = (float)sqrt(1.0 - pow((double)a.b2, t));
    }
    const bool soft = a.soft != 0;
    // which elements a finisher owns
    long long flat[4] = {-1, -1, -1, -1};
    bool scalar_word = false;
    if (finisher) {
        const int off = unit * 4;
        if (off >= a.L.scalar_off) scalar_word = true;
        else {
            int si = 0;
            while (si + 1 < a.L.nseg && off >= a.L.seg[si + 1].slab_off) ++si;
            const Td3Seg sg = a.L.seg[si];
            const int u = (off - sg.slab_off) / 4;
            if (u < sg.n4) {
                if (sg.tb > 0) {
                    const int blk = u >> 6, ln = u & 63;
                    const int row = (blk / sg.tb) * 16 + 4 * (ln >> 4), col = (blk % sg.tb) * 16 + (ln & 15);
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (col < sg.ncols) flat[k] = sg.flat_off + (long long)(row + k) * sg.ldw + col;
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (u * 4 + k < sg.n) flat[k] = sg.flat_off + u * 4 + k;
                }
            }
        }
    }
    // optimizer state of those elements, requested before the slab loads
    float pm[4] = {0.f, 0.f, 0.f, 0.f}, pv[4] = {0.f, 0.f, 0.f, 0.f}, pp[4] = {0.f, 0.f, 0.f, 0.f}, pt[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (flat[k] >= 0) {
            pm[k] = a.exp_avg[flat[k]]; pv[k] = a.exp_avg_sq[flat[k]]; pp[k] = a.param[flat[k]];
            if (soft) pt[k] = a.target[flat[k]];
        }
    float gin[4] = {0.f, 0.f, 0.f, 0.f};   // mode 2: the (all-reduced) gradient stands in for the slab sums
    if (a.mode == 2) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (flat[k] >= 0) gin[k] = a.grad[flat[k]];
    }
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (unit < nwords && a.mode != 2) {
        const float* base = a.slab + (size_t)unit * 4;
        const size_t stride = (size_t)a.L.stride;
        int s = g;
        for (; s + (kApplyBatch - 1) * kApplyGroups < a.nslabs; s += kApplyBatch * kApplyGroups) {   // kApplyBatch loads in flight (256 slabs: one batch)
            float4 v[kApplyBatch];
#pragma unroll
            for (int k = 0; k < kApplyBatch; ++k) v[k] = *reinterpret_cast<const float4*>(base + (size_t)(s + kApplyGroups * k) * stride);
#pragma unroll
            for (int k = 0; k < kApplyBatch; ++k) { acc.x += v[k].x; acc.y += v[k].y; acc.z += v[k].z; acc.w += v[k].w; }
        }
        for (; s + 3 * kApplyGroups < a.nslabs; s += 4 * kApplyGroups) {
            float4 v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = *reinterpret_cast<const float4*>(base + (size_t)(s + kApplyGroups * k) * stride);
#pragma unroll
            for (int k = 0; k < 4; ++k) { acc.x += v[k].x; acc.y += v[k].y; acc.z += v[k].z; acc.w += v[k].w; }
        }
        for (; s < a.nslabs; s += kApplyGroups) {
            const float4 v = *reinterpret_cast<const float4*>(base + (size_t)s * stride);
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
    }
    part[g][l] = acc;
    __syncthreads();
    const float step_size = adam_sh[1], bc2_sqrt = adam_sh[2];
    if (finisher) {
        float4 t = part[0][l];
        for (int w = 1; w < kApplyGroups; ++w) { const float4 v = part[w][l]; t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w; }
        const float gv[4] = {a.mode == 2 ? gin[0] : t.x, a.mode == 2 ? gin[1] : t.y, a.mode == 2 ? gin[2] : t.z, a.mode == 2 ? gin[3] : t.w};
        if (scalar_word) {
            if (a.loss && a.mode != 2) {
                const float val = (a.loss_slot == 0 ? -gv[0] : gv[0]) * a.inv_B;
                a.loss[a.loss_slot] += val;
                a.loss[2 + a.loss_slot] = val;
            }
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (flat[k] < 0) continue;
                const long long e = flat[k];
                if (a.mode != 2) a.grad[e] = gv[k];
                if (a.mode == 1) continue;
                const float mi = pm[k] + (gv[k] - pm[k]) * (1.0f - a.b1);
                const float vi = pv[k] * a.b2 + gv[k] * gv[k] * (1.0f - a.b2);
                a.exp_avg[e] = mi;
                a.exp_avg_sq[e] = vi;
                const float pn = pp[k] - step_size * (mi / (sqrtf(vi) / bc2_sqrt + a.eps));
                a.param[e] = pn;
                if (soft) a.target[e] = pn * a.tau + pt[k] * (1.0f - a.tau);
            }
        }
    }
}

// ======================================================================================================== host side
int td3_grid(int B) {
    // PIME_TD3_GRID=<n>: tuning aid -- fewer workgroups than sample tiles, so that a workgroup runs several groups back to back
    // (the second one from a warm instruction cache); production: one group per workgroup up to kTd3MaxSlabs
    static const int cap = [] {
        const char* e = std::getenv("PIME_TD3_GRID");
        const int v = e ? std::atoi(e) : 0;
        return v > 0 && v < kTd3MaxSlabs ? v : kTd3MaxSlabs;
    }();
    const int ngroups = (B + kTd3Tile - 1) / kTd3Tile;
    return ngroups < cap ? ngroups : cap;
}
int64_t td3_workspace_floats(int D, int md, int B) {
    const int64_t g = td3_grid(B);
    return g * (td3_actor_slab(D, md).stride + td3_critic_slab(D, md).stride) + (int64_t)2 * B * 8;   // slabs + the gathered rows [2][B][8] (by row parity)
}
bool td3_supported(int D, int A, int md) { return A == 1 && D >= 1 && D <= kTd3MaxD && (md == 64 || md == 128); }

// Waves per workgroup of the gradient kernels at width 128: 4 = one wave per SIMD owning two of a layer's eight output tiles,
// 8 = two waves per SIMD owning one tile each (the non-MFMA instructions of one wave issue behind the other's MFMAs).
// Width 64 has four output tiles: four waves.  PIME_TD3_WAVES=4|8 is the A/B knob.
static int td3_waves(int md) {
    static const int w = [] {
        const char* e = std::getenv("PIME_TD3_WAVES");
        const int v = e ? std::atoi(e) : kTd3DefaultWaves;
        return v == 8 ? 8 : 4;
    }();
    return md == 128 ? w : 4;
}
template <int MD, int DD, int NW>
static int launch_grad_w(bool critic, const Td3GradArgs& a, int grid, hipStream_t s) {
    const size_t lds_bytes = sizeof(float) * (size_t)td3_lds(MD / 16, a.D).total;
    if (critic) hipLaunchKernelGGL((td3_critic_kernel<MD, DD, NW>), dim3(grid), dim3(NW * 64), lds_bytes, s, a);
    else hipLaunchKernelGGL((td3_actor_kernel<MD, DD, NW>), dim3(grid), dim3(NW * 64), lds_bytes, s, a);
    PIME_HIP_TRY(hipGetLastError());
    return PIME_OK;
}
template <int MD, int DD>
static int launch_grad_d(bool critic, const Td3GradArgs& a, int grid, hipStream_t s) {
    if constexpr (MD == 128) {
        if (td3_waves(MD) == 8) return launch_grad_w<MD, DD, 8>(critic, a, grid, s);
    }
    return launch_grad_w<MD, DD, 4>(critic, a, grid, s);
}
template <int MD>
static int launch_grad(bool critic, const Td3GradArgs& a, int grid, hipStream_t s) {
    if (a.D == 3) return launch_grad_d<MD, 3>(critic, a, grid, s);   // pH observation
    if (a.D == 4) return launch_grad_d<MD, 4>(critic, a, grid, s);   // water-tank Integrator observation
    return launch_grad_d<MD, 0>(critic, a, grid, s);
}
int launch_td3_grad(bool critic, int md, const Td3GradArgs& a, int grid, hipStream_t s) {
    if (md == 128) return launch_grad<128>(critic, a, grid, s);
    if (md == 64) return launch_grad<64>(critic, a, grid, s);
    set_error("no fused TD3 instantiation for width %d", md);
    return PIME_ERR_ARG;
}
int launch_td3_apply(const Td3ApplyArgs& a, hipStream_t s) {
    const int nwords = a.L.stride / 4;
    hipLaunchKernelGGL(td3_apply_kernel, dim3((nwords + kApplyWords - 1) / kApplyWords), dim3(kApplyThreads), 0, s, a);
    PIME_HIP_TRY(hipGetLastError());
    return PIME_OK;
}

}  // namespace pime
