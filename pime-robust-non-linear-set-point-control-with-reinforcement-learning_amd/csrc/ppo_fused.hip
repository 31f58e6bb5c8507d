// PPO minibatch gradients, one launch per net: forward, loss gradient, backward chain AND the weight gradients of a
// 256-sample group are formed by one 8-wave workgroup without the dZ tensors ever leaving the CU.
//
// replaces (reference, /root/reference/elegantrl/agent.py:629-657): minibatch gather, compute_logprob
// (net_residual.py:48-54,182-190), clipped surrogate + entropy proxy (:637-645), CriticAdv forward + SmoothL1
// (:648-649) and `obj_united.backward()` (:654-655) for one net.
//
// Why: the split design (ppo_train.hip) stashes every dZ_l and H_l in HBM (2.5-3 KB per sample and net) and re-reads
// them in a separate TN-GEMM kernel; profiles/r01_c_pmc_hbm_traffic.json shows ~1 GB of HBM traffic per minibatch and
// the matrix pipe 33-47 % busy, i.e. the pipeline sits on the MFMA/HBM ridge.  Here only ONE hidden activation per
// sample and net (1 KB) round-trips through HBM/L2, read back by the workgroup that wrote it (0.5 GB per minibatch,
// profiles/r01_h_pmc_hbm_traffic.json).
//
// Workgroup = 8 waves = 8 tiles of 32 samples.  LDS map: S (first-layer images, biases, head, the group's states,
// per-wave partial sums), W (one md x md MFMA image), X (two sample-major transposition buffers; during the forward it
// holds the second MFMA image).  The last hidden layer stays in registers across the loss: its dZ is formed in place and
// the head weight gradient is a DPP lane reduction.  Per earlier layer l of the backward:
//   dW_l = dZ_l^T H_{l-1}: eight rounds, round t multiplies tile t (dw_rounds).  The owning wave publishes its dZ_l tile,
//          ALL waves publish 1/8 of the matching H_{l-1} tile each (from the stash -- first-layer activations included since
//          round 3 --, fetched two rounds ahead, FEATURE-major: dw_rounds); every wave owns 1/8 of the output blocks
//          in accumulators for the eight rounds.  X is double buffered (one LDS-only barrier per round); the next
//          layer's transposed weight image is copied into W one slice per round, behind the MFMAs.  The bias gradient
//          is the sum of the A operands a wave reads anyway.
//   dH_{l-1} = W_l^T dZ_l: the forward chain code with the transposed image (mlp_device.hpp).
// First-layer gradients (fan-in of a few floats) run on the vector ALUs from a feature-major, XOR-swizzled image (first_grad_valu).
// The forward images arrive by LDS-DMA, each layer waiting for its own; the critic runs its first dH step in front of the last
// layer's rounds with H2 kept in registers (DX_FIRST).  f32 MFMAs share the SIMD's issue slots with every vector / LDS instruction
// (tools/dw_round_bench.hip): what counts in the MFMA phases is the instruction count, not latency hiding.
// Every workgroup stores its partial gradient into its own slab (no atomics: memory-side float atomics cost 0.5 TB/s
// here); ppo_grad_reduce_kernel sums the slabs in slab order, so gradients are reproducible bit for bit, applies Adam to the
// element it has just reduced and writes the new value into the packed images (pime_ppo_minibatch_step, pime_ppo_image_map).
// PIME_FUSED_TRACE=<workgroup> prints wall-clock phase marks of that workgroup (tuning aid, not a production path).
#include "ppo_device.hpp"
#include "ppo_train.hpp"

#include <cstdlib>
#include <type_traits>

namespace pime {

constexpr int kFusedThreads = 512;
constexpr int kFusedWaves = kFusedThreads / 64;
// Transposition buffers are SAMPLE-major: row s = the NT*32 features of sample s, padded by 4 floats.  A lane (sample)
// writes its accumulator registers four features at a time (ds_write_b128; registers 4g..4g+3 of a tile are four
// consecutive features), the MFMA operand reads take consecutive features on consecutive lanes: both conflict-free.
__host__ __device__ constexpr int tpitch(int nt) { return nt * 32 + 4; }
__host__ __device__ constexpr int tsize(int nt) { return 32 * tpitch(nt); }   // floats of one 32-sample tile image
// The B operand of a weight-gradient round (the activation tile) is FEATURE-major instead: row = one feature's 32 samples, even
// samples in columns 0..15, odd ones in 16..31, pitch 36 -- a lane's operands of the 16 k-steps (samples 2s + h of feature li) are 16
// consecutive floats = four ds_read_b128 per output block instead of sixteen ds_read_b32 (LDS instructions cost SIMD issue slots,
// not bytes: tools/dw_round_bench.hip); the eight waves publish their slices as ds_write_b32 (lanes = samples: conflict-free).
constexpr int kBPitch = 36;
__host__ __device__ constexpr int bsize(int nt) { return nt * 32 * kBPitch; }

// first-layer gradients on the vector ALUs up to this fan-in (first_grad_valu); its state rows are padded to 4 / 8 floats
constexpr int kFirstValuMaxD = 7;
__host__ __device__ constexpr int first_valu_pad(int D) { return D <= 3 ? 4 : (D <= kFirstValuMaxD ? 8 : 0); }

struct FusedLds {
    int first0, first1, bias[3], headw, headb, xs, xsp, hacc, wsum, wbuf, x, total;
};

__host__ __device__ inline FusedLds fused_lds(int kind, int D, int Di, int T) {
    FusedLds F{};
    int o = 0;
    auto seg = [&](int& f, int floats) { f = o; o = align4(o + floats); };
    const int md = T * 32;
    // Everything whose size depends on the state width comes LAST: the offsets in front are then compile-time constants in the
    // kernels (T and the kind are template parameters) -- immediates in the ds instructions instead of live scalars.
    if (kind == MLP_MODULAR_ACTOR) {
        const int H = T / 2;
        seg(F.bias[0], H * 32);
        seg(F.bias[1], H * 32);
        seg(F.bias[2], md);
    } else {
        seg(F.bias[0], md);
        seg(F.bias[1], md);
        F.bias[2] = 0;
    }
    seg(F.headw, md);
    seg(F.headb, 4);
    seg(F.hacc, kFusedWaves * md);     // per-wave head weight gradients (summed in wave order at the end)
    seg(F.wsum, kFusedWaves * 6 * 2);  // per-wave float64 totals of the scalar sums
    seg(F.wbuf, T * T * 1024);
    // X: two (A+B) buffers of the widest job that keeps W alive (2T blocks each); the merged modular jobs (3T and 2T+1
    // blocks per buffer) run while W is dead and use W and X as one region
    int xf = 2 * (tsize(T) + bsize(T));
    const int need3 = 2 * (tsize(T) + bsize(2 * T)) - T * T * 1024, need1 = 2 * (tsize(2 * T) + bsize(1)) - T * T * 1024;
    if (kind == MLP_MODULAR_ACTOR) xf = xf > need3 ? (xf > need1 ? xf : need1) : (need3 > need1 ? need3 : need1);
    seg(F.x, xf);
    seg(F.xsp, kFusedWaves * 32 * first_valu_pad(D));   // the group's rows padded to 4 / 8 floats: (x, 1, 0 ..) for first_grad_valu
    seg(F.xs, kFusedWaves * 32 * D);   // the group's gathered states [wave][sample][D]
    if (kind == MLP_MODULAR_ACTOR) {
        const int Do = D - Di;
        seg(F.first0, md * (Do + 1));
        seg(F.first1, md * (Di + 1));
    } else {
        seg(F.first0, md * (D + 1));
        F.first1 = 0;
    }
    F.total = o;
    return F;
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for every outstanding global operation
// (vmcnt(0)), i.e. for the gradient atomics of the previous job to come back from L2 - 10-15 us per job when all 256
// workgroups flush the same tensor.  The rounds only exchange data through LDS.
#define PIME_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

// A forward image goes global -> LDS by LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write; a wave
// instruction lands 1 KB at a wave-uniform LDS base + lane * 16): issued at the top of a group, waited for (counted vmcnt) only in
// front of the layer that reads it, so the first layer(s) run while the images of the later ones are still in flight.
// NI = 8 KB pieces of the image (one DMA instruction per wave and piece).  The source base is made provably wave-uniform
// (readfirstlane): a per-piece 64-bit VGPR address would be spilled, and every reload's vmcnt(0) would drain the DMAs.
typedef __attribute__((address_space(3))) void* lds_void_ptr;
typedef const __attribute__((address_space(1))) void* glb_void_ptr;
template <int NI>
__device__ __forceinline__ void dma_image(float* __restrict__ lds_dst, const float* __restrict__ src, int tid) {
    const unsigned long long u = reinterpret_cast<unsigned long long>(src);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
    const char* sbase = reinterpret_cast<const char*>(((unsigned long long)hi << 32) | lo);
    const unsigned voff = (unsigned)tid * 16u;
    float* ldst = lds_dst + (tid >> 6) * 256;   // this wave's 1 KB inside every 8 KB piece
#pragma unroll
    for (int p = 0; p < NI; ++p)
        __builtin_amdgcn_global_load_lds((glb_void_ptr)(sbase + p * (kFusedThreads * 16) + voff),
                                         (lds_void_ptr)(ldst + p * kFusedThreads * 4), 16, 0, 0);
}
template <int N>
__device__ __forceinline__ void wait_dma_then_barrier() {   // all but the N youngest vector-memory operations of this wave are done
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory");
}

// Accumulator-layout tile (sample on the lane) -> feature-major rows [t*32 + feat][sample], pitch 33.
template <int NTL>
__device__ __forceinline__ void put_tile(float* __restrict__ dst, int lane, const f32x16 (&v)[NTL]) {
    float* p = dst + (lane & 31) * tpitch(NTL) + 4 * (lane >> 5);
#pragma unroll
    for (int t = 0; t < NTL; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g)
            *reinterpret_cast<float4*>(p + t * 32 + 8 * g) = make_float4(v[t][4 * g], v[t][4 * g + 1], v[t][4 * g + 2], v[t][4 * g + 3]);
}

// Stash tiles through a wave-uniform base: one 32-bit lane offset + immediates instead of a 64-bit address per row
// (which the compiler hoists out of the group loop and spills).  Layout of a tile: [t][g = r >> 2][lane][4] -- a lane's four
// consecutive accumulator registers are ONE 16-byte word, a wave instruction moves 1 KB contiguous: 16 vector-memory
// instructions per activation instead of 64 (round 3; a vector-memory instruction costs ~10 issue cycles on the SIMD).
template <int NTL>
__device__ __forceinline__ void stash_put(float* __restrict__ base, int lane, const f32x16 (&a)[NTL]) {
    int off = lane * 4;
    asm volatile("" : "+v"(off));
#pragma unroll
    for (int t = 0; t < NTL; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g)
            *reinterpret_cast<float4*>(base + (t * 4 + g) * 256 + off) = make_float4(a[t][4 * g], a[t][4 * g + 1], a[t][4 * g + 2], a[t][4 * g + 3]);
}
template <int NTL>
__device__ __forceinline__ void stash_get(const float* __restrict__ base, int lane, f32x16 (&a)[NTL]) {
    int off = lane * 4;
    asm volatile("" : "+v"(off));
#pragma unroll
    for (int t = 0; t < NTL; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 v = *reinterpret_cast<const float4*>(base + (t * 4 + g) * 256 + off);
            a[t][4 * g] = v.x; a[t][4 * g + 1] = v.y; a[t][4 * g + 2] = v.z; a[t][4 * g + 3] = v.w;
        }
}
// NP consecutive registers r0 .. r0 + NP - 1 (r0 and NP multiples of 4) of tile t of a stashed activation: NP / 4 16-byte loads
template <int NP>
__device__ __forceinline__ void stash_fetch(const float* __restrict__ tile, int t, int r0, int lane, float (&v)[NP]) {
    static_assert(NP % 4 == 0, "whole 16-byte words");
    const float* p = tile + (t * 4 + (r0 >> 2)) * 256 + lane * 4;
#pragma unroll
    for (int q = 0; q < NP / 4; ++q) {
        const float4 w = *reinterpret_cast<const float4*>(p + q * 256);
        v[4 * q] = w.x; v[4 * q + 1] = w.y; v[4 * q + 2] = w.z; v[4 * q + 3] = w.w;
    }
}

// ---- B-operand sources: NP consecutive registers r0..r0+NP-1 of tile t of the accumulator-layout activation tile of
// wave `ow`, for this lane.  Any wave can produce any tile's elements, so the eight waves share the publishing work.
// (Rounds 1-2 also had sources that RECOMPUTED a first-layer activation from the group's states; since round 3 those
// activations come from the stash like every other.)
struct StashB {   // a hidden activation stashed by the forward: [wave tile][t][4][64][4] (stash_put)
    const float* base;   // stash of the group's first tile, at the wanted activation
    int tile_stride;     // floats between consecutive tiles
    template <int NP>
    __device__ __forceinline__ void fetch(int ow, int t, int r0, int lane, float (&v)[NP]) const {
        stash_fetch<NP>(base + (size_t)ow * tile_stride, t, r0, lane, v);
    }
};
struct StateB {   // the raw state columns [col0, col0+Din) padded with zeros to one 32-feature tile
    const float* xs;
    int Din, D, col0;
    template <int NP>
    __device__ __forceinline__ void fetch(int ow, int, int r0, int lane, float (&v)[NP]) const {
        const float* x = xs + (ow * 32 + (lane & 31)) * D + col0;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int f = feat32(r0 + i, lane >> 5);
            v[i] = f < Din ? x[f] : 0.f;
        }
    }
};

// Work split of a (AT x BT)-block weight gradient over the eight waves.
template <int AT, int BT>
struct DwPlan {
    static constexpr int NBLK = AT * BT;
    static constexpr int PER = NBLK >= kFusedWaves ? NBLK / kFusedWaves : 1;       // output blocks per wave
    static constexpr int NKS = 16;
    static_assert(BT % PER == 0, "a wave's blocks must share the A tile");
    int ao, bi0;
    bool active;  // with fewer blocks than waves the spare waves only help publishing (every output element has ONE
                  // owner lane, so the partial gradients can be plain stores)
    bool bias;    // this wave stores the bias gradient of its A tile (the first of the waves that share the tile)
    __device__ __forceinline__ explicit DwPlan(int wave) {
        const int b0 = wave * PER;
        active = b0 < NBLK;
        ao = active ? b0 / BT : 0; bi0 = active ? b0 % BT : 0;
        bias = active && bi0 == 0;
    }
};

// acc[n] (+)= sum over the group's 256 samples of A[s][a-block ao] (x) B[s][b-block bi0+n];  bsum = sum_s A[s][ao*32 + li]
// (this lane's k parity).  az: this wave's A tile (AT feature tiles, accumulator layout), published by the wave
// itself; the B tile of every round is produced by all eight waves (BT*2 elements per lane each, fetched two rounds
// ahead).  While the rounds run, stage_n4 float4s are copied from stage_src into LDS at stage_dst.
template <int AT, int BT, class BSrc, class Plan = DwPlan<AT, BT>>
__device__ __forceinline__ void dw_rounds(float* __restrict__ X, int lane, int wave, const f32x16 (&az)[AT],
                                          const BSrc& bsrc, f32x16 (&acc)[Plan::PER], float& bsum,
                                          float* __restrict__ stage_dst, const float* __restrict__ stage_src,
                                          int stage_n4, long long* tr = nullptr) {
    constexpr int PER = Plan::PER, NKS = 16;
    constexpr int BUF = tsize(AT) + bsize(BT);
    constexpr int NP = BT * 16 / kFusedWaves;                               // B elements this wave publishes per round
    const Plan pl(wave);
    const int tid = wave * 64 + lane, h = lane >> 5, li = lane & 31;
    const int pt = (wave * NP) / 16, pr0 = (wave * NP) % 16;                // published elements: tile pt, regs pr0..pr0+NP
#pragma unroll
    for (int n = 0; n < PER; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
    f32x2_t bs2 = {0.f, 0.f};   // bias gradient: the A operands summed, two k-steps per v_pk_add_f32, on the wave that stores it only
    const int per4 = stage_n4 / kFusedWaves;
    const float4* src4 = reinterpret_cast<const float4*>(stage_src);
    float4* dst4 = reinterpret_cast<float4*>(stage_dst);
    float pv[NP], pv2[NP];
    auto b_fetch = [&](int ow, float (&v)[NP]) { bsrc.template fetch<NP>(ow, pt, pr0, lane, v); };
    auto b_publish = [&](float* buf) {   // register pr0 + i of tile pt = feature pt * 32 + 4 h + 8 ((pr0 + i) >> 2) + ((pr0 + i) & 3)
        float* p = buf + tsize(AT) + (pt * 32 + 4 * h) * kBPitch + (li & 1) * 16 + (li >> 1);
#pragma unroll
        for (int i = 0; i < NP; ++i) p[(8 * ((pr0 + i) >> 2) + ((pr0 + i) & 3)) * kBPitch] = pv[i];
    };

    unsigned long long tsum[5] = {0, 0, 0, 0, 0}, tprev = 0;
    const bool stamping = tr != nullptr && wave == 0;
#define PIME_STAMP(k)                                                                           \
    if (stamping) {                                                                             \
        unsigned long long tnow;                                                                \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tnow)::"memory");            \
        tsum[k] += tnow - tprev;                                                                \
        tprev = tnow;                                                                           \
    }
    PIME_LDS_BARRIER();  // X free
    b_fetch(0, pv);
    if (wave == 0) put_tile<AT>(X, lane, az);
    b_publish(X);
    b_fetch(1, pv);
#pragma unroll 1
    for (int t = 0; t < kFusedWaves; ++t) {
        PIME_LDS_BARRIER();  // buffer t&1 published; buffer (t+1)&1 no longer read
        if (tr && tid == 0) tr[t] = wall_clock64();
        PIME_STAMP(4);   // barrier wait (the first one also absorbs the time before the loop)
        const float* cur = X + (t & 1) * BUF;
        float* nxt = X + ((t + 1) & 1) * BUF;
        float4 sv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (tid < per4) sv = src4[t * per4 + tid];
        // B elements for round t+2: global loads (or a few LDS reads), issued now, landed behind this round's MFMAs
        if (t + 2 < kFusedWaves) b_fetch(t + 2, pv2);
        if (wave == t + 1) put_tile<AT>(nxt, lane, az);
        PIME_STAMP(0);   // fetch (+ A publish on the owner)
        PIME_NO_HOIST();
        if (pl.active) {
            const float* Ap = cur + h * tpitch(AT) + pl.ao * 32 + li;                  // k-step s: samples 2s + h
            const float* Bp = cur + tsize(AT) + (pl.bi0 * 32 + li) * kBPitch + h * 16;
            float av[NKS], bv[PER][NKS];
#pragma unroll
            for (int s = 0; s < NKS; ++s) av[s] = Ap[2 * s * tpitch(AT)];
#pragma unroll
            for (int n = 0; n < PER; ++n)
#pragma unroll
                for (int q = 0; q < NKS / 4; ++q) {
                    const float4 b4 = *reinterpret_cast<const float4*>(Bp + n * 32 * kBPitch + 4 * q);
                    bv[n][4 * q] = b4.x; bv[n][4 * q + 1] = b4.y; bv[n][4 * q + 2] = b4.z; bv[n][4 * q + 3] = b4.w;
                }
            PIME_STAMP(1);   // operand reads back
#pragma unroll
            for (int s = 0; s < NKS; ++s) {
#pragma unroll
                for (int n = 0; n < PER; ++n)
                    acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], bv[n][s], acc[n], 0, 0, 0);
            }
            if (pl.bias) {
#pragma unroll
                for (int s = 0; s < NKS; s += 2) bs2 += f32x2_t{av[s], av[s + 1]};
            }
            if (stamping) {
                float sink;
                asm volatile("v_mov_b32 %0, %1" : "=v"(sink) : "v"(acc[PER - 1][0]));   // waits for the last MFMA
            }
            PIME_STAMP(2);   // MFMAs
        }
        PIME_NO_HOIST();
        if (t + 1 < kFusedWaves) b_publish(nxt);
#pragma unroll
        for (int i = 0; i < NP; ++i) pv[i] = pv2[i];
        if (tid < per4) dst4[t * per4 + tid] = sv;
        PIME_STAMP(3);   // B publish, staging write
    }
#undef PIME_STAMP
    bsum = bs2.x + bs2.y;
    if (tr && tid == 0) {
        tr[8] = wall_clock64();
        for (int k = 0; k < 5; ++k) tr[16 + k] = (long long)tsum[k];
    }
}

// Element offset of accumulator register 0 of a wave's first output block (D[i = a feature][j = b feature]: column on
// the lane, rows in the registers).  Made opaque: otherwise every row address of every job is hoisted to the kernel
// prologue as a 64-bit pointer and spilled.
template <int AT, int BT>
__device__ __forceinline__ int dw_base(int lane, int wave, int ldw) {
    const DwPlan<AT, BT> pl(wave);
    int base = (pl.ao * 32 + 4 * (lane >> 5)) * ldw + pl.bi0 * 32 + (lane & 31);
    asm volatile("" : "+v"(base));
    return base;
}

// Writes a finished job into the workgroup's gradient slab (plain stores: every element has one owner lane).  Rows >=
// nrows and columns >= ncols of the product are dropped; accum: add to what an earlier sample group of this workgroup
// stored.  gb: slab position of the bias gradient (sum of the A operands).
template <int AT, int BT>
__device__ __forceinline__ void dw_store(int lane, int wave, const f32x16 (&acc)[DwPlan<AT, BT>::PER], float bsum,
                                         float* __restrict__ gW, int ldw, int nrows, int ncols, float* __restrict__ gb,
                                         bool accum) {
    const DwPlan<AT, BT> pl(wave);
    if (!pl.active) return;
    const int h = lane >> 5, li = lane & 31;
    const int base = dw_base<AT, BT>(lane, wave, ldw);
#pragma unroll
    for (int n = 0; n < DwPlan<AT, BT>::PER; ++n)
        if ((pl.bi0 + n) * 32 + li < ncols) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (pl.ao * 32 + feat32(r, h) < nrows) {
                    float* q = &gW[base + n * 32 + ((r & 3) + 8 * (r >> 2)) * ldw];
                    *q = accum ? *q + acc[n][r] : acc[n][r];
                }
        }
    bsum += __shfl_xor(bsum, 32);  // the two k parities
    if (gb && pl.bi0 == 0 && h == 0) {
        float* q = &gb[pl.ao * 32 + li];
        *q = accum ? *q + bsum : bsum;
    }
}

// The same for a job whose blocks are all inside the tensor (the md x md layers): row pitch and block origin are compile-time, no
// per-element bounds tests, and the accumulate / overwrite decision is taken once (dw_store spends ~3 vector + 5 scalar
// instructions and two branches per element on them: with f32 MFMAs every one of those is time on the SIMD).
template <int AT, int BT>
__device__ __forceinline__ void dw_store_full(int lane, int wave, const f32x16 (&acc)[DwPlan<AT, BT>::PER], float bsum,
                                              float* __restrict__ gW, float* __restrict__ gb, bool accum) {
    constexpr int LDW = BT * 32, PER = DwPlan<AT, BT>::PER;
    const DwPlan<AT, BT> pl(wave);
    if (!pl.active) return;   // fewer blocks than waves (width 64)
    float* const q = gW + dw_base<AT, BT>(lane, wave, LDW);
    if (!accum) {
#pragma unroll
        for (int n = 0; n < PER; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) q[n * 32 + ((r & 3) + 8 * (r >> 2)) * LDW] = acc[n][r];
    } else {
#pragma unroll
        for (int n = 0; n < PER; ++n) {
            f32x16 old;
#pragma unroll
            for (int r = 0; r < 16; ++r) old[r] = q[n * 32 + ((r & 3) + 8 * (r >> 2)) * LDW];
#pragma unroll
            for (int r = 0; r < 16; ++r) q[n * 32 + ((r & 3) + 8 * (r >> 2)) * LDW] = old[r] + acc[n][r];
        }
    }
    bsum += __shfl_xor(bsum, 32);  // the two k parities
    if (pl.bi0 == 0 && lane < 32) {
        float* qb = &gb[pl.ao * 32 + lane];
        *qb = accum ? *qb + bsum : bsum;
    }
}

// ---- modular actor: the two branch layers of a level share one set of rounds ---------------------------------------
// Level 2 (other_net.2 | integrator_net.2): A = [dZo2 | dZi2] (T tiles), B = [h_o1 | h_i1] (2T tiles).  Only the
// block-diagonal products are wanted: A tiles [0,H) x B tiles [0,T) and A tiles [H,T) x B tiles [T,2T).
template <int T>
struct CatPlan {
    static constexpr int H = T / 2 > 0 ? T / 2 : 1;
    static constexpr int NBLK = 2 * H * T;
    static constexpr int PER = NBLK >= kFusedWaves ? NBLK / kFusedWaves : 1;
    int ao, bi0;
    bool active, bias;
    __device__ __forceinline__ explicit CatPlan(int wave) {
        const int b0 = wave * PER;
        active = b0 < NBLK;
        const int branch = active ? b0 / (H * T) : 0, within = active ? b0 % (H * T) : 0;
        ao = branch * H + within / T;
        bi0 = branch * T + within % T;
        bias = active && bi0 % T == 0;
    }
};
template <int T>
struct CatStashB {   // tiles [0,T): other_net's first-layer activation (stash region 1), [T,2T): integrator_net's (region 2)
    const float *o, *i;   // the group's first tile in either region
    int tile_stride;
    template <int NP>
    __device__ __forceinline__ void fetch(int ow, int t, int r0, int lane, float (&v)[NP]) const {
        stash_fetch<NP>((t < T ? o : i) + (size_t)ow * tile_stride, t < T ? t : t - T, r0, lane, v);
    }
};

// One 32x32 accumulator block -> rows row0.. of a row-major [.. x ldw] slab tensor at column `col` (this lane's).  The
// accumulate / overwrite decision is wave-uniform: taken once, not per element.
__device__ __forceinline__ void block_store(const f32x16& acc, float* __restrict__ gW, int ldw, int row0, int col,
                                            bool col_ok, int lane, bool accum) {
    int base = (row0 + 4 * (lane >> 5)) * ldw + col;
    asm volatile("" : "+v"(base));
    if (!col_ok) return;
    float* const q = gW + base;
    if (!accum) {
#pragma unroll
        for (int r = 0; r < 16; ++r) q[((r & 3) + 8 * (r >> 2)) * ldw] = acc[r];
    } else {
        f32x16 old;
#pragma unroll
        for (int r = 0; r < 16; ++r) old[r] = q[((r & 3) + 8 * (r >> 2)) * ldw];
#pragma unroll
        for (int r = 0; r < 16; ++r) q[((r & 3) + 8 * (r >> 2)) * ldw] = old[r] + acc[r];
    }
}
__device__ __forceinline__ void bias_store(float bsum, float* __restrict__ gb, int lane, bool accum) {
    bsum += __shfl_xor(bsum, 32);  // the two k parities
    if (lane < 32) {
        float* q = &gb[lane];
        *q = accum ? *q + bsum : bsum;
    }
}

// First-layer weight / bias gradient for a fan-in of a few columns, on the vector ALUs:
//   gW[f][j] = sum over the group's 256 samples of dZ[s][f] * x[s][j],   gb[f] = sum_s dZ[s][f].
// As a matrix job this is a 32-column product of which Din columns are real (90 % wasted MFMA work in eight
// barrier-separated rounds).  f32 MFMAs and vector instructions share the SIMD's issue slots (tools/dw_round_bench.hip), so
// what counts is the instruction count: two passes of 128 samples; the pass's four waves write their dZ tiles FEATURE-major
// ([feature][128 samples], 16-byte groups XOR-swizzled with the feature so that both sides are conflict-free), thread
// (feature f, slice q) then reads its samples four at a time (ds_read_b128) and the padded state rows xsp[s] = (x_0 .. x_{D-1},
// 1, 0 ..) as one or two broadcast ds_read_b128: per sample 1/4 + 1 LDS instructions and P fmas for ALL columns and the
// bias at once.  The Q slices are combined in a fixed order (reproducible).
template <int T>
__device__ __forceinline__ void first_grad_valu(float* __restrict__ X, int lane, int wave, const f32x16 (&dz)[T],
                                                const float* __restrict__ xsp, int D, int col0, int Din,
                                                float* __restrict__ gW, float* __restrict__ gb, bool accum) {
    constexpr int R = T * 32, Q = kFusedThreads / R, SP = 128, GPT = SP / Q / 4;   // groups of 4 samples per thread and pass
    static_assert(2 * SP == kFusedWaves * 32 && SP * R <= 2 * (tsize(T) + tsize(T)), "two passes over the X region");
    const int tid = wave * 64 + lane, f = tid % R, q = tid / R, h = lane >> 5, li = lane & 31;
    const int P = first_valu_pad(D);
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
#pragma unroll 1
    for (int pass = 0; pass < 2; ++pass) {
        PIME_LDS_BARRIER();  // X free
        if ((wave >> 2) == pass) {
            const int s = (wave & 3) * 32 + li;
            float* p = X + (4 * h) * SP + (s & 3);
            const int g = s >> 2;
#pragma unroll
            for (int t = 0; t < T; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {   // feature t*32 + (r&3) + 8*(r>>2) + 4h; feature & 7 = (r&3) + 4h
                    const int fl = (r & 3) + 8 * (r >> 2);
                    p[(t * 32 + fl) * SP + ((g ^ ((r & 3) + 4 * h)) << 2)] = dz[t][r];
                }
        }
        PIME_LDS_BARRIER();
        const float* row = X + f * SP;
        const float* xr = xsp + (pass * SP + q * (GPT * 4)) * P;
#pragma unroll
        for (int g = 0; g < GPT; ++g) {
            const float4 d4 = *reinterpret_cast<const float4*>(row + (((q * GPT + g) ^ (f & 7)) << 2));
            const float dk[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float4 x4 = *reinterpret_cast<const float4*>(xr + (g * 4 + k) * P);
                acc[0] = fmaf(dk[k], x4.x, acc[0]); acc[1] = fmaf(dk[k], x4.y, acc[1]);
                acc[2] = fmaf(dk[k], x4.z, acc[2]); acc[3] = fmaf(dk[k], x4.w, acc[3]);
                if (P == 8) {
                    const float4 y4 = *reinterpret_cast<const float4*>(xr + (g * 4 + k) * P + 4);
                    acc[4] = fmaf(dk[k], y4.x, acc[4]); acc[5] = fmaf(dk[k], y4.y, acc[5]);
                    acc[6] = fmaf(dk[k], y4.z, acc[6]); acc[7] = fmaf(dk[k], y4.w, acc[7]);
                }
            }
        }
    }
    PIME_LDS_BARRIER();  // the dZ image is dead: its first 8 * Q * R floats hold the slices' partial sums
#pragma unroll
    for (int j = 0; j < 8; ++j)
        if (j < P) X[(q * 8 + j) * R + f] = acc[j];
    PIME_LDS_BARRIER();
    for (int e = tid; e < R * (Din + 1); e += kFusedThreads) {
        const int j = e / R, ff = e % R, col = j < Din ? col0 + j : D;   // column D of the padded rows is the constant 1: the bias
        float t = 0.f;
        for (int p = 0; p < Q; ++p) t += X[(p * 8 + col) * R + ff];
        float* qd = j < Din ? &gW[ff * Din + j] : &gb[ff];
        *qd = accum ? *qd + t : t;
    }
}

// Sums over the 32 lanes of a half for SIXTEEN registers at once (the head weight gradient: one register = one feature, a lane =
// one sample).  Sixteen separate DPP ladders cost 80 vector instructions; here every level adds a PAIR of registers into one, each
// lane group keeping the partner's sum of the register it "owns", so the register count halves from level to level:
//   rows (16 lanes):   v_permlane16_swap + add                 16 -> 8 registers, 2 instructions per pair
//   8-lane groups:     x + ror8(x) for both, masked dpp select  8 -> 4, 3 per pair
//   banks (4 lanes):   x + half_mirror(x), masked dpp select    4 -> 2, 3 per pair
//   quad:              two quad_perm adds on the last 2 registers
// = 38 instructions.  Result: out[k] (k = 0, 1) in lane li holds the sum over the half's 32 lanes of register
// 8 k + 4 bit2(li) + 2 bit3(li) + bit4(li), the same value in the four lanes of a quad.  Fixed order: reproducible.
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
template <int BANK_MASK>
__device__ __forceinline__ float dpp_select(float old, float v) {   // v in the banks of the mask, old elsewhere
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), 0xe4, 0xf, BANK_MASK, false));
}
template <int CTRL>
__device__ __forceinline__ float dpp_add_full(float v) {
    return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ void half_sums16(const float (&p)[16], float (&out)[2]) {
    float a8[8], a4[4];
#pragma unroll
    for (int i = 0; i < 8; ++i) {   // rows 0 / 2 keep p[2i]'s sum over the row pair, rows 1 / 3 p[2i+1]'s
        const u32x2_t sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(p[2 * i]), __float_as_uint(p[2 * i + 1]), false, false);
        a8[i] = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)     // lanes 0-7 of a row: a8[2i] + its lane 8 apart; lanes 8-15: a8[2i+1]
        a4[i] = dpp_select<0xc>(dpp_add_full<0x128>(a8[2 * i]), dpp_add_full<0x128>(a8[2 * i + 1]));
#pragma unroll
    for (int i = 0; i < 2; ++i) {   // banks 0 / 2: a4[2i] + its mirror inside the 8 lanes; banks 1 / 3: a4[2i+1]
        float v = dpp_select<0xa>(dpp_add_full<0x141>(a4[2 * i]), dpp_add_full<0x141>(a4[2 * i + 1]));
        v = dpp_add_full<0xb1>(v);  // quad_perm [1,0,3,2]
        out[i] = dpp_add_full<0x4e>(v);  // quad_perm [2,3,0,1]
    }
}

// TRACE = false compiles the marks out (the dual kernel: the trace pointers and the traced workgroup's index are live scalars of the
// whole body otherwise)
#define PIME_MARK(i)                                                             \
    do {                                                                         \
        if constexpr (TRACE)                                                     \
            if (a.trace && bid == a.trace_wg && threadIdx.x == 0) a.trace[i] = wall_clock64(); \
    } while (0)

// bid / nb: this workgroup's index among the nb workgroups that work on THIS net (ppo_fused_kernel: the grid; ppo_fused_dual_kernel:
// the net's share of a grid that serves both nets)
template <int T, int KIND, bool TRACE>
__device__ __forceinline__ void ppo_fused_body(const PpoArgs& a, float* __restrict__ lds, const int bid, const int nb) {
    constexpr bool MODULAR = KIND == MLP_MODULAR_ACTOR;
    constexpr bool CRITIC = KIND == MLP_CRITIC;
    constexpr int ACT = CRITIC ? 0 : 1;
    constexpr int H = T / 2 > 0 ? T / 2 : 1;
    constexpr int md = T * 32;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ntiles = (a.B + 31) / 32, ngroups = (ntiles + kFusedWaves - 1) / kFusedWaves;
    const float invB = 1.0f / (float)a.B;
    const FusedLds F = fused_lds(KIND, a.D, a.Di, T);
    const MlpLayout L = mlp_layout(KIND, a.D, a.Di, md);
    const BwdLayout Lb = bwd_layout(KIND, a.D, a.Di, md);
    float* const wbuf = lds + F.wbuf;
    float* const X = lds + F.x;
    float* const xs = lds + F.xs;
    float* const xsp = lds + F.xsp;
    float* const hacc = lds + F.hacc;   // head weight gradient of the workgroup
    const int Do = a.D - a.Di;

    PIME_MARK(0);
    if constexpr (TRACE)
        if (a.trace_span && threadIdx.x == 0 && bid < 512) a.trace_span[2 * bid] = wall_clock64();
    const float asl = CRITIC ? 0.f : a.a_std_log[0];
    for (int e = tid; e < kFusedWaves * md; e += kFusedThreads) hacc[e] = 0.f;
    // small segments live in LDS for the whole kernel.  All their loads are issued before the first LDS write: one
    // L2 round trip instead of one per segment (7 segments cost the modular actor 10 us).  Named registers, not an array:
    // hipcc put a float4 v[NS] on the stack (144 B of scratch per lane; every reload waits on the shared vmcnt counter).
    {
#define PIME_SEG_LOAD(k, imgoff, floats)                                                       \
    const int n4_##k = (floats) / 4;                                                            \
    const float4* src_##k = reinterpret_cast<const float4*>(a.img_fwd + (imgoff));              \
    float4 v_##k = make_float4(0.f, 0.f, 0.f, 0.f);                                             \
    if (tid < n4_##k) v_##k = src_##k[tid];
#define PIME_SEG_STORE(k, ldsoff)                                                               \
    {                                                                                           \
        float4* dst_ = reinterpret_cast<float4*>(lds + (ldsoff));                               \
        if (tid < n4_##k) dst_[tid] = v_##k;                                                    \
        for (int i = tid + kFusedThreads; i < n4_##k; i += kFusedThreads) dst_[i] = src_##k[i]; /* wide first layers only */ \
    }
        if constexpr (MODULAR) {
            PIME_SEG_LOAD(0, L.off[0], md * (Do + 1))
            PIME_SEG_LOAD(1, L.off[3], md * (a.Di + 1))
            PIME_SEG_LOAD(2, L.off[2], H * 32)
            PIME_SEG_LOAD(3, L.off[5], H * 32)
            PIME_SEG_LOAD(4, L.off[7], md)
            PIME_SEG_LOAD(5, L.off[8], md)
            PIME_SEG_LOAD(6, L.off[9], 4)
            PIME_SEG_STORE(0, F.first0) PIME_SEG_STORE(1, F.first1) PIME_SEG_STORE(2, F.bias[0]) PIME_SEG_STORE(3, F.bias[1])
            PIME_SEG_STORE(4, F.bias[2]) PIME_SEG_STORE(5, F.headw) PIME_SEG_STORE(6, F.headb)
        } else {
            PIME_SEG_LOAD(0, L.off[0], md * (a.D + 1))
            PIME_SEG_LOAD(1, L.off[2], md)
            PIME_SEG_LOAD(2, L.off[4], md)
            PIME_SEG_LOAD(3, L.off[5], md)
            PIME_SEG_LOAD(4, L.off[6], 4)
            PIME_SEG_STORE(0, F.first0) PIME_SEG_STORE(1, F.bias[0]) PIME_SEG_STORE(2, F.bias[1]) PIME_SEG_STORE(3, F.headw)
            PIME_SEG_STORE(4, F.headb)
        }
#undef PIME_SEG_LOAD
#undef PIME_SEG_STORE
    }
    // scalar sums (losses, d/d a_std_log, head bias gradient, target moments): reduced over the wave right where they
    // are produced and kept in LDS, not in loop-carried registers (which hipcc spills around the MFMA phases)
    double* const wsum = reinterpret_cast<double*>(lds + F.wsum);   // [wave][6]
    if (tid < kFusedWaves * 6) wsum[tid] = 0.0;

#pragma unroll 1
    for (int group = bid; group < ngroups; group += nb) {
        // Lane-derived offsets are made loop-variant on purpose: hipcc otherwise hoists ~100 per-lane LDS / global
        // offsets of the whole body out of this (usually single-trip) loop and spills them.
        int lane = tid & 63;
        asm volatile("" : "+v"(lane));
        const int h = lane >> 5, li = lane & 31;
        // the minibatch gather (index -> row -> per-sample inputs) is two dependent HBM round trips: start it before
        // the weight staging so that the two overlap
        const int tile = group * kFusedWaves + wave;  // tiles past the batch run on clamped rows with dOut = 0
        const int pos = tile * 32 + li;
        const bool valid = pos < a.B;
        const int64_t* const idx = a.indices + (a.index_row ? (size_t)a.index_row[0] * a.B : 0);
#ifdef PIME_PPO_ABLATE_GATHER   // timing ablation only: the minibatch is rows 0 .. B-1 (no index load in front of the row loads; wrong results)
        const long long row = valid ? pos : a.B - 1;
#else
        const long long row = idx[valid ? pos : a.B - 1];
#endif
        const float* xrow = a.state + (size_t)row * a.D;
#ifdef PIME_PPO_ABLATE_STASH   // timing ablation only: every workgroup stashes into the first group's tiles (L2-resident, wrong results)
        float* st = a.stash + (size_t)wave * T * 1024;
        const float* st0 = a.stash;
#else
        float* st = a.stash + (size_t)tile * T * 1024;
        const float* st0 = a.stash + (size_t)group * kFusedWaves * T * 1024;  // the group's first tile
#endif
        // Stash regions 1 (and 2): the FIRST-layer activations (round 3).  Rounds 1-2 recomputed them from the states wherever the
        // backward needed them -- as B operand of the second layer's weight-gradient rounds and for act'(H1) -- which is ~1 000
        // vector instructions per wave for the critic and ~2 300 (640 of them transcendental pairs) for the modular actor; with
        // f32 MFMAs every one of those is time on the SIMD, while 64 fire-and-forget stores and prefetched loads are not.
        const size_t region = (size_t)ngroups * kFusedWaves * T * 1024;
        float* st1 = st + region;
        const float* st1_0 = st0 + region;
        [[maybe_unused]] float* st2 = st + 2 * region;
        [[maybe_unused]] const float* st2_0 = st0 + 2 * region;
        const float in_rsum = CRITIC ? a.r_sum[row] : 0.f;
        const float in_action = CRITIC ? 0.f : a.action[row];
        const float in_logprob = CRITIC ? 0.f : a.logprob[row];
        const float in_adv = CRITIC ? 0.f : a.adv[row];
        float x0 = 0.f, x1 = 0.f, x2 = 0.f, x3 = 0.f;   // the first state columns ride along (all of them for the pH / tank envs)
        if (a.D > 0) x0 = xrow[0];
        if (a.D > 1) x1 = xrow[1];
        if (a.D > 2) x2 = xrow[2];
        if (a.D > 3) x3 = xrow[3];
        __syncthreads();  // the previous group's backward is done with W / X
        // the forward images, in the order the layers read them; each layer waits for its own (see the forward below)
        constexpr int NI_TT = T * T * 1024 / (kFusedThreads * 4), NI_TH = T * H * 1024 / (kFusedThreads * 4);   // 8 KB pieces
        constexpr int kStashOps = T * 4;   // vector-memory instructions of one stash_put (vmcnt is an in-order 6-bit counter: the
        //                                   counted waits below name how many YOUNGER operations may stay outstanding)
        static_assert(T * H * 1024 % (kFusedThreads * 4) == 0, "image = whole 8 KB pieces");
        if constexpr (MODULAR) {
            dma_image<NI_TH>(X, a.img_fwd + L.off[1], tid);
            dma_image<NI_TH>(X + T * H * 1024, a.img_fwd + L.off[4], tid);
            dma_image<NI_TT>(wbuf, a.img_fwd + L.off[6], tid);
        } else {
            dma_image<NI_TT>(wbuf, a.img_fwd + L.off[1], tid);
            dma_image<NI_TT>(X, a.img_fwd + L.off[3], tid);
        }
        if (h == 0) {
            float* xw = xs + (wave * 32 + li) * a.D;
            if (a.D > 0) xw[0] = x0;
            if (a.D > 1) xw[1] = x1;
            if (a.D > 2) xw[2] = x2;
            if (a.D > 3) xw[3] = x3;
            for (int c = 4; c < a.D; ++c) xw[c] = xrow[c];
            if (const int P = first_valu_pad(a.D)) {   // padded copy: columns [0, D) the state, column D the constant 1
                auto col = [&](int c, float v) { return c < a.D ? v : (c == a.D ? 1.f : 0.f); };
                float4* xp = reinterpret_cast<float4*>(xsp + (wave * 32 + li) * P);
                xp[0] = make_float4(col(0, x0), col(1, x1), col(2, x2), col(3, x3));
                if (P == 8)
                    xp[1] = make_float4(col(4, a.D > 4 ? xrow[4] : 0.f), col(5, a.D > 5 ? xrow[5] : 0.f),
                                        col(6, a.D > 6 ? xrow[6] : 0.f), col(7, 0.f));
            }
        }
        PIME_LDS_BARRIER();   // the states are in LDS (the images are still in flight)
        PIME_MARK(1);
        const float* xl = xs + (wave * 32 + li) * a.D;   // this lane's state row

        // ---------------------------------------------------------------------------------- forward + loss gradient
        float y;
        f32x16 hl[T];  // last hidden activation; afterwards dZ of the last hidden layer
        // Critic: the first backward step (dH2 = W4^T dZ3) runs BEFORE the weight-gradient rounds of net.4, with H2 kept in
        // registers: net.4's transposed image lands in wbuf behind the last forward layer's MFMAs (wbuf's forward image is dead by
        // then), the stash is not read back for act'(H2), and the stash stores have that whole dX step to drain before the rounds
        // fetch them -- no drain barrier (3.3 us) in front of the rounds: 144 -> 141 us.  The actor kernels keep the old order: with
        // dZ3 and dZ2 both live across the rounds hipcc spills 58 - 63 registers there (+5 us; profiles/r02_q_kernel_variant_ab.txt).
        constexpr bool DX_FIRST = CRITIC;
        f32x16 hk[T];  // DX_FIRST: H2, kept for act'(H2)
        if constexpr (MODULAR) {
            f32x16 cat[T];
            {
                f32x16 a0[T];
                layer_first<T, 2>(lds + F.first0, xl, Do, h, a0);   // activations are applied by the consuming layer
                wait_dma_then_barrier<NI_TH + NI_TT>();               // other_net.2's image has landed
                layer_mfma_in<T, H, 2, 1>(X, lds + F.bias[0], lane, a0, *reinterpret_cast<f32x16(*)[H]>(&cat[0]));
                stash_put<T>(st1, lane, a0);                         // h_o1 (activated in place by the layer above)
            }
            // (counted waits: the stash stores issued since are YOUNGER than the image DMAs and may stay in flight)
            {
                f32x16 a0[T];
                PIME_NO_HOIST();
                layer_first<T, 2>(lds + F.first1, xl + Do, a.Di, h, a0);
                wait_dma_then_barrier<NI_TT + kStashOps>();   // integrator_net.2's: net.0's pieces and other_net's stash stores are younger
                layer_mfma_in<T, H, 2, 1>(X + T * H * 1024, lds + F.bias[1], lane, a0, *reinterpret_cast<f32x16(*)[H]>(&cat[H]));
                stash_put<T>(st2, lane, a0);                         // h_i1
            }
            wait_dma_then_barrier<2 * kStashOps>();   // net.0's (older than both towers' stash stores)
            layer_mfma_in<T, T, 1, 1>(wbuf, lds + F.bias[2], lane, cat, hl);   // cat: tanh applied in place
            stash_put<T>(st, lane, cat);
            PIME_NO_HOIST();
            y = layer_head<T>(lds + F.headw, lds[F.headb], lane, hl);
        } else {
            f32x16(&a1)[T] = hk;
            {
                f32x16 a0[T];
                layer_first<T, 2>(lds + F.first0, xl, a.D, h, a0);   // activations are applied by the consuming layer
                wait_dma_then_barrier<NI_TT>();                       // net.2's image has landed
                layer_mfma_in<T, T, 2, ACT>(wbuf, lds + F.bias[0], lane, a0, a1);
                stash_put<T>(st1, lane, a0);                         // H1 (activated in place by the layer above)
            }
            // net.4's image (older than the stash stores just issued, which may stay in flight); every
            // wave is done with net.2's (wbuf)
            wait_dma_then_barrier<kStashOps>();
            if constexpr (DX_FIRST) dma_image<NI_TT>(wbuf, a.img_bwd + Lb.off[2], tid);
            layer_mfma_in<T, T, ACT, ACT>(X, lds + F.bias[1], lane, a1, hl);     // a1 (H2): activated in place
            stash_put<T>(st, lane, a1);
            PIME_NO_HOIST();
            y = layer_head<T>(lds + F.headw, lds[F.headb], lane, hl);
        }
        PIME_MARK(2);
        float dout = 0.f;
        float s0 = 0.f, s1 = 0.f, gstd = 0.f;
        double m1 = 0.0, m2 = 0.0;
        if (valid) {
            if constexpr (CRITIC) {
                const float d = y - in_rsum, ad = fabsf(d);         // SmoothL1, beta = 1 (agent.py:567,649)
                const float l = ad < 1.f ? 0.5f * d * d : ad - 0.5f;
                const float g = ad < 1.f ? d : (d > 0.f ? 1.f : -1.f);
                dout = g * invB;  // unscaled: critic_scale_kernel applies 1/(std+1e-5) to the finished gradients
                if (h == 0) {
                    s0 = l;
                    m1 = (double)in_rsum;
                    m2 = (double)in_rsum * (double)in_rsum;
                }
            } else {
                const float inv_sigma = __expf(-asl);   // asl = a_std_log, loaded at kernel entry (an L2 round trip here is exposed)
                const float z = (y - in_action) * inv_sigma;
                const float logp = -(asl + kLogSqrt2Pi + 0.5f * z * z);           // compute_logprob
                const float ratio = __expf(logp - in_logprob);
                const float lo = 1.f - a.ratio_clip, hi = 1.f + a.ratio_clip;
                const float clamped = fminf(fmaxf(ratio, lo), hi);
                const float adv = in_adv;
                const float u = adv * ratio, c = adv * clamped;                   // agent.py:639-641
                const float w_u = u < c ? 1.f : (u == c ? 0.5f : 0.f);            // torch.min backward (ties split)
                const float w_c = c < u ? 1.f : (u == c ? 0.5f : 0.f);
                const bool in_range = ratio >= lo && ratio <= hi;
                const float g_sur = w_u * u + (in_range ? w_c * u : 0.f);
                const float p = __expf(logp);
                const float ent = p * logp;                                       // entropy proxy (:643)
                const float g_logp = (-g_sur + a.lambda_entropy * p * (logp + 1.f)) * invB;
                dout = g_logp * (-z * inv_sigma);
                if (h == 0) {
                    gstd = g_logp * (z * z - 1.f);
                    s0 = -fminf(u, c);
                    s1 = ent;
                }
            }
        }
        {
            float ghb = h == 0 ? dout : 0.f;   // head bias gradient
            s0 = wave_total_dpp(s0); ghb = wave_total_dpp(ghb);   // totals valid in lane 63
            if constexpr (CRITIC) {
                for (int o = 32; o > 0; o >>= 1) { m1 += __shfl_xor(m1, o); m2 += __shfl_xor(m2, o); }
            } else {
                s1 = wave_total_dpp(s1); gstd = wave_total_dpp(gstd);
            }
            if (lane == 63) {
                double* w = wsum + wave * 6;
                w[0] += s0; w[1] += s1; w[2] += gstd; w[3] += ghb; w[4] += m1; w[5] += m2;
            }
        }
        // Head: dZ of the last hidden layer, and the head weight gradient gW[f] += sum_s dOut[s] * H_last[s][f].  H_last
        // is still in registers; the sum over the tile's samples is a DPP reduction over the lanes.
        {   // one 32-feature tile at a time: 16 live temporaries instead of a whole T-tile dl (the actor kernel sits at the
            // 256-VGPR cap; spill reloads here queue behind the 64 stash stores just issued on the same vmcnt counter)
            const float* hw = lds + F.headw;
#pragma unroll
            for (int t = 0; t < T; ++t) {
                PIME_NO_HOIST();
                f32x16 dl;
                float hwv[16], prod[16], hs[2];
                load16(vec_at<T>(hw, t, h), hwv);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float hv = hl[t][r];
                    dl[r] = hwv[r] * dout * act_grad_from_output<ACT>(hv);   // head_backward . act'
                    prod[r] = dout * hv;
                }
                half_sums16(prod, hs);   // hs[k]: register r = 8 k + 4 bit2(li) + 2 bit3(li) + bit4(li), feature (r&3) + 8 (r>>2) + 4h
                if ((li & 3) == 0) {     // own slots: reproducible
                    float* q = hacc + wave * md + t * 32 + 4 * h + ((li >> 3) & 1) * 2 + ((li >> 4) & 1) + 8 * ((li >> 2) & 1);
                    q[0] += hs[0];
                    q[16] += hs[1];
                }
                hl[t] = dl;
            }
        }
        // ---------------------------------------------------------------------------------- backward + weight gradients
        // forward images dead; net.4's transposed image is in: the only vector-memory operations issued after its DMA are the
        // stash stores of H2, which may stay outstanding
        if constexpr (DX_FIRST) wait_dma_then_barrier<kStashOps>();
        else __syncthreads();                                  // forward images dead, stash visible to the whole workgroup
        PIME_MARK(3);
        float* const sl = a.slab + (size_t)bid * a.slab_stride;   // this workgroup's partial gradients
        const bool accum = group != bid;                      // a later sample group of the same workgroup
        if constexpr (MODULAR) {
            f32x16(&dn0)[T] = hl;                                                                   // dZn0
            f32x16 dcat[T];
            PIME_MARK(4);
            {
                f32x16 acc[DwPlan<T, T>::PER];
                float bsum;
                PIME_NO_HOIST();
                dw_rounds<T, T>(X, lane, wave, dn0, StashB{st0, T * 1024}, acc, bsum, wbuf, a.img_bwd + Lb.off[3],
                                T * T * 256);
                dw_store_full<T, T>(lane, wave, acc, bsum, sl + a.poff[8], sl + a.poff[9], accum);   // net.0
            }
            PIME_LDS_BARRIER();
            PIME_MARK(5);
            PIME_NO_HOIST();
            layer_mfma<T, T, 2, false>(wbuf, nullptr, lane, dn0, dcat);
            {
                f32x16 cat[T];
                stash_get<T>(st, lane, cat);
                times_act_grad<T, 1>(dcat, cat);                                                    // [dZo2 | dZi2]
            }
            PIME_MARK(6);
            {   // other_net.2 and integrator_net.2: one set of rounds over the W+X region (W is dead until the next dX)
                f32x16 acc[CatPlan<T>::PER];
                float bsum;
                PIME_NO_HOIST();
                dw_rounds<T, 2 * T, CatStashB<T>, CatPlan<T>>(
                    wbuf, lane, wave, dcat, CatStashB<T>{st1_0, st2_0, T * 1024},
                    acc, bsum, nullptr, nullptr, 0, (TRACE && a.trace && bid == a.trace_wg) ? a.trace + 16 : nullptr);
                const CatPlan<T> pl(wave);
                if (pl.active) {
                    const int br = pl.ao >= H ? 1 : 0;
                    float* gW = sl + a.poff[br ? 6 : 2];
#pragma unroll
                    for (int n = 0; n < CatPlan<T>::PER; ++n)
                        block_store(acc[n], gW, md, (pl.ao - br * H) * 32, (pl.bi0 + n - br * T) * 32 + li, true, lane, accum);
                    if (pl.bias) bias_store(bsum, sl + a.poff[br ? 7 : 3] + (pl.ao - br * H) * 32, lane, accum);
                }
            }
            PIME_LDS_BARRIER();   // the rounds are done with the region
            PIME_MARK(7);
            stage_image(wbuf, a.img_bwd + Lb.off[4], 2 * H * T * 256);   // other_net.2^T | integrator_net.2^T
            PIME_LDS_BARRIER();
            PIME_MARK(8);
            if constexpr (T == 4) {
                // One branch at a time (64 instead of 128 live dZ registers): dX with act'(h1) formed behind its MFMAs,
                // then that branch's first-layer gradient.  The branch's transposed image (its half of W) is dead by
                // then and holds the partial sums of the vector form.
#pragma unroll
                for (int br = 0; br < 2; ++br) {
                    const int Din = br ? a.Di : Do, col0 = br ? Do : 0;
                    float* const img = wbuf + br * H * T * 1024;
                    f32x16 d1[T];
                    {
                        f32x16 v[T];   // h1 (activated) from the stash; the layer turns it into act'(h1) behind its MFMAs
                        stash_get<T>(br ? st2 : st1, lane, v);
                        PIME_NO_HOIST();
                        layer_mfma_gate<H, T, 1, true>(img, lane, *reinterpret_cast<f32x16(*)[H]>(&dcat[br * H]), d1, v);   // dZ1
                    }
                    PIME_MARK(9 + br);
                    float* gW = sl + a.poff[br ? 4 : 0], *gb = sl + a.poff[br ? 5 : 1];
                    if (a.D <= kFirstValuMaxD) {
                        first_grad_valu<T>(X, lane, wave, d1, xsp, a.D, col0, Din, gW, gb, accum);
                    } else {
                        f32x16 acc[DwPlan<T, 1>::PER];
                        float bsum;
                        dw_rounds<T, 1>(X, lane, wave, d1, StateB{xs, Din, a.D, col0}, acc, bsum, nullptr, nullptr, 0);
                        dw_store<T, 1>(lane, wave, acc, bsum, gW, Din, md, Din, gb, accum);
                    }
                }
                PIME_MARK(11);
            } else
            {
                f32x16 d1[2 * T];   // [dZo1 | dZi1]
                PIME_NO_HOIST();
                layer_mfma<H, T, 2, false>(wbuf, nullptr, lane, *reinterpret_cast<f32x16(*)[H]>(&dcat[0]),
                                           *reinterpret_cast<f32x16(*)[T]>(&d1[0]));
                {
                    f32x16 hh[T];
                    stash_get<T>(st1, lane, hh);
                    times_act_grad<T, 1>(*reinterpret_cast<f32x16(*)[T]>(&d1[0]), hh);                  // dZo1
                }
                PIME_MARK(9);
                PIME_NO_HOIST();
                layer_mfma<H, T, 2, false>(wbuf + H * T * 1024, nullptr, lane, *reinterpret_cast<f32x16(*)[H]>(&dcat[H]),
                                           *reinterpret_cast<f32x16(*)[T]>(&d1[T]));
                {
                    f32x16 hh[T];
                    stash_get<T>(st2, lane, hh);
                    times_act_grad<T, 1>(*reinterpret_cast<f32x16(*)[T]>(&d1[T]), hh);                  // dZi1
                }
                PIME_MARK(10);
                if (a.D <= kFirstValuMaxD) {   // other_net.0, integrator_net.0 on the vector ALUs
                    first_grad_valu<T>(X, lane, wave, *reinterpret_cast<f32x16(*)[T]>(&d1[0]), xsp, a.D, 0, Do,
                                       sl + a.poff[0], sl + a.poff[1], accum);
                    first_grad_valu<T>(X, lane, wave, *reinterpret_cast<f32x16(*)[T]>(&d1[T]), xsp, a.D, Do, a.Di,
                                       sl + a.poff[4], sl + a.poff[5], accum);
                } else {
                    // matrix form: A = [dZo1 | dZi1], B = the state columns (one tile); wave w owns A tile w; its
                    // product's columns [0,Do) or [Do,D) are the wanted gradient
                    f32x16 acc[DwPlan<2 * T, 1>::PER];
                    float bsum;
                    dw_rounds<2 * T, 1>(wbuf, lane, wave, d1, StateB{xs, a.D, a.D, 0}, acc, bsum, nullptr, nullptr, 0);
                    const DwPlan<2 * T, 1> pl(wave);
                    if (pl.active) {
                        const int br = pl.ao >= T ? 1 : 0;
                        const int ldw = br ? a.Di : Do, col = li - (br ? Do : 0);
                        block_store(acc[0], sl + a.poff[br ? 4 : 0], ldw, (pl.ao - br * T) * 32, col, col >= 0 && col < ldw, lane, accum);
                        bias_store(bsum, sl + a.poff[br ? 5 : 1] + (pl.ao - br * T) * 32, lane, accum);
                    }
                }
                PIME_MARK(11);
            }
        } else {
            f32x16(&d)[T] = hl;                                                                     // dZ3
            f32x16 d2[T];
            PIME_MARK(4);
            if constexpr (DX_FIRST) {
                PIME_NO_HOIST();
                layer_mfma<T, T, 2, false>(wbuf, nullptr, lane, d, d2);                              // dH2
                times_act_grad<T, ACT>(d2, hk);                                                     // dZ2 (H2 from registers)
                __syncthreads();   // wbuf read; the stash stores (issued a whole dX step ago) are visible to the workgroup
                PIME_MARK(5);
            }
            {
                f32x16 acc[DwPlan<T, T>::PER];
                float bsum;
                PIME_NO_HOIST();
                // behind the rounds: the transposed image of the NEXT dX step (DX_FIRST: net.2's, else net.4's)
                dw_rounds<T, T>(X, lane, wave, d, StashB{st0, T * 1024}, acc, bsum, wbuf, a.img_bwd + Lb.off[DX_FIRST ? 3 : 2],
                                T * T * 256);
                dw_store_full<T, T>(lane, wave, acc, bsum, sl + a.poff[4], sl + a.poff[5], accum);      // net.4
            }
            if constexpr (!DX_FIRST) {
                PIME_LDS_BARRIER();
                PIME_MARK(5);
                PIME_NO_HOIST();
                layer_mfma<T, T, 2, false>(wbuf, nullptr, lane, d, d2);
                {
                    f32x16 hh[T];
                    stash_get<T>(st, lane, hh);                                                    // H2
                    times_act_grad<T, ACT>(d2, hh);                                                 // dZ2
                }
            }
            PIME_MARK(6);
            {
                f32x16 acc[DwPlan<T, T>::PER];
                float bsum;
                PIME_NO_HOIST();
                dw_rounds<T, T>(X, lane, wave, d2, StashB{st1_0, T * 1024}, acc, bsum,
                                DX_FIRST ? nullptr : wbuf, DX_FIRST ? nullptr : a.img_bwd + Lb.off[3], DX_FIRST ? 0 : T * T * 256,
                                (TRACE && a.trace && bid == a.trace_wg) ? a.trace + 16 : nullptr);
                dw_store_full<T, T>(lane, wave, acc, bsum, sl + a.poff[2], sl + a.poff[3], accum);      // net.2
            }
            PIME_LDS_BARRIER();
            PIME_MARK(7);
            PIME_NO_HOIST();
            layer_mfma<T, T, 2, false>(wbuf, nullptr, lane, d2, d);
            {
                f32x16 hh[T];
                stash_get<T>(st1, lane, hh);                                                        // H1
                times_act_grad<T, ACT>(d, hh);                                                      // dZ1
            }
            PIME_MARK(8);
            if (a.D <= kFirstValuMaxD) {   // net.0 on the vector ALUs (W is dead: its LDS holds the partial sums)
                first_grad_valu<T>(X, lane, wave, d, xsp, a.D, 0, a.D, sl + a.poff[0], sl + a.poff[1], accum);
            } else {
                f32x16 acc[DwPlan<T, 1>::PER];
                float bsum;
                dw_rounds<T, 1>(X, lane, wave, d, StateB{xs, a.D, a.D, 0}, acc, bsum, nullptr, nullptr, 0);
                dw_store<T, 1>(lane, wave, acc, bsum, sl + a.poff[0], a.D, md, a.D, sl + a.poff[1], accum);    // net.0
            }
        }
    }

    PIME_MARK(12);
    if constexpr (TRACE)
        if (a.trace_span && threadIdx.x == 0 && bid < 512) a.trace_span[2 * bid + 1] = wall_clock64();
    // ---- workgroup totals of the scalar sums, combined in a fixed order (the slabs make the gradients reproducible
    // bit for bit; only the loss sums, which are for logging, use atomics)
    __syncthreads();
    float* const sl = a.slab + (size_t)bid * a.slab_stride;
    constexpr int NP = MODULAR ? 12 : 8;
    if (tid < md) {   // head weight
        float t = hacc[tid];
        for (int w = 1; w < kFusedWaves; ++w) t += hacc[w * md + tid];
        sl[a.poff[NP - 2] + tid] = t;
    }
    if (tid == 0) {
        double t[6] = {0, 0, 0, 0, 0, 0};
        for (int w = 0; w < kFusedWaves; ++w)
            for (int k = 0; k < 6; ++k) t[k] += wsum[w * 6 + k];
        sl[a.poff[NP - 1]] = (float)t[3];                 // head bias
        if constexpr (CRITIC) {
            atomicAdd(&a.loss_sums[2], (float)t[0]);
            double* mo = reinterpret_cast<double*>(sl + a.poff[NP]);
            mo[0] = t[4]; mo[1] = t[5];
        } else {
            atomicAdd(&a.loss_sums[0], (float)t[0]);
            atomicAdd(&a.loss_sums[1], (float)t[1]);
            sl[a.poff[NP]] = (float)t[2];                 // d loss / d a_std_log
        }
    }
}

template <int T, int KIND>
__global__ __launch_bounds__(kFusedThreads) void ppo_fused_kernel(PpoArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // tuning aid: shader-clock ticks (s_memtime) and 100 MHz ticks (s_memrealtime) of the traced workgroup -> the clock the SIMDs ran at
    const bool timing = a.trace && (int)blockIdx.x == a.trace_wg && threadIdx.x == 0;
    long long c0 = 0, w0 = 0;
    if (timing) { c0 = (long long)__builtin_readcyclecounter(); w0 = wall_clock64(); }
    ppo_fused_body<T, KIND, true>(a, lds, (int)blockIdx.x, (int)gridDim.x);
    if (timing) { a.trace[38] = wall_clock64() - w0; a.trace[39] = (long long)__builtin_readcyclecounter() - c0; }
}

// Both nets of an optimizer step in ONE launch: workgroups [0, na) run the actor's body, [na, na + nc) the critic's.  The two
// gradients are independent, so nothing orders them; as separate launches the second could not start before the slowest
// workgroup of the first had finished (launch ramp + tail + launch boundary, ~9 us of a 320 us step), here a compute unit that
// finishes an actor workgroup picks up a critic one at once.  The longer body (the actor's) is dispatched first, so the tail is
// made of the shorter ones.  (Round 1 tried a dual launch on the first version of these kernels and lost 1.5 %; the bodies have
// since shed their spill scratch and ~30 us each.)
template <int T, int AKIND>
__global__ __launch_bounds__(kFusedThreads) void ppo_fused_dual_kernel(PpoArgs actor, PpoArgs critic, int na) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int b = (int)blockIdx.x;
    if (b < na) ppo_fused_body<T, AKIND, false>(actor, lds, b, na);
    else ppo_fused_body<T, MLP_CRITIC, false>(critic, lds, b - na, (int)gridDim.x - na);
}

// ==================================================================================================== slab reduction
// grad[e] += scale * sum over the workgroups' slabs, in slab order (reproducible), for both nets in one launch.  The
// critic's scale 1/(std(targets)+1e-5) (agent.py:652) comes from the float64 target moments the critic kernel left in
// its slabs.  Workgroup = 64 consecutive float4 of one parameter; wave w sums slabs w, w+8, ...; LDS combines the waves.
struct ReduceSeg {
    float* dst;
    int off, n, net;   // slab offset (floats), element count, 0 = critic / 1 = actor
    int perm_tb;       // 0: slab in tensor order; TB > 0: block-major accumulator order with TB column tiles (slab_layout16)
    int ldw, ncols;    // perm_tb > 0: row length and valid columns of the destination tensor
};
// Optional optimizer step fused into the reduction (pime_ppo_minibatch_step): the gradient tensors are views into ONE flat
// buffer and the parameters views into another at the same offsets, so the element a thread has just reduced is also the
// element it updates (torch.optim.Adam semantics, as adam_kernel).  A whole launch and its boundary less per optimizer step.
struct ReduceArgs {
    ReduceAdam adam;    // flat_grad == nullptr: no optimizer step
    ReduceSeg seg[24];
    int nseg, nslabs[2], B, moments_off, overwrite;   // nslabs per net (the two nets may come from different kernel families)
    int has_critic;     // 0: the critic went through the split pipeline (critic_scale_kernel finishes it): no scale, no cursor
    float* scale_sum;   // loss_sums + 3: [0] += scale, [1] += critic loss sum of this call * scale, [2] = critic sum so far
    int64_t* index_row; // NULL, or the index table's row cursor: advanced once per call, after both nets read it
    const float* slab[2];
    int stride[2];
    float* scale_out;
    double* moments_out;
    float* dp_moments;   // not NULL (data parallel): the critic's gradient stays unscaled; [0..2] = sum r, sum r^2, B of this rank's minibatch
};

#ifndef PIME_REDUCE_BATCH
#define PIME_REDUCE_BATCH 4
#endif
constexpr int kReduceBatch = PIME_REDUCE_BATCH;
__global__ __launch_bounds__(512) void ppo_grad_reduce_kernel(ReduceArgs a) {
    __shared__ float4 part[8][64];
    __shared__ float scale_sh;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // which segment / chunk
    int c = blockIdx.x, si = 0;
    for (; si < a.nseg; ++si) {
        const int chunks = ((a.seg[si].n + 3) / 4 + 63) / 64;
        if (c < chunks) break;
        c -= chunks;
    }
    if (si >= a.nseg) return;
    const ReduceSeg sg = a.seg[si];
    // Adam's bias corrections (two float64 pow, a dependent load of the step count in front of them) are worked out by wave 1 and
    // handed over through LDS: wave 0, which finishes the elements, is the workgroup's critical path.  Every workgroup reads the
    // OLD step count; the last one to finish stores the new one.
    __shared__ float adam_sh[3];   // t_new, step_size, bc2_sqrt
    if (a.adam.flat_grad && wave == 1) {
        const float tn = a.adam.step[0] + 1.0f;
        const double t = (double)tn;
        const float ss = a.adam.lr / (float)(1.0 - pow((double)a.adam.b1, t));
        const float bs = (float)sqrt(1.0 - pow((double)a.adam.b2, t));
        if (lane == 0) { adam_sh[0] = tn; adam_sh[1] = ss; adam_sh[2] = bs; }
    }
    if (wave == 0 && (!a.has_critic || (sg.net != 0 && blockIdx.x != 0))) {
        if (lane == 0) scale_sh = 1.0f;   // an actor segment needs no scale (workgroup 0 publishes it: always computed there)
    } else if (wave == 0) {  // critic scale from the moments (every workgroup that needs it: same slabs, same order, same value)
        double m1 = 0.0, m2 = 0.0;
        // wave 0 is the workgroup's critical path (it also sums its share of the slabs): the moment loads of up to 512 slabs are
        // issued together -- one memory round trip, not one per 64 slabs -- and summed in the same order as before
        constexpr int kMaxIt = 8;
        double2 mv[kMaxIt];
#pragma unroll
        for (int k = 0; k < kMaxIt; ++k) {
            const int s = lane + 64 * k;
            mv[k] = make_double2(0.0, 0.0);
            if (s < a.nslabs[0])
                mv[k] = *reinterpret_cast<const double2*>(a.slab[0] + (size_t)s * a.stride[0] + a.moments_off);
        }
#pragma unroll
        for (int k = 0; k < kMaxIt; ++k) { m1 += mv[k].x; m2 += mv[k].y; }
        for (int s = lane + 64 * kMaxIt; s < a.nslabs[0]; s += 64) {   // (grids beyond 512 slabs: not used today)
            const double* mo = reinterpret_cast<const double*>(a.slab[0] + (size_t)s * a.stride[0] + a.moments_off);
            m1 += mo[0]; m2 += mo[1];
        }
        for (int o = 32; o > 0; o >>= 1) { m1 += __shfl_xor(m1, o); m2 += __shfl_xor(m2, o); }
        const double B = (double)a.B;
        const double var = a.B > 1 ? fmax((m2 - m1 * m1 / B) / (B - 1.0), 0.0) : 0.0;
        const float scale = (float)(1.0 / ((double)(float)sqrt(var) + 1e-5));
        if (lane == 0) {
            scale_sh = a.dp_moments ? 1.0f : scale;   // data parallel: the scale of the UNION minibatch is applied behind the all-reduce
            if (blockIdx.x == 0) {
                if (a.dp_moments) { a.dp_moments[0] = (float)m1; a.dp_moments[1] = (float)m2; a.dp_moments[2] = (float)a.B; a.dp_moments[3] = 0.f; }
                a.scale_out[0] = scale; a.moments_out[0] = m1; a.moments_out[1] = m2; a.scale_sum[0] += scale;
                // loss_sums[4] += (this call's SmoothL1 sum) * scale: the logged united loss is the mean of the per-step
                // values actor + critic * scale (agent.py:652), not mean(critic) * mean(scale).  scale_sum = loss_sums + 3.
                const float csum = a.scale_sum[-1];
                a.scale_sum[1] += (csum - a.scale_sum[2]) * scale;
                a.scale_sum[2] = csum;
                if (a.index_row) a.index_row[0] += 1;
            }
        }
    }
    const int unit = c * 64 + lane, n4 = (sg.n + 3) / 4;
    // Where this lane's four sums go (wave 0 finishes them), and -- wave 0 is the workgroup's critical path -- the optimizer state
    // of those four parameters, requested NOW so that it arrives behind the slab loads instead of in a chain of its own at the end.
    float* q4[4];
    bool ok4[4], adam4[4];
    float pm[4], pv[4], pp[4];
    int2 pmap[4];
    {
        const int blk = unit >> 6, ln = unit & 63;   // perm_tb > 0: one lane's four accumulator registers of block (blk / TB, blk % TB)
        const int tb = sg.perm_tb > 0 ? sg.perm_tb : 1;
        const int row = (blk / tb) * 16 + 4 * (ln >> 4), col = (blk % tb) * 16 + (ln & 15);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (sg.perm_tb > 0) {
                q4[k] = &sg.dst[(row + k) * sg.ldw + col];
                ok4[k] = unit < n4 && col < sg.ncols;
            } else {
                q4[k] = &sg.dst[unit * 4 + k];
                ok4[k] = unit * 4 + k < sg.n;
            }
            adam4[k] = false;
            pm[k] = pv[k] = pp[k] = 0.f;
            pmap[k] = make_int2(-1, -1);
            if (wave == 0 && a.adam.flat_grad && ok4[k]) {
                const long long off = q4[k] - a.adam.flat_grad;
                if (off >= 0 && off < a.adam.n) {   // (gradients of frozen parameters land in a dump buffer outside the flat one)
                    adam4[k] = true;
                    pm[k] = a.adam.exp_avg[off]; pv[k] = a.adam.exp_avg_sq[off]; pp[k] = a.adam.flat_param[off];
                    if (a.adam.image_map) pmap[k] = reinterpret_cast<const int2*>(a.adam.image_map)[off];
                }
            }
        }
    }
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (unit < n4) {
        const float* base = a.slab[sg.net] + sg.off + unit * 4;
        const size_t stride = (size_t)a.stride[sg.net];
        const int nslabs = a.nslabs[sg.net];
        int s = wave;
        // kReduceBatch loads in flight per thread (summed in slab order whatever the batch: the result does not depend on it)
        for (; s + 8 * (kReduceBatch - 1) < nslabs; s += 8 * kReduceBatch) {
            float4 v[kReduceBatch];
#pragma unroll
            for (int k = 0; k < kReduceBatch; ++k) v[k] = *reinterpret_cast<const float4*>(base + (size_t)(s + 8 * k) * stride);
#pragma unroll
            for (int k = 0; k < kReduceBatch; ++k) { acc.x += v[k].x; acc.y += v[k].y; acc.z += v[k].z; acc.w += v[k].w; }
        }
        for (; s < nslabs; s += 8) {
            const float4 v = *reinterpret_cast<const float4*>(base + (size_t)s * stride);
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
    }
    part[wave][lane] = acc;
    __syncthreads();
    const float t_new = a.adam.flat_grad ? adam_sh[0] : 0.f, step_size = a.adam.flat_grad ? adam_sh[1] : 0.f,
                bc2_sqrt = a.adam.flat_grad ? adam_sh[2] : 1.f;
    if (wave == 0 && unit < n4) {
        float4 t = part[0][lane];
        for (int w = 1; w < 8; ++w) { const float4 v = part[w][lane]; t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w; }
        const float sc = sg.net == 0 ? scale_sh : 1.0f;
        const float o[4] = {t.x * sc, t.y * sc, t.z * sc, t.w * sc};
        // store the finished gradient elements and, if asked, apply Adam to their parameters (torch.optim.Adam, as adam_kernel)
        float gv[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            gv[k] = o[k];
            if (ok4[k] && !a.overwrite) gv[k] += *q4[k];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (!ok4[k]) continue;
            *q4[k] = gv[k];
            if (adam4[k]) {
                const long long off = q4[k] - a.adam.flat_grad;
                const float mi = pm[k] + (gv[k] - pm[k]) * (1.0f - a.adam.b1);
                const float vi = pv[k] * a.adam.b2 + gv[k] * gv[k] * (1.0f - a.adam.b2);
                a.adam.exp_avg[off] = mi;
                a.adam.exp_avg_sq[off] = vi;
                const float pn = pp[k] - step_size * (mi / (sqrtf(vi) / bc2_sqrt + a.adam.eps));
                a.adam.flat_param[off] = pn;
                // the packed images are permutations of the parameters: keep them current here (bits 28..29 of a map entry: the net,
                // for pime_adam_step_images)
                if (pmap[k].x >= 0) a.adam.img[sg.net][0][pmap[k].x & 0x0fffffff] = pn;
                if (pmap[k].y >= 0) a.adam.img[sg.net][1][pmap[k].y & 0x0fffffff] = pn;
            }
        }
    }
    if (a.adam.flat_grad) {
        __syncthreads();
        if (tid == 0) {
            unsigned int* arrivals = reinterpret_cast<unsigned int*>(a.adam.step + 1);
            if (atomicAdd(arrivals, 1u) == gridDim.x - 1) {
                *arrivals = 0;
                a.adam.step[0] = t_new;
            }
        }
    }
}

// ==================================================================================================== re-pack
// After an optimizer step both nets' forward and transposed images are stale: one launch re-lays all four
// (blockIdx.y: critic fwd, critic bwd, actor fwd, actor bwd) instead of four ~5 us launches per step.
struct RepackArgs {
    PackArgs net[2];
    float* img[4];
};

__global__ void ppo_repack_kernel(RepackArgs a) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x, nthr = gridDim.x * blockDim.x;
    const PackArgs& n = a.net[blockIdx.y >> 1];
    if (blockIdx.y & 1) pack_backward_image(n, a.img[blockIdx.y], tid, nthr);
    else pack_forward_image(n, a.img[blockIdx.y], tid, nthr);
}

int launch_repack(const PackArgs& critic, const PackArgs& actor, float* c_fwd, float* c_bwd, float* a_fwd, float* a_bwd,
                  hipStream_t s) {
    RepackArgs r{};
    r.net[0] = critic; r.net[1] = actor;
    r.img[0] = c_fwd; r.img[1] = c_bwd; r.img[2] = a_fwd; r.img[3] = a_bwd;
    hipLaunchKernelGGL(ppo_repack_kernel, dim3(32, 4), dim3(256), 0, s, r);
    PIME_HIP_TRY(hipGetLastError());
    return PIME_OK;
}

int fused_grid(int B) {
    // PIME_FUSED_GRID=<n>: tuning aid -- fewer workgroups than 256-sample groups, so that a workgroup runs several groups back to
    // back (the second one from a warm instruction cache); production: one workgroup per group up to 256
    static const int cap = [] {
        const char* e = std::getenv("PIME_FUSED_GRID");
        const int v = e ? std::atoi(e) : 0;
        return v > 0 && v < 256 ? v : 256;
    }();
    const int ntiles = (B + 31) / 32;
    int grid = (ntiles + kFusedWaves - 1) / kFusedWaves;
    return grid > cap ? cap : grid;
}

int launch_grad_reduce(const PpoArgs& critic, const PpoArgs& actor, int kind_c, int md_c, int kind_a, int md_a,
                       bool f16_c, bool f16_a, bool use_c, bool use_a, int nslabs_c, int nslabs_a, float* const* grads_c, float* const* grads_a, float* g_std, float* scale_out,
                       double* moments_out, float* scale_sum, int overwrite, int64_t* index_row, const ReduceAdam* adam,
                       float* dp_moments, hipStream_t s) {
    ReduceArgs r{};
    if (adam) r.adam = *adam;
    int poff[13], psize[12], chunks = 0;
    auto add = [&](float* dst, int off, int n, int net, int perm_tb = 0, int ldw = 0, int ncols = 0) {
        r.seg[r.nseg++] = ReduceSeg{dst, off, n, net, perm_tb, ldw, ncols};
        chunks += ((n + 3) / 4 + 63) / 64;
    };
    auto add_net = [&](const PpoArgs& a, int kind, int md, bool f16, float* const* grads, int net, int* scalar_off) {
        const int np = kind == MLP_MODULAR_ACTOR ? 12 : 8;
        if (f16 && kind == MLP_MODULAR_ACTOR) {   // slab_layout16m: five block-major matrices
            slab_layout16m(md, poff, psize);
            const int T = md / 16, Do = a.D - a.Di;
            for (int i = 0; i < np; ++i) {
                if (i == 0) add(grads[i], poff[i], psize[i], net, 1, Do, Do);
                else if (i == 4) add(grads[i], poff[i], psize[i], net, 1, a.Di, a.Di);
                else if (i == 2 || i == 6 || i == 8) add(grads[i], poff[i], psize[i], net, T, md, md);
                else add(grads[i], poff[i], psize[i], net);
            }
        } else if (f16) {   // block-major weight gradients (slab_layout16)
            slab_layout16(a.D, md, poff, psize);
            const int tb0 = ((a.D + 3) & ~3) <= 16 ? 1 : 2, T = md / 16;
            for (int i = 0; i < np; ++i) {
                if (i == 0) add(grads[i], poff[i], psize[i], net, tb0, a.D, a.D);
                else if (i == 2 || i == 4) add(grads[i], poff[i], psize[i], net, T, md, md);
                else add(grads[i], poff[i], psize[i], net);
            }
        } else {
            slab_layout(kind, a.D, a.Di, md, poff, psize);
            for (int i = 0; i < np; ++i) add(grads[i], poff[i], psize[i], net);
        }
        *scalar_off = poff[np];
    };
    // use_c / use_a: which nets left slabs (a net served by the split pipeline wrote its gradients with atomics already)
    int scalar_c = 0, scalar_a = 0;
    r.has_critic = use_c ? 1 : 0;
    if (use_c) add_net(critic, kind_c, md_c, f16_c, grads_c, 0, &scalar_c);
    r.moments_off = scalar_c;
    if (use_a) {
        add_net(actor, kind_a, md_a, f16_a, grads_a, 1, &scalar_a);
        add(g_std, scalar_a, 1, 1);
    }
    if (chunks == 0) return PIME_OK;
    r.nslabs[0] = nslabs_c; r.nslabs[1] = nslabs_a; r.B = critic.B;
    r.slab[0] = critic.slab; r.slab[1] = actor.slab;
    r.stride[0] = critic.slab_stride; r.stride[1] = actor.slab_stride;
    r.scale_out = scale_out; r.moments_out = moments_out; r.scale_sum = scale_sum; r.overwrite = overwrite; r.index_row = index_row;
    r.dp_moments = dp_moments;
    hipLaunchKernelGGL(ppo_grad_reduce_kernel, dim3(chunks), dim3(512), 0, s, r);
    PIME_HIP_TRY(hipGetLastError());
    return PIME_OK;
}

// Floats of workspace the fused kernel needs: activation stash (one hidden activation per sample) + gradient slabs.
// One stash region holds one width-md activation of every sample: [tile][md / 32][16][64].  Regions: 0 = the hidden activation in
// front of the last hidden layer (critic / plain actor H2, modular actor `cat`), 1 = the first-layer activation H1 (modular actor:
// other_net's), 2 = the modular actor's integrator_net first-layer activation.
int64_t fused_stash_region_floats(int B, int md) {
    const int64_t ngroups = ((B + 31) / 32 + kFusedWaves - 1) / kFusedWaves;
    return ngroups * kFusedWaves * (md / 32) * 1024;
}
int fused_stash_regions(int kind) { return kind == MLP_MODULAR_ACTOR ? 3 : 2; }
int64_t fused_stash_floats(int kind, int B, int md) { return fused_stash_regions(kind) * fused_stash_region_floats(B, md); }
int64_t fused_workspace_floats(int kind, int B, int D, int Di, int md) {
    int poff[13], psize[12];
    return fused_stash_floats(kind, B, md) + (int64_t)fused_grid(B) * slab_layout(kind, D, Di, md, poff, psize);
}

template <int T, int KIND>
static int launch_fused(const PpoArgs& a, hipStream_t s) {
    const FusedLds F = fused_lds(KIND, a.D, a.Di, T);
    const size_t lds_bytes = sizeof(float) * (size_t)F.total;
    PIME_REQUIRE(lds_bytes <= 160 * 1024, "fused PPO kernel needs %zu B of LDS (> 160 KB) for state_dim %d", lds_bytes, a.D);
    static LdsLimit lds_limit;  // per instantiation
    PIME_RAISE_LDS(lds_limit, (ppo_fused_kernel<T, KIND>), 160 * 1024);
    const int grid = fused_grid(a.B);
    hipLaunchKernelGGL((ppo_fused_kernel<T, KIND>), dim3(grid), dim3(kFusedThreads), lds_bytes, s, a);
    PIME_HIP_TRY(hipGetLastError());
    return PIME_OK;
}

template <int T, int AKIND>
static int launch_dual(const PpoArgs& actor, const PpoArgs& critic, hipStream_t s) {
    const size_t la = sizeof(float) * (size_t)fused_lds(AKIND, actor.D, actor.Di, T).total;
    const size_t lc = sizeof(float) * (size_t)fused_lds(MLP_CRITIC, critic.D, critic.Di, T).total;
    const size_t lds_bytes = la > lc ? la : lc;
    PIME_REQUIRE(lds_bytes <= 160 * 1024, "fused PPO kernels need %zu B of LDS (> 160 KB)", lds_bytes);
    static LdsLimit lds_limit;  // per instantiation
    PIME_RAISE_LDS(lds_limit, (ppo_fused_dual_kernel<T, AKIND>), 160 * 1024);
    const int na = fused_grid(actor.B), nc = fused_grid(critic.B);
    hipLaunchKernelGGL((ppo_fused_dual_kernel<T, AKIND>), dim3(na + nc), dim3(kFusedThreads), lds_bytes, s, actor, critic, na);
    PIME_HIP_TRY(hipGetLastError());
    return PIME_OK;
}

// Actor and critic of the same width in one launch (see ppo_fused_dual_kernel); PIME_ERR_ARG if there is no instantiation.
int launch_ppo_fused_dual(int actor_kind, int md, const PpoArgs& actor, const PpoArgs& critic, hipStream_t s) {
    const int T = md / 32;
#define PIME_DUAL(TT, KK) \
    if (T == TT && actor_kind == KK) return launch_dual<TT, KK>(actor, critic, s);
    PIME_DUAL(4, MLP_MODULAR_ACTOR) PIME_DUAL(4, MLP_PLAIN_ACTOR) PIME_DUAL(2, MLP_MODULAR_ACTOR) PIME_DUAL(2, MLP_PLAIN_ACTOR)
#undef PIME_DUAL
    set_error("no dual fused PPO instantiation for actor kind %d width %d", actor_kind, md);
    return PIME_ERR_ARG;
}

// Does the kernel's LDS map fit?  (wide observations, e.g. the stacked water tank, do not: the caller then uses the
// split net + dW pipeline of ppo_train.hip)
bool fused_fits(int kind, int D, int Di, int md) {
    return sizeof(float) * (size_t)fused_lds(kind, D, Di, md / 32).total <= 160 * 1024;
}

int launch_ppo_fused(int kind, int md, const PpoArgs& a, hipStream_t s) {
    const int T = md / 32;
#define PIME_NET(TT, KK) \
    if (T == TT && kind == KK) return launch_fused<TT, KK>(a, s);
    PIME_NET(2, MLP_CRITIC) PIME_NET(4, MLP_CRITIC)
    PIME_NET(2, MLP_PLAIN_ACTOR) PIME_NET(4, MLP_PLAIN_ACTOR)
    PIME_NET(2, MLP_MODULAR_ACTOR) PIME_NET(4, MLP_MODULAR_ACTOR)
#undef PIME_NET
    set_error("no fused PPO instantiation for kind %d width %d", kind, md);
    return PIME_ERR_ARG;
}

}  // namespace pime
