// The 16-sample-tile kernel family: 4-layer MLPs (CriticAdv / ActorPPO / ActorResidualPPO) of width 64, 128 or 256 on
// v_mfma_f32_16x16x4_f32 with the md x md weight images STREAMED through LDS in k-slices instead of resident in it.
//
// replaces (reference, /root/reference): the value pass elegantrl/agent.py:619-620 (net.py:274-277), the policy mean
// net_residual.py:19-22,45-48, and -- per optimizer step of AgentPPO.update_net, agent.py:629-657 -- gather, compute_logprob,
// clipped surrogate + entropy proxy, SmoothL1, the united loss and `obj_united.backward()`, at the width the reference's live
// water-tank script trains (net_dim 256 on the 30-float Stacking10 observation, run_watertank_changing.sh:20-27), which the
// LDS-resident 32x32x2 family (mlp_mfma.hip / ppo_fused.hip) cannot hold.
//
// Why 16-sample tiles: a 16x16 accumulator block is 4 registers, so a whole width-md activation of a tile is md/4 registers
// per lane (32 at md 128, 64 at md 256) instead of md/2.  At md 256 that is what makes "input + output activation of a layer
// in registers" possible at all.  Workgroups are FOUR waves (one per SIMD, 64 samples per pass): at md <= 128 the LDS map is
// under 80 KB, so two independent workgroups share a CU and every SIMD holds two waves whose barrier-separated phases are NOT
// in lock-step with each other (the 32x32x2 gradient kernel runs one 8-wave workgroup per CU and its matrix pipe idles through
// every non-MFMA phase: profiles/r01_l_pmc_pipe.json); at md 256 a wave owns its SIMD's whole 512-register file, which is what
// holds three 64-register activations plus a 64-register gradient patch without a stash.
//
// Layout facts (v_mfma_f32_16x16x4_f32; lane l: i = l & 15, g = l >> 4):
//   A[i][k = g] (weights: i = output feature), B[k = g][j = i] (activations: j = sample), D[4g + r][j] in register r.
//   So the accumulator of a layer has sample l&15 on the lane and output features 16t + 4g + r in register r of tile t, and
//   register r of tile t of every lane IS the B operand of the next layer's k-step (t, r), which covers the input features
//   {16t + 4g' + r : g' = 0..3}: the weight image is packed in that k order ("chain order"), activations never leave
//   registers between layers.  Weight gradients contract over the SAMPLE axis, which sits on the lanes: both operands go
//   through a sample-major LDS image (row = one sample's features, pitch md + 16 floats: conflict-free both ways).
//
// Gradient kernel, per 64-sample group (4 waves x 16 samples), everything on chip, no activation stash:
//   forward  h1 = act(W0 x) [MFMA from the LDS-resident first-layer image], h2, h3 [streamed], y [VALU dot + 2 shuffles]
//   loss gradient, head gradient (DPP row sums), dZ3 in place of h3
//   dW2 = dZ3^T h2 [2 rounds of 32 samples, operands published from registers], dH2 = W2^T dZ3 [streamed], dZ2
//   dW1 = dZ2^T h1 [h1 recomputed], dH1 = W1^T dZ2, dZ1,  dW0 = dZ1^T x
//   partial gradients -> this workgroup's slab (16-byte stores in accumulator order, slab_layout16), summed in slab order and
//   un-permuted by ppo_grad_reduce_kernel: reproducible bit for bit.
#include "env_device.hpp"
#include "ppo_device.hpp"
#include "ppo_train.hpp"
#include "rollout.hpp"

namespace pime {

using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int k16Threads = 256;
constexpr int k16Waves = 4;     // one wave per SIMD; a second workgroup on the CU supplies each SIMD's second wave
constexpr int k16Group = 64;    // samples per workgroup pass: 4 tiles of 16

// ---- images ------------------------------------------------------------------------------------------------------------
// MFMA layer image [k-step][quad q][lane][4]: element e of lane (i, g) of quad q at k-step ks is W[16(4q + e) + i][k(ks, g)],
// k(ks, g) = 16 (ks >> 2) + 4g + (ks & 3) in chain order, 4 ks + g in natural order (first layer: the B operand is x itself).
struct Layout16 {
    int T, Dp, KS0;
    int w0, b0, w1, b1, w2, b2, w3, b3, total;   // forward image (float offsets)
};
struct LayoutB16 {
    int w2t, w1t, total;                         // backward image: W2^T, W1^T as chain-order layers
};

__host__ __device__ inline Layout16 layout16(int D, int md) {
    Layout16 L{};
    L.T = md / 16;
    L.Dp = (D + 3) & ~3;
    L.KS0 = L.Dp / 4;
    int o = 0;
    auto seg = [&](int& f, int n) { f = o; o += (n + 3) & ~3; };
    seg(L.w0, L.KS0 * L.T * 64);
    seg(L.b0, md);
    seg(L.w1, md * md);
    seg(L.b1, md);
    seg(L.w2, md * md);
    seg(L.b2, md);
    seg(L.w3, md);
    seg(L.b3, 4);
    L.total = o;
    return L;
}
__host__ __device__ inline LayoutB16 layoutb16(int md) {
    LayoutB16 L{};
    L.w2t = 0; L.w1t = md * md; L.total = 2 * md * md;
    return L;
}

// W: nn.Linear [O][K] row-major.  transposed: the layer computes V = W^T (outputs = W's columns, k = W's rows).
__device__ inline void pack16_layer(float* __restrict__ dst, const float* __restrict__ W, int O, int K, int n_out, int n_k,
                                    int KS, bool natural, bool transposed, int tid, int nthr) {
    const int Q = n_out / 64 > 0 ? n_out / 64 : 1;   // quads of 4 output tiles (n_out is a multiple of 64)
    const int n = KS * Q * 256;
    for (int idx = tid; idx < n; idx += nthr) {
        const int e = idx & 3, lane = (idx >> 2) & 63, q = (idx >> 8) % Q, ks = idx / (Q * 256);
        const int i = lane & 15, g = lane >> 4;
        const int out = 16 * (4 * q + e) + i;
        const int k = natural ? 4 * ks + g : 16 * (ks >> 2) + 4 * g + (ks & 3);
        float v = 0.f;
        if (out < n_out && k < n_k) v = transposed ? W[(size_t)k * K + out] : W[(size_t)out * K + k];
        dst[idx] = v;
    }
    (void)O;
}

__global__ void pack16_kernel(PackArgs a, float* __restrict__ fwd, float* __restrict__ bwd) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x, nthr = gridDim.x * blockDim.x;
    const int md = a.md;
    if (fwd) {
        const Layout16 L = layout16(a.D, md);
        pack16_layer(fwd + L.w0, a.p[0], md, a.D, md, a.D, L.KS0, true, false, tid, nthr);
        pack16_layer(fwd + L.w1, a.p[2], md, md, md, md, md / 4, false, false, tid, nthr);
        pack16_layer(fwd + L.w2, a.p[4], md, md, md, md, md / 4, false, false, tid, nthr);
        for (int i = tid; i < md; i += nthr) {
            fwd[L.b0 + i] = a.p[1][i];
            fwd[L.b1 + i] = a.p[3][i];
            fwd[L.b2 + i] = a.p[5][i];
            fwd[L.w3 + i] = a.p[6][i];
        }
        if (tid < 4) fwd[L.b3 + tid] = tid == 0 ? a.p[7][0] : 0.f;
    }
    if (bwd) {
        const LayoutB16 L = layoutb16(md);
        pack16_layer(bwd + L.w2t, a.p[4], md, md, md, md, md / 4, false, true, tid, nthr);
        pack16_layer(bwd + L.w1t, a.p[2], md, md, md, md, md / 4, false, true, tid, nthr);
    }
}

#define PIME16_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

template <int ACT>
__device__ __forceinline__ float act16(float v) {
    if constexpr (ACT == 0) return v > 0.f ? v : 0.f;
    else if constexpr (ACT == 1) return fast_tanh(v);
    else return v;
}

// ---- a chain layer with its weight image streamed through three LDS slice buffers by LDS-DMA ---------------------------
// out = act(W in + b); the whole workgroup calls this together (every wave works on its own 16-sample tile, all share the
// slices).  Slice = SKS k-steps of the image (16 KB at width <= 128, 32 KB at 256), moved by global_load_lds_dwordx4 (no
// staging registers, no ds_write; per wave-instruction 1 KB lands at a wave-uniform LDS base + lane * 16, the global address
// is per lane).  Iteration s: the DMA of slice s+2 is issued into buffer (s+2)%3 (last read in iteration s-1), the MFMAs of
// slice s run from buffer s%3 with the weight fragments of the next group of k-steps (16 MFMAs = 512 pipe cycles) in flight,
// then a counted s_waitcnt leaves only the DMA just issued outstanding (slice s+1 has landed) and ONE barrier publishes it.
// hipcc sinks register-staged loads down to their ds_write (the L2 round trip then sits exposed in front of every barrier)
// and lets one ds_read run ahead at most: the DMA has no register to sink, and sched_barrier pins the fragment reads.
// wbuf must be free on entry (callers barrier before); it is free again on return.
// TK input tiles (k = 16 TK features), TO output tiles; the square layers of the plain nets are TK = TO = T, the modular actor's
// tower layers md -> md/2 are (T, T/2) and their transposes (T/2, T).
template <int TK, int TO = TK>
struct Layer16Geom {
    static constexpr int Q = TO / 4;
    static constexpr int KS = TK * 4;                      // k-steps of the layer (input width / 4)
    static constexpr int SKS = TK >= 8 ? 8 : KS;           // k-steps per slice
    static constexpr int NS = KS / SKS;
    static constexpr int SLICE = SKS * Q * 256;            // floats
    static constexpr int NBUFW = NS >= 3 ? 3 : NS;
    static constexpr int PER = SLICE / 4 / k16Threads;     // 1 KB DMA pieces per wave per slice
    static constexpr int KSG = TO >= 16 ? 1 : 16 / TO;     // k-steps per fragment group: KSG * TO = 16 MFMAs
    static constexpr int NG = SKS / KSG;
    static constexpr int NF = KSG * Q;                     // float4 fragments per group
    static_assert(SLICE / 4 % k16Threads == 0, "slice must split evenly over the workgroup");
    static_assert(TO % 4 == 0 && SKS % KSG == 0, "whole quads of output tiles, whole fragment groups per slice");
};

typedef __attribute__((address_space(3))) void* lds_void_ptr;
typedef const __attribute__((address_space(1))) void* glb_void_ptr;

template <int N>
__device__ __forceinline__ void wait_vmcnt() {   // all but the N youngest vector-memory operations of this wave are done
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int TK, int TO, int ACT, bool HAS_BIAS>
__device__ __forceinline__ void layer16r(const float* __restrict__ gimg, const float* __restrict__ bias, float* __restrict__ wbuf,
                                         int lane, int tid, const f32x4 (&in)[TK], f32x4 (&out)[TO]) {
    using G = Layer16Geom<TK, TO>;
    constexpr int T = TO;   // the output-side loops below run over the TO output tiles
    constexpr int Q = G::Q, SKS = G::SKS, NS = G::NS, SLICE = G::SLICE, PER = G::PER, KSG = G::KSG, NG = G::NG, NF = G::NF;
    const int g = lane >> 4;
    const int wave_base = (tid >> 6) * 256;   // floats: this wave's 1 KB piece inside every 4 KB of a slice
    // Source address = SCALAR base (the slice, made provably wave-uniform) + one 32-bit lane offset shared by every piece: the
    // global_load_lds saddr form.  Left to itself hipcc materialises a 64-bit VGPR address per piece, slice and layer, spills
    // them, and every reload's s_waitcnt vmcnt(0) drains the DMAs in flight (measured: 2.9 us per slice instead of 0.9).
    const unsigned voff = (unsigned)tid * 16u;
    auto dma_slice = [&](int slice) {
        const unsigned long long u = reinterpret_cast<unsigned long long>(gimg + slice * SLICE);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
        const char* sbase = reinterpret_cast<const char*>(((unsigned long long)hi << 32) | lo);
        float* ldst = wbuf + (slice % 3) * SLICE + wave_base;
#pragma unroll
        for (int p = 0; p < PER; ++p)
            __builtin_amdgcn_global_load_lds((glb_void_ptr)(sbase + p * (k16Threads * 16) + voff),
                                             (lds_void_ptr)(ldst + p * k16Threads * 4), 16, 0, 0);
    };
    dma_slice(0);
    if constexpr (NS > 1) dma_slice(1);
#pragma unroll
    for (int t = 0; t < T; ++t) {
        if constexpr (HAS_BIAS) out[t] = *reinterpret_cast<const f32x4*>(bias + t * 16 + 4 * g);
        else out[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if constexpr (NS > 1) wait_vmcnt<PER>(); else wait_vmcnt<0>();   // slice 0 landed (slice 1 may still fly)
    PIME16_BARRIER();
    float4 wf[2][NF];
    auto load_group = [&](int slice, int gidx, float4 (&dst)[NF]) {
        const float4* wl = reinterpret_cast<const float4*>(wbuf + (slice % 3) * SLICE) + lane;
#pragma unroll
        for (int kk = 0; kk < KSG; ++kk)
#pragma unroll
            for (int q = 0; q < Q; ++q) dst[kk * Q + q] = wl[((gidx * KSG + kk) * Q + q) * 64];
    };
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        if (s + 2 < NS) dma_slice(s + 2);
        load_group(s, 0, wf[(s * NG) & 1]);
#pragma unroll
        for (int gi = 0; gi < NG; ++gi) {
            const int cur = (s * NG + gi) & 1;
            // The group's 16 MFMAs in two halves with the NEXT group's fragment reads issued between them: hipcc waits
            // lgkmcnt(0) in front of a group's first MFMA, so reads issued right there would sit exposed; issued half a group
            // (256 pipe cycles) earlier they have landed.
#pragma unroll
            for (int half = 0; half < 2; ++half) {
#pragma unroll
                for (int m = half * (NF / 2); m < (half + 1) * (NF / 2); ++m) {
                    const int kk = m / Q, q = m % Q;
                    const int ks = s * SKS + gi * KSG + kk;
                    const float b = in[ks >> 2][ks & 3];
                    const float4 w = wf[cur][m];
                    out[4 * q + 0] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.x, b, out[4 * q + 0], 0, 0, 0);
                    out[4 * q + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.y, b, out[4 * q + 1], 0, 0, 0);
                    out[4 * q + 2] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.z, b, out[4 * q + 2], 0, 0, 0);
                    out[4 * q + 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.w, b, out[4 * q + 3], 0, 0, 0);
                }
                if (half == 0 && gi + 1 < NG) load_group(s, gi + 1, wf[cur ^ 1]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // slice s+1 must have landed before the barrier publishes it; the DMA of slice s+2 (the PER youngest) may still fly
        if (s + 2 < NS) wait_vmcnt<PER>(); else wait_vmcnt<0>();
        PIME16_BARRIER();   // slice s+1 published; slice s's reads retired (after the last slice: wbuf is free again)
    }
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) out[t][r] = act16<ACT>(out[t][r]);
}

// ---- the same chain layer on bf16 matrix instructions: every f32 operand as hi + mid + lo ("bf16x3", opt-in) ---------------------
// v_mfma_f32_16x16x32_bf16 runs at 16x the f32 MFMA rate; a product of two f32 values split into three bf16 pieces each is, to
// f32 rounding, hi*hi + (hi*mid + mid*hi) + (hi*lo + lo*hi + mid*mid): six bf16 MFMAs (accumulated in f32, smallest terms first)
// for what eight 16x16x4 f32 MFMAs compute -- 96 against 256 pipe cycles.  The dropped terms are below 2^-24 of the product
// (tools/bf16x3_layer_bench.hip: error against float64 3.1e-7 of max |y|, the f32 chain's own is 5.1e-7).
//   * weights: split once per optimizer step into three bf16 planes (pack16_b3_layer), image [slice][tile][plane][lane][8]:
//     element e of lane (i, g), k-step ks is W[16 to + i][16 (2 ks + (e >> 2)) + 4 g + (e & 3)] -- a k-step covers the input tiles
//     2 ks and 2 ks + 1, and the k order inside it is the order a lane HOLDS those two tiles' accumulators (registers 0..3 of tile
//     2 ks, then of tile 2 ks + 1), so the activation operand is again made from registers, with no lane movement;
//   * activations: split on the fly (v_cvt_pk_bf16_f32 + a subtraction per level), once per k-step;
//   * streaming as layer16r: LDS-DMA into three slice buffers, one barrier per slice; slice = one k-step x OT output tiles (24 KB
//     at OT = 8).  Every wave reads every slice, so the layer moves 1.5x the bytes of the f32 image through LDS in 3/8 of the time.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int TK, int TO = TK, int OTX = 0>
struct Layer16B3Geom {
    static_assert(TK % 2 == 0 && TO % 2 == 0, "k-steps of two input tiles; output tiles in pairs");
    static constexpr int KS = TK / 2;                       // k-steps of 32 input features
    static constexpr int OT = OTX ? OTX : (TO >= 8 ? 8 : TO);   // output tiles per slice (the image does not depend on it: [ks][to][plane])
    static constexpr int SPK = TO / OT;                     // slices per k-step
    static constexpr int NS = KS * SPK;
    static constexpr int SLICE = OT * 3 * 256;              // floats: 3 KB per output tile (three planes of 64 lanes x 16 bytes)
    static constexpr int PER = SLICE / 4 / k16Threads;      // 4 KB DMA pieces of the workgroup per slice
    static constexpr int NBUFW = NS >= 3 ? 3 : NS;
    static constexpr int IMAGE = NS * SLICE;                // floats of the whole layer image
    static_assert(SLICE / 4 % k16Threads == 0, "slice must split evenly over the workgroup");
};

__host__ __device__ inline void split_bf16x3(float v, float& hi, float& mid, float& lo) {   // the three pieces, as f32 values
    auto rne = [](float x) {   // round to nearest even at bf16 precision
        unsigned u = __builtin_bit_cast(unsigned, x);
        u = (u + 0x7fffu + ((u >> 16) & 1u)) & 0xffff0000u;
        return __builtin_bit_cast(float, u);
    };
    hi = rne(v);
    const float r1 = v - hi;
    mid = rne(r1);
    lo = rne(r1 - mid);
}

// W: nn.Linear [O][K] row-major; transposed: the layer computes V = W^T (outputs = W's columns, k = W's rows).  dst: bf16 (uint16).
__device__ inline void pack16_b3_layer(unsigned short* __restrict__ dst, const float* __restrict__ W, int K, int n_out, int n_k,
                                       bool transposed, int tid, int nthr) {
    const int TO = n_out / 16, KS = n_k / 32;
    const int OT = TO >= 8 ? 8 : TO, SPK = TO / OT;
    const int n = KS * TO * 64 * 8;   // elements of one plane
    for (int idx = tid; idx < n; idx += nthr) {
        const int e = idx & 7, lane = (idx >> 3) & 63, tt = (idx >> 9) % OT, sl = idx / (512 * OT);
        const int ks = sl / SPK, to = (sl % SPK) * OT + tt;
        const int i = lane & 15, g = lane >> 4;
        const int out = 16 * to + i, k = 16 * (2 * ks + (e >> 2)) + 4 * g + (e & 3);
        const float v = transposed ? W[(size_t)k * K + out] : W[(size_t)out * K + k];
        float p3[3];
        split_bf16x3(v, p3[0], p3[1], p3[2]);
#pragma unroll
        for (int p = 0; p < 3; ++p)
            dst[((size_t)((sl * OT + tt) * 3 + p) * 64 + lane) * 8 + e] = (unsigned short)(__builtin_bit_cast(unsigned, p3[p]) >> 16);
    }
}

// OTX: output tiles per slice (0: 8, or all of a narrower layer); NB: slice buffers in the caller's region (NB * SLICE floats) --
// slice s + NB - 1 is requested while slice s multiplies.
template <int TK, int TO, int ACT, bool HAS_BIAS, int OTX = 0, int NB = 3>
__device__ __forceinline__ void layer16r_b3(const float* __restrict__ gimg, const float* __restrict__ bias, float* __restrict__ wbuf,
                                            int lane, int tid, const f32x4 (&in)[TK], f32x4 (&out)[TO]) {
    using G = Layer16B3Geom<TK, TO, OTX>;
    constexpr int OT = G::OT, SPK = G::SPK, NS = G::NS, SLICE = G::SLICE, PER = G::PER;
    constexpr int AHEAD = (NB < NS ? NB : NS) - 1;   // slices requested beyond the one being multiplied
    static_assert(NB >= 2 && NB <= 5 && (AHEAD - 1) * PER < 64, "vmcnt is a 6-bit counter; the counted waits cover three younger slices");
    const int g = lane >> 4;
    const int wave_base = (tid >> 6) * 256;   // floats: this wave's 1 KB piece inside every 4 KB of a slice
    const unsigned voff = (unsigned)tid * 16u;
    auto dma_slice = [&](int slice) {         // (source = scalar base + one 32-bit lane offset: see layer16r)
        const unsigned long long u = reinterpret_cast<unsigned long long>(gimg + slice * SLICE);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
        const char* sbase = reinterpret_cast<const char*>(((unsigned long long)hi << 32) | lo);
        float* ldst = wbuf + (slice % NB) * SLICE + wave_base;
#pragma unroll
        for (int p = 0; p < PER; ++p)
            __builtin_amdgcn_global_load_lds((glb_void_ptr)(sbase + p * (k16Threads * 16) + voff),
                                             (lds_void_ptr)(ldst + p * k16Threads * 4), 16, 0, 0);
    };
#pragma unroll
    for (int s0 = 0; s0 < (AHEAD > 0 ? AHEAD : 1); ++s0) dma_slice(s0);
#pragma unroll
    for (int t = 0; t < TO; ++t) {
        if constexpr (HAS_BIAS) out[t] = *reinterpret_cast<const f32x4*>(bias + t * 16 + 4 * g);
        else out[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    wait_vmcnt<(AHEAD > 1 ? AHEAD - 1 : 0) * PER>();   // slice 0 landed (the younger requests may still fly)
    PIME16_BARRIER();
    bf16x8 bh, bm, bl;       // the activation operand of the current k-step
    bf16x8 wf[2][6];         // fragments of two output tiles (hi, mid, lo each), double-buffered
    auto load_pair = [&](int slice, int pr, bf16x8 (&dst)[6]) {
        const bf16x8* wl = reinterpret_cast<const bf16x8*>(wbuf + (slice % NB) * SLICE) + lane;
#pragma unroll
        for (int m = 0; m < 6; ++m) dst[m] = wl[(pr * 6 + m) * 64];
    };
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        if (AHEAD > 0 && s + AHEAD < NS) dma_slice(s + AHEAD);   // into the buffer of slice s - 1, which every wave has left
        load_pair(s, 0, wf[0]);
        if (s % SPK == 0) {   // a new k-step: input tiles 2 ks, 2 ks + 1 in three bf16 pieces
            const int ks = s / SPK;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float v = in[2 * ks + (e >> 2)][e & 3];
                const __bf16 a = (__bf16)v;
                const float r1 = v - (float)a;
                const __bf16 b = (__bf16)r1;
                bh[e] = a; bm[e] = b; bl[e] = (__bf16)(r1 - (float)b);
            }
        }
#pragma unroll
        for (int pr = 0; pr < OT / 2; ++pr) {
            const int cur = pr & 1;
            const int t0 = (s % SPK) * OT + 2 * pr;
            __builtin_amdgcn_sched_barrier(0);
            // The NEXT pair's six fragment reads go between the two halves of this pair's twelve MFMAs: hipcc waits lgkmcnt(0) in front
            // of a pair's first MFMA whatever is outstanding, so reads issued right in front of that wait sit exposed; half a pair
            // earlier they have landed (+5 %).  Hand-placed reads two pairs ahead with counted waits (asm) were bit-identical and
            // SLOWER, as were more or larger slice buffers (tools/layer16_b3_bench.hip): on random data the layer runs at ~1.7 PFLOP/s
            // of bf16 products, 70 % of the matrix peak (matrix pipe busy 0.55 - 0.71 of the cycles at an unchanged clock:
            // profiles/r04_o_layer16_b3_bench_pmc.txt), whichever of those pipelines feeds it.
            const bf16x8 (&w)[6] = wf[cur];   // [0..2] = hi, mid, lo of tile t0; [3..5] of tile t0 + 1
            f32x4 c0 = out[t0], c1 = out[t0 + 1];
            c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[2], bh, c0, 0, 0, 0);   // the small terms first
            c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[5], bh, c1, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[0], bl, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[3], bl, c1, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[1], bm, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[4], bm, c1, 0, 0, 0);
            if (pr + 1 < OT / 2) load_pair(s, pr + 1, wf[cur ^ 1]);
            __builtin_amdgcn_sched_barrier(0);
            c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[1], bh, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[4], bh, c1, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[0], bm, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[3], bm, c1, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[0], bh, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[3], bh, c1, 0, 0, 0);
            out[t0] = c0; out[t0 + 1] = c1;
            __builtin_amdgcn_sched_barrier(0);
        }
        // slice s+1 must have landed before the barrier publishes it; the requests younger than it may still fly
        {
            const int last = s + AHEAD < NS - 1 ? s + AHEAD : NS - 1;   // youngest slice requested so far
            const int younger = last > s + 1 ? last - (s + 1) : 0;
            if (younger == 0) wait_vmcnt<0>();
            else if (younger == 1) wait_vmcnt<PER>();
            else if (younger == 2) wait_vmcnt<(2 * PER < 64 ? 2 * PER : 0)>();
            else wait_vmcnt<(3 * PER < 64 ? 3 * PER : 0)>();
        }
        PIME16_BARRIER();   // slice s+1 published; slice s's reads retired (after the last slice: wbuf is free again)
    }
#pragma unroll
    for (int t = 0; t < TO; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) out[t][r] = act16<ACT>(out[t][r]);
}

template <int TK, int TO = TK, int OTX = 0, int NB = 3>
__host__ __device__ constexpr int layer16_b3_lds_floats() {
    using G = Layer16B3Geom<TK, TO, OTX>;
    return (G::NS < NB ? G::NS : NB) * G::SLICE;
}

// The bf16x3 planes of a net's streamed layers sit BEHIND its transposed f32 image (img_bwd + layoutb16(md).total): forward layers,
// then their transposes.  A layer <TK, TO> takes (TK / 2) * TO * 768 floats (1.5x its f32 image).
__host__ __device__ constexpr int b3_layer_floats(int TK, int TO) { return (TK / 2) * TO * 768; }
struct LayoutB3_16 { int w1, w2, w2t, w1t, total; };
__host__ __device__ inline LayoutB3_16 layout_b3_16(int md) {
    const int T = md / 16, I = b3_layer_floats(T, T);
    return LayoutB3_16{0, I, 2 * I, 3 * I, 4 * I};
}
struct LayoutB3_16M { int w1o, w1i, wn, wnt, w1ot, w1it, total; };
__host__ __device__ inline LayoutB3_16M layout_b3_16m(int md) {
    const int T = md / 16, I = b3_layer_floats(T, T), Ih = b3_layer_floats(T, T / 2), It = b3_layer_floats(T / 2, T);
    return LayoutB3_16M{0, Ih, 2 * Ih, 2 * Ih + I, 2 * Ih + 2 * I, 2 * Ih + 2 * I + It, 2 * Ih + 2 * I + 2 * It};
}

__global__ void pack16_b3_kernel(PackArgs a, float* __restrict__ b3) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x, nthr = gridDim.x * blockDim.x;
    const int md = a.md, hd = md / 2;
    unsigned short* const d = reinterpret_cast<unsigned short*>(b3);
    if (a.kind == MLP_MODULAR_ACTOR) {   // params: other_net.0, other_net.2 [hd][md], integrator_net.0, integrator_net.2, net.0 [md][md], net.2
        const LayoutB3_16M L = layout_b3_16m(md);
        pack16_b3_layer(d + 2 * L.w1o, a.p[2], md, hd, md, false, tid, nthr);
        pack16_b3_layer(d + 2 * L.w1i, a.p[6], md, hd, md, false, tid, nthr);
        pack16_b3_layer(d + 2 * L.wn, a.p[8], md, md, md, false, tid, nthr);
        pack16_b3_layer(d + 2 * L.wnt, a.p[8], md, md, md, true, tid, nthr);
        pack16_b3_layer(d + 2 * L.w1ot, a.p[2], md, md, hd, true, tid, nthr);   // V = W^T: md outputs, hd inputs; W's row length is md
        pack16_b3_layer(d + 2 * L.w1it, a.p[6], md, md, hd, true, tid, nthr);
    } else {
        const LayoutB3_16 L = layout_b3_16(md);
        pack16_b3_layer(d + 2 * L.w1, a.p[2], md, md, md, false, tid, nthr);
        pack16_b3_layer(d + 2 * L.w2, a.p[4], md, md, md, false, tid, nthr);
        pack16_b3_layer(d + 2 * L.w2t, a.p[4], md, md, md, true, tid, nthr);
        pack16_b3_layer(d + 2 * L.w1t, a.p[2], md, md, md, true, tid, nthr);
    }
}

// One streamed layer of a gradient kernel: the f32 image, or (B3) the bf16x3 planes.
template <int TK, int TO, int ACT, bool HAS_BIAS, bool B3>
__device__ __forceinline__ void chain16(const float* __restrict__ gimg, const float* __restrict__ b3img, const float* __restrict__ bias,
                                        float* __restrict__ wbuf, int lane, int tid, const f32x4 (&in)[TK], f32x4 (&out)[TO]) {
    if constexpr (B3) layer16r_b3<TK, TO, ACT, HAS_BIAS>(b3img, bias, wbuf, lane, tid, in, out);
    else layer16r<TK, TO, ACT, HAS_BIAS>(gimg, bias, wbuf, lane, tid, in, out);
}

template <int T, int ACT, bool HAS_BIAS>
__device__ __forceinline__ void layer16(const float* __restrict__ gimg, const float* __restrict__ bias, float* __restrict__ wbuf,
                                        int lane, int tid, const f32x4 (&in)[T], f32x4 (&out)[T]) {
    layer16r<T, T, ACT, HAS_BIAS>(gimg, bias, wbuf, lane, tid, in, out);
}

// QUAD form of a square layer (fused rollout, launches of <= 4 096 lanes: one 16-lane tile per WORKGROUP): the workgroup streams the
// slices exactly as layer16r does, but wave w computes only the quad of output tiles [4 w, 4 w + 4) -- component q = w of every
// k-step's fragments, a quarter of the MFMA chain -- and the quarters meet in an LDS exchange buffer (xb: T tiles x 64 lanes x 4
// floats), from which every wave reads the whole activation back.  A fragment group is four k-steps of the wave's quad (16 MFMAs,
// as in layer16r); per accumulator the k-steps run in the same order, so the result is bit-identical to layer16r's.
template <int T, int ACT>
__device__ __forceinline__ void layer16q(const float* __restrict__ gimg, const float* __restrict__ bias, float* __restrict__ wbuf,
                                         float* __restrict__ xb, int lane, int tid, const f32x4 (&in)[T], f32x4 (&out)[T]) {
    using G = Layer16Geom<T, T>;
    static_assert(G::Q == k16Waves && G::SKS % 4 == 0, "one quad of output tiles per wave; whole groups of four k-steps per slice");
    constexpr int Q = G::Q, SKS = G::SKS, NS = G::NS, SLICE = G::SLICE, PER = G::PER, NG = SKS / 4;
    const int g = lane >> 4, wave = tid >> 6;
    const int wave_base = wave * 256;
    const unsigned voff = (unsigned)tid * 16u;
    auto dma_slice = [&](int slice) {
        const unsigned long long u = reinterpret_cast<unsigned long long>(gimg + slice * SLICE);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
        const char* sbase = reinterpret_cast<const char*>(((unsigned long long)hi << 32) | lo);
        float* ldst = wbuf + (slice % 3) * SLICE + wave_base;
#pragma unroll
        for (int p = 0; p < PER; ++p)
            __builtin_amdgcn_global_load_lds((glb_void_ptr)(sbase + p * (k16Threads * 16) + voff),
                                             (lds_void_ptr)(ldst + p * k16Threads * 4), 16, 0, 0);
    };
    dma_slice(0);
    if constexpr (NS > 1) dma_slice(1);
    f32x4 o4[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) o4[j] = *reinterpret_cast<const f32x4*>(bias + (4 * wave + j) * 16 + 4 * g);
    if constexpr (NS > 1) wait_vmcnt<PER>(); else wait_vmcnt<0>();
    PIME16_BARRIER();
    float4 wf[2][4];
    auto load_group = [&](int slice, int gidx, float4 (&dst)[4]) {
        const float4* wl = reinterpret_cast<const float4*>(wbuf + (slice % 3) * SLICE) + lane;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) dst[kk] = wl[((gidx * 4 + kk) * Q + wave) * 64];
    };
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        if (s + 2 < NS) dma_slice(s + 2);
        load_group(s, 0, wf[(s * NG) & 1]);
#pragma unroll
        for (int gi = 0; gi < NG; ++gi) {
            const int cur = (s * NG + gi) & 1;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
#pragma unroll
                for (int m = half * 2; m < half * 2 + 2; ++m) {
                    const int ks = s * SKS + gi * 4 + m;
                    const float b = in[ks >> 2][ks & 3];
                    const float4 w = wf[cur][m];
                    o4[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.x, b, o4[0], 0, 0, 0);
                    o4[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.y, b, o4[1], 0, 0, 0);
                    o4[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.z, b, o4[2], 0, 0, 0);
                    o4[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.w, b, o4[3], 0, 0, 0);
                }
                if (half == 0 && gi + 1 < NG) load_group(s, gi + 1, wf[cur ^ 1]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (s + 2 < NS) wait_vmcnt<PER>(); else wait_vmcnt<0>();
        PIME16_BARRIER();
    }
    // the wave's quad -> the exchange buffer; everyone reads the whole activation back (the layer's last barrier above ordered the
    // previous readers of xb: they read it before they entered this layer's slice loop)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int r = 0; r < 4; ++r) o4[j][r] = act16<ACT>(o4[j][r]);
        *reinterpret_cast<f32x4*>(xb + ((4 * wave + j) * 64 + lane) * 4) = o4[j];
    }
    PIME16_BARRIER();
#pragma unroll
    for (int t = 0; t < T; ++t) out[t] = *reinterpret_cast<const f32x4*>(xb + (t * 64 + lane) * 4);
}

// QUAD form of the modular actor's 2:1 tower layers (TO = 8 output tiles at width 256: Q = 2 quads): wave w computes the two output
// tiles [2 w, 2 w + 2) -- the .xy or .zw half of quad w >> 1 of every k-step's fragments, eight k-steps per group (16 MFMAs) -- and
// writes them, activated, into tiles tile0 + 2 w, tile0 + 2 w + 1 of the exchange buffer; the caller barriers and reads.
template <int TK, int TO, int ACT>
__device__ __forceinline__ void layer16rq(const float* __restrict__ gimg, const float* __restrict__ bias, float* __restrict__ wbuf,
                                          float* __restrict__ xb, int tile0, int lane, int tid, const f32x4 (&in)[TK]) {
    using G = Layer16Geom<TK, TO>;
    static_assert(G::Q * 2 == k16Waves && G::SKS % 8 == 0, "two output tiles per wave; whole groups of eight k-steps per slice");
    constexpr int Q = G::Q, SKS = G::SKS, NS = G::NS, SLICE = G::SLICE, PER = G::PER, NG = SKS / 8;
    const int g = lane >> 4, wave = tid >> 6, q = wave >> 1, hf = wave & 1;
    const int wave_base = wave * 256;
    const unsigned voff = (unsigned)tid * 16u;
    auto dma_slice = [&](int slice) {
        const unsigned long long u = reinterpret_cast<unsigned long long>(gimg + slice * SLICE);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
        const char* sbase = reinterpret_cast<const char*>(((unsigned long long)hi << 32) | lo);
        float* ldst = wbuf + (slice % 3) * SLICE + wave_base;
#pragma unroll
        for (int p = 0; p < PER; ++p)
            __builtin_amdgcn_global_load_lds((glb_void_ptr)(sbase + p * (k16Threads * 16) + voff),
                                             (lds_void_ptr)(ldst + p * k16Threads * 4), 16, 0, 0);
    };
    dma_slice(0);
    if constexpr (NS > 1) dma_slice(1);
    f32x4 o2[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) o2[j] = *reinterpret_cast<const f32x4*>(bias + (2 * wave + j) * 16 + 4 * g);
    if constexpr (NS > 1) wait_vmcnt<PER>(); else wait_vmcnt<0>();
    PIME16_BARRIER();
    float2 wf[2][8];
    auto load_group = [&](int slice, int gidx, float2 (&dst)[8]) {
        const float2* wl = reinterpret_cast<const float2*>(wbuf + (slice % 3) * SLICE) + 2 * lane + hf;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) dst[kk] = wl[((gidx * 8 + kk) * Q + q) * 128];
    };
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        if (s + 2 < NS) dma_slice(s + 2);
        load_group(s, 0, wf[(s * NG) & 1]);
#pragma unroll
        for (int gi = 0; gi < NG; ++gi) {
            const int cur = (s * NG + gi) & 1;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
#pragma unroll
                for (int m = half * 4; m < half * 4 + 4; ++m) {
                    const int ks = s * SKS + gi * 8 + m;
                    const float b = in[ks >> 2][ks & 3];
                    const float2 w = wf[cur][m];
                    o2[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.x, b, o2[0], 0, 0, 0);
                    o2[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.y, b, o2[1], 0, 0, 0);
                }
                if (half == 0 && gi + 1 < NG) load_group(s, gi + 1, wf[cur ^ 1]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (s + 2 < NS) wait_vmcnt<PER>(); else wait_vmcnt<0>();
        PIME16_BARRIER();
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
#pragma unroll
        for (int r = 0; r < 4; ++r) o2[j][r] = act16<ACT>(o2[j][r]);
        *reinterpret_cast<f32x4*>(xb + ((tile0 + 2 * wave + j) * 64 + lane) * 4) = o2[j];
    }
}

template <int TK, int TO = TK>
__host__ __device__ constexpr int layer16_lds_floats() {
    return Layer16Geom<TK, TO>::NBUFW * Layer16Geom<TK, TO>::SLICE;
}

// First layer from the LDS-resident natural-order image: h = act(W0 x + b0), x[s][4 ks + g] in xr[ks].
template <int T, int ACT>
__device__ __forceinline__ void first16(const float* __restrict__ w0, const float* __restrict__ b0, int KS0, int lane,
                                        const float (&xr)[8], f32x4 (&out)[T]) {
    constexpr int Q = T / 4;
    const int g = lane >> 4;
#pragma unroll
    for (int t = 0; t < T; ++t) out[t] = *reinterpret_cast<const f32x4*>(b0 + t * 16 + 4 * g);
    const float4* wl = reinterpret_cast<const float4*>(w0) + lane;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
        if (ks < KS0) {
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                const float4 w = wl[(ks * Q + q) * 64];
                out[4 * q + 0] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.x, xr[ks], out[4 * q + 0], 0, 0, 0);
                out[4 * q + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.y, xr[ks], out[4 * q + 1], 0, 0, 0);
                out[4 * q + 2] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.z, xr[ks], out[4 * q + 2], 0, 0, 0);
                out[4 * q + 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.w, xr[ks], out[4 * q + 3], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) out[t][r] = act16<ACT>(out[t][r]);
}

// y = w3 . h + b3: every lane sums its 4T features, the four feature groups of a sample are lanes l, l^16, l^32, l^48.
template <int T>
__device__ __forceinline__ float head16(const float* __restrict__ w3, float b3, int lane, const f32x4 (&h)[T]) {
    const int g = lane >> 4;
    float acc = 0.f;
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const f32x4 w = *reinterpret_cast<const f32x4*>(w3 + t * 16 + 4 * g);
#pragma unroll
        for (int r = 0; r < 4; ++r) acc = fmaf(h[t][r], w[r], acc);
    }
    acc += __shfl_xor(acc, 16);
    acc += __shfl_xor(acc, 32);
    return acc + b3;
}

// Sum over the 16 lanes of this lane's row (= the 16 samples of the tile at fixed feature group); valid in lane 15 of the row.
__device__ __forceinline__ float row_sum16(float v) {
#define PIME16_DPP(x, ctrl) x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), ctrl, 0xf, 0xf, true))
    PIME16_DPP(v, 0x111);  // row_shr:1
    PIME16_DPP(v, 0x112);  // row_shr:2
    PIME16_DPP(v, 0x114);  // row_shr:4
    PIME16_DPP(v, 0x118);  // row_shr:8
#undef PIME16_DPP
    return v;
}

// ---- weight gradient of one layer: dW[a][b] = sum over the group's 64 samples of A[s][a] B[s][b] ----------------------------
// Rounds of RT tiles (RT = 4: ONE round, every wave publishes its tile's A rows and B rows into the sample-major image X, one
// barrier, 16 k-steps; RT = 2 at width 256, where the 64-row image does not fit: two rounds, waves 2t and 2t+1 publish round
// t); every wave owns an NA x NB patch of 16x16 output blocks per pass.  The operand reads of the next k-step are issued
// half-way through a k-step's MFMAs.
template <int TA, int TB>
struct Dw16Plan {
    static constexpr int NBLK = TA * TB;
    static constexpr int TOT = (NBLK + k16Waves - 1) / k16Waves;
    static constexpr int NB = TB >= 4 ? 4 : TB;
    static constexpr int PERP = TOT < 16 ? TOT : 16;
    static constexpr int NA = PERP / NB > 0 ? PERP / NB : 1;
    static constexpr int PR = TB / NB;                        // patches per row of A tiles
    static constexpr int NPATCH = (TA / NA) * PR;
    static constexpr int PASSES = (NPATCH + k16Waves - 1) / k16Waves;
    // Row pitch = 20 (mod 32) floats: the publishers' ds_write_b128 (8 consecutive rows per LDS pass, 4 banks each) hit 32
    // distinct banks; the operand ds_read_b32 (rows 4ks + {0, 1} in one 32-lane pass) conflict 2-way on 4 lanes of 32, which
    // costs nothing beside 16 MFMAs per k-step.  (Pitch = 16 mod 32: reads clean, writes 4-way: 0.7 us per round.)
    static constexpr int PA = TA * 16 + 20, PB = TB * 16 + 20;
    static_assert(TA % NA == 0 && TB % NB == 0, "patches must tile the block grid");
};

template <int TA, int TB, int RT, class PubA, class PubB>
__device__ __forceinline__ void dw16(float* __restrict__ X, int lane, int wave, const PubA& pub_a, const PubB& pub_b,
                                     float* __restrict__ gW, float* __restrict__ gb, bool accum) {
    using P = Dw16Plan<TA, TB>;
    constexpr int NA = P::NA, NB = P::NB, ROWS = RT * 16, NKS = RT * 4, ROUNDS = k16Waves / RT;
    const int i = lane & 15, kk = lane >> 4;
#pragma unroll 1
    for (int pass = 0; pass < P::PASSES; ++pass) {
        const int patch = pass * k16Waves + wave;
        const bool active = patch < P::NPATCH;
        const int a0 = active ? (patch / P::PR) * NA : 0, b0 = active ? (patch % P::PR) * NB : 0;
        // The accumulators START from what an earlier group of this workgroup stored (zero for its first group): the slab
        // loads are issued here and land behind the publish / barriers, instead of a load -> add -> store chain at the end.
        f32x4 acc[NA][NB];
        f32x4* q[NA][NB];
        float bsum[NA];
#pragma unroll
        for (int x = 0; x < NA; ++x) {
            bsum[x] = 0.f;
#pragma unroll
            for (int y = 0; y < NB; ++y) {
                int off = (((a0 + x) * TB + b0 + y) * 64 + lane) * 4;   // slab_layout16: block-major, accumulator order
                asm volatile("" : "+v"(off));
                q[x][y] = reinterpret_cast<f32x4*>(gW + off);
                acc[x][y] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (accum && active) acc[x][y] = *q[x][y];
            }
        }
#pragma unroll 1
        for (int t = 0; t < ROUNDS; ++t) {
            PIME16_BARRIER();   // X free: the previous round / pass / job / chain layer is done with the region
            if (wave / RT == t) {
                pub_a(X + ((wave % RT) * 16 + i) * P::PA);
                pub_b(X + ROWS * P::PA + ((wave % RT) * 16 + i) * P::PB);
            }
            PIME16_BARRIER();
            PIME_NO_HOIST();
            if (active) {
                const float* Ap = X + kk * P::PA + a0 * 16 + i;
                const float* Bp = X + ROWS * P::PA + kk * P::PB + b0 * 16 + i;
                float av[2][NA], bv[2][NB];   // operands of k-step ks+1 are in flight while k-step ks multiplies
                auto load_ops = [&](int ks, float (&a_)[NA], float (&b_)[NB]) {
#pragma unroll
                    for (int x = 0; x < NA; ++x) a_[x] = Ap[4 * ks * P::PA + x * 16];
#pragma unroll
                    for (int y = 0; y < NB; ++y) b_[y] = Bp[4 * ks * P::PB + y * 16];
                };
                load_ops(0, av[0], bv[0]);
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks) {
#pragma unroll
                    for (int x = 0; x < NA; ++x) {
#pragma unroll
                        for (int y = 0; y < NB; ++y)
                            acc[x][y] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ks & 1][x], bv[ks & 1][y], acc[x][y], 0, 0, 0);
                        bsum[x] += av[ks & 1][x];
                        if (x == (NA - 1) / 2) {   // half-way: the next k-step's operand reads, landed by the time they are waited for
                            if (ks + 1 < NKS) load_ops(ks + 1, av[(ks + 1) & 1], bv[(ks + 1) & 1]);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        if (active) {
            // one 16-byte store per lane and block: a wave writes 1 KB contiguous
#pragma unroll
            for (int x = 0; x < NA; ++x) {
#pragma unroll
                for (int y = 0; y < NB; ++y) *q[x][y] = acc[x][y];
                float bs = bsum[x];
                bs += __shfl_xor(bs, 16);
                bs += __shfl_xor(bs, 32);
                if (gb && b0 == 0 && kk == 0) {
                    float* qd = &gb[(a0 + x) * 16 + i];
                    *qd = accum ? *qd + bs : bs;
                }
            }
        }
    }
}

// ---- width 256: the weight gradient of a T x T layer with the operands published ONCE ---------------------------------------
// dw16<16, 16, 2> re-publishes both operands in each of its four passes (two rounds each: the 64-row image of both operands does
// not fit beside nothing), and in a round only two of the four waves publish while the others wait at the barrier: with one
// wave per SIMD and ~30 SIMD cycles per ds_write_b128 (tools/dw_round_bench.hip) that was ~45 % of the job (r02_i trace: 30-33 us
// per job against 13.7 us of MFMAs).  Here pass p needs the SAME 64 A features in every wave (a0 = 4p) and each wave its own 64 B
// features (b0 = 4 wave) in every pass: the whole B image (64 samples x 256 features) is published once, the A image one
// 64-feature slice per pass into two alternating buffers (the slice of pass p+1 is written while pass p multiplies): every wave
// publishes its own 16 samples at the same time (no skew), 6 barriers per job instead of 16.
// Both images keep the four tiles of a 64-feature group INTERLEAVED (position 4 i + x for feature 16 x + i): a publishing lane
// holds exactly those four values (v[4q+x][r], x = 0..3) -> one ds_write_b128; a reading lane gets its four A (B) operands of a
// k-step as ONE ds_read_b128 instead of four ds_read_b32 (LDS instructions cost the SIMD issue slots, not bytes).
template <int T>
struct Dw16Sliced {
    static_assert(T == 4 * k16Waves, "one 4-tile B patch per wave");
    static constexpr int NA = 4, NB = 4, PASSES = T / NA, ROWS = k16Waves * 16, NKS = ROWS / 4;
    // row pitch = 4 (mod 32) floats: eight consecutive rows of one ds_write_b128 pass land on 32 distinct banks; the b128 operand
    // reads are contiguous over eight lanes whatever the pitch
    static constexpr int PB = T * 16 + 4, PAS = NA * 16 + 4;
    static constexpr int FLOATS = ROWS * PB + 2 * ROWS * PAS;
};

template <int T>
__device__ __forceinline__ void dw16_sliced(float* __restrict__ X, int lane, int wave, const f32x4 (&va)[T], const f32x4 (&vb)[T],
                                            float* __restrict__ gW, float* __restrict__ gb, bool accum, long long* tr = nullptr) {
    using P = Dw16Sliced<T>;
    constexpr int NA = P::NA, NB = P::NB, PB = P::PB, PAS = P::PAS, NKS = P::NKS;
    const int i = lane & 15, kk = lane >> 4;   // reading: feature i of a block, k index kk; publishing: sample i, register group kk
    float* const Bimg = X;
    float* const Aimg = X + P::ROWS * PB;
    const int row = wave * 16 + i;
    auto pub_a = [&](int p, float* buf) {   // tiles 4p .. 4p+3 of this lane's sample
        float* rp = buf + row * PAS + 16 * kk;
#pragma unroll
        for (int r = 0; r < 4; ++r)
            *reinterpret_cast<f32x4*>(rp + 4 * r) = f32x4{va[4 * p][r], va[4 * p + 1][r], va[4 * p + 2][r], va[4 * p + 3][r]};
    };
    auto slab_ptr = [&](int p, int x, int y) {
        int off = (((p * NA + x) * T + wave * NB + y) * 64 + lane) * 4;   // slab_layout16: block-major, accumulator order
        asm volatile("" : "+v"(off));
        return reinterpret_cast<f32x4*>(gW + off);
    };
    if (tr && lane == 0 && wave == 0) tr[0] = wall_clock64();
    PIME16_BARRIER();   // X free
    {
        float* rp = Bimg + row * PB + 16 * kk;
#pragma unroll
        for (int q = 0; q < T / 4; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                *reinterpret_cast<f32x4*>(rp + 64 * q + 4 * r) = f32x4{vb[4 * q][r], vb[4 * q + 1][r], vb[4 * q + 2][r], vb[4 * q + 3][r]};
    }
    pub_a(0, Aimg);
    // accumulators start from what an earlier group of this workgroup stored; the loads for pass p+1 are issued during pass p
    f32x4 nxt[NA][NB];
    if (accum) {
#pragma unroll
        for (int x = 0; x < NA; ++x)
#pragma unroll
            for (int y = 0; y < NB; ++y) nxt[x][y] = *slab_ptr(0, x, y);
    }
    PIME16_BARRIER();
#pragma unroll
    for (int p = 0; p < P::PASSES; ++p) {
        if (p + 1 < P::PASSES) pub_a(p + 1, Aimg + ((p + 1) & 1) * P::ROWS * PAS);
        f32x4 acc[NA][NB];
#pragma unroll
        for (int x = 0; x < NA; ++x)
#pragma unroll
            for (int y = 0; y < NB; ++y) acc[x][y] = accum ? nxt[x][y] : f32x4{0.f, 0.f, 0.f, 0.f};
        if (accum && p + 1 < P::PASSES) {
#pragma unroll
            for (int x = 0; x < NA; ++x)
#pragma unroll
                for (int y = 0; y < NB; ++y) nxt[x][y] = *slab_ptr(p + 1, x, y);
        }
        const float* Ap = Aimg + (p & 1) * P::ROWS * PAS + kk * PAS + 4 * i;
        const float* Bp = Bimg + kk * PB + wave * (NB * 16) + 4 * i;
        f32x4 av[2], bv[2];
        f32x4 bsum = f32x4{0.f, 0.f, 0.f, 0.f};
        av[0] = *reinterpret_cast<const f32x4*>(Ap);
        bv[0] = *reinterpret_cast<const f32x4*>(Bp);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
#pragma unroll
            for (int x = 0; x < NA; ++x) {
#pragma unroll
                for (int y = 0; y < NB; ++y)
                    acc[x][y] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ks & 1][x], bv[ks & 1][y], acc[x][y], 0, 0, 0);
                if (x == 1) {   // half-way: the next k-step's two operand reads
                    if (ks + 1 < NKS) {
                        av[(ks + 1) & 1] = *reinterpret_cast<const f32x4*>(Ap + 4 * (ks + 1) * PAS);
                        bv[(ks + 1) & 1] = *reinterpret_cast<const f32x4*>(Bp + 4 * (ks + 1) * PB);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (wave == 0) bsum += av[ks & 1];   // bias gradient: the column sums of A, needed once per A block (wave 0: b0 = 0)
            __builtin_amdgcn_sched_barrier(0);
        }
        // one 16-byte store per lane and block (a wave writes 1 KB contiguous).  The 16 stores of a pass are 16.8 MB chip-wide
        // when all workgroups reach them together and take ~2.3 us to leave (r02 sub-marks): the width-256 gradient is bound by
        // this slab traffic (256 KB per 64-sample group and layer); spreading them over the next pass's k-steps only moves the
        // stall into the MFMA loop (tried: 1 149 -> 1 163 us).
#pragma unroll
        for (int x = 0; x < NA; ++x) {
#pragma unroll
            for (int y = 0; y < NB; ++y) *slab_ptr(p, x, y) = acc[x][y];
        }
        if (gb && wave == 0) {
#pragma unroll
            for (int x = 0; x < NA; ++x) {
                float bs = bsum[x];
                bs += __shfl_xor(bs, 16);
                bs += __shfl_xor(bs, 32);
                if (kk == 0) {
                    float* qd = &gb[(p * NA + x) * 16 + i];
                    *qd = accum ? *qd + bs : bs;
                }
            }
        }
        PIME16_BARRIER();   // pass p's reads are done (its A buffer may be rewritten); slice p+1 is visible
        if (tr && lane == 0 && wave == 0) tr[1 + p] = wall_clock64();
    }
}

// ---- width 256, bf16x3: the same weight gradient on v_mfma_f32_16x16x32_bf16 ----------------------------------------------------
// dW[a][b] = sum over the group's 64 samples of A[s][a] B[s][b] contracts over the SAMPLE axis, which sits on the lanes of both
// operands: an MFMA wants it in the elements of a lane (8 consecutive k).  Both operands go through a sample-major bf16 image in LDS
// (three planes: hi, mid, lo; a lane writes its four consecutive features of a tile as ONE ds_write_b64 per plane) and come back
// through ds_read_b64_tr_b16, the transposing read: per 16-lane group a block of 4 samples x 16 features, lane i receives feature
// i's four samples -- two reads make the 8-sample fragment of a k-step.  Two k-steps of 32 samples x six products = 12 MFMAs
// (192 pipe cycles) per 16 x 16 block, where the f32 path runs 16 (512 cycles).
// Images: B [plane][sample 0..63][16 tiles x 32 bytes] = 96 KB, published once; A one 4-tile slice per pass [plane][sample][4 x 32 B]
// = 24 KB (single buffer: with both images padded or double-buffered the map does not fit beside a 32 KB first-layer image).
// Rows are unpadded; the 32-byte chunk (= tile) index is XOR-swizzled with row bits so that the eight rows a 32-lane half reads
// (samples 4 hh + q and 8 + 4 hh + q) fall on distinct banks, and the 8-byte piece inside a chunk with row bits 2..3 so that the
// publishers' ds_write_b64 (16 rows x 2 pieces per 32-lane half) are 2-way instead of 4-way (the transposing read takes whatever
// address a lane supplies for "row q, columns 4p .. 4p+3").  Patch assignment and slab layout as dw16_sliced.
typedef short s16x4 __attribute__((ext_vector_type(4)));
template <int T>
struct Dw16SlicedB3 {
    static_assert(T == 4 * k16Waves, "one 4-tile B patch per wave");
    static constexpr int NA = 4, NB = 4, PASSES = T / NA, ROWS = k16Waves * 16;
    static constexpr int BROW = T * 32, AROW = NA * 32;                       // bytes per sample row
    static constexpr int BBYTES = 3 * ROWS * BROW, ABYTES = 3 * ROWS * AROW;
    static constexpr int FLOATS = (BBYTES + ABYTES) / 4;
    // byte offset of (plane, sample row, tile, byte inside the tile's 32)
    __device__ static __forceinline__ int boff(int plane, int row, int tile, int sub) {
        return (plane * ROWS + row) * BROW + ((tile ^ ((row & 3) | ((row >> 1) & 4))) << 5) + (sub ^ (((row >> 2) & 3) << 3));
    }
    __device__ static __forceinline__ int aoff(int plane, int row, int tile4, int sub) {
        return (plane * ROWS + row) * AROW + ((tile4 ^ (((row >> 1) & 1) | ((row >> 2) & 2))) << 5) + (sub ^ (((row >> 2) & 3) << 3));
    }
};

// four f32 -> the hi / mid / lo bf16 pieces, packed (element r in bits 16 r .. 16 r + 15)
__device__ __forceinline__ void split4_bf16x3(const f32x4& v, uint2& hi, uint2& mid, uint2& lo) {
    unsigned short h[4], m[4], l[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const __bf16 a = (__bf16)v[r];
        const float r1 = v[r] - (float)a;
        const __bf16 b = (__bf16)r1;
        const __bf16 c = (__bf16)(r1 - (float)b);
        h[r] = __builtin_bit_cast(unsigned short, a); m[r] = __builtin_bit_cast(unsigned short, b); l[r] = __builtin_bit_cast(unsigned short, c);
    }
    hi = make_uint2(h[0] | ((unsigned)h[1] << 16), h[2] | ((unsigned)h[3] << 16));
    mid = make_uint2(m[0] | ((unsigned)m[1] << 16), m[2] | ((unsigned)m[3] << 16));
    lo = make_uint2(l[0] | ((unsigned)l[1] << 16), l[2] | ((unsigned)l[3] << 16));
}

template <int T>
__device__ __forceinline__ void dw16_sliced_b3(float* __restrict__ X, int lane, int wave, const f32x4 (&va)[T], const f32x4 (&vb)[T],
                                               float* __restrict__ gW, float* __restrict__ gb, bool accum) {
    using P = Dw16SlicedB3<T>;
    constexpr int NA = P::NA, NB = P::NB;
    char* const Bimg = reinterpret_cast<char*>(X);
    char* const Aimg = Bimg + P::BBYTES;
    const int s = lane & 15, g = lane >> 4;   // publishing: sample s of this wave's tile, register group g
    const int row = wave * 16 + s;
    const int q = s >> 2, p4 = s & 3;         // reading: row q, columns 4 p4 .. of a 4 x 16 block; the lane receives feature s
    auto pub_a = [&](int p) {                 // tiles 4p .. 4p+3 of this lane's sample, three planes
#pragma unroll
        for (int x = 0; x < NA; ++x) {
            uint2 hi, mid, lo;
            split4_bf16x3(va[4 * p + x], hi, mid, lo);
            *reinterpret_cast<uint2*>(Aimg + P::aoff(0, row, x, 8 * g)) = hi;
            *reinterpret_cast<uint2*>(Aimg + P::aoff(1, row, x, 8 * g)) = mid;
            *reinterpret_cast<uint2*>(Aimg + P::aoff(2, row, x, 8 * g)) = lo;
        }
    };
    auto slab_ptr = [&](int p, int x, int y) {
        int off = (((p * NA + x) * T + wave * NB + y) * 64 + lane) * 4;   // slab_layout16: block-major, accumulator order
        asm volatile("" : "+v"(off));
        return reinterpret_cast<f32x4*>(gW + off);
    };
    // the 8-sample fragment (k-step ks) of feature s of a tile, one plane: two transposing reads
    auto frag = [&](const char* img, int byte_off0, int byte_off1) {
        const s16x4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(img + byte_off0));
        const s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(img + byte_off1));
        typedef short s16x8 __attribute__((ext_vector_type(8)));
        const s16x8 v = __builtin_shufflevector(lo4, hi4, 0, 1, 2, 3, 4, 5, 6, 7);
        return __builtin_bit_cast(bf16x8, v);
    };
    PIME16_BARRIER();   // X free
#pragma unroll
    for (int t = 0; t < T; ++t) {
        uint2 hi, mid, lo;
        split4_bf16x3(vb[t], hi, mid, lo);
        *reinterpret_cast<uint2*>(Bimg + P::boff(0, row, t, 8 * g)) = hi;
        *reinterpret_cast<uint2*>(Bimg + P::boff(1, row, t, 8 * g)) = mid;
        *reinterpret_cast<uint2*>(Bimg + P::boff(2, row, t, 8 * g)) = lo;
    }
    const bf16x8 ones = {(__bf16)1.f, (__bf16)1.f, (__bf16)1.f, (__bf16)1.f, (__bf16)1.f, (__bf16)1.f, (__bf16)1.f, (__bf16)1.f};
#pragma unroll   // (p indexes the register array va: a rolled loop would put it in scratch)
    for (int p = 0; p < P::PASSES; ++p) {
        if (p > 0) PIME16_BARRIER();   // the previous pass's reads of the A slice are done
        pub_a(p);
        f32x4 acc[NA][NB];
#pragma unroll
        for (int x = 0; x < NA; ++x)
#pragma unroll
            for (int y = 0; y < NB; ++y) acc[x][y] = accum ? *slab_ptr(p, x, y) : f32x4{0.f, 0.f, 0.f, 0.f};
        f32x4 bacc = f32x4{0.f, 0.f, 0.f, 0.f};   // bias gradient of A tile 4p + wave: column sums, as a product with an all-ones operand
        PIME16_BARRIER();              // the slice (first pass: and the B image) is visible
        // eight steps (k-step ks, A tile x); the A fragments of step + 1 are requested one block (six MFMAs) into step: hipcc waits for
        // every LDS read in flight in front of the first MFMA that uses a fresh fragment, so a request right in front of its use
        // is a fully exposed LDS round trip -- ten per pass before this ordering, four now
        bf16x8 af[2][3], bf[NB][3];
        auto load_a = [&](int step, bf16x8 (&dst)[3]) {
            const int r0 = 32 * (step >> 2) + 8 * g + q, x = step & 3;
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) dst[pl] = frag(Aimg, P::aoff(pl, r0, x, 8 * p4), P::aoff(pl, r0 + 4, x, 8 * p4));
        };
        load_a(0, af[0]);
#pragma unroll
        for (int step = 0; step < 2 * NA; ++step) {
            const int ks = step >> 2, x = step & 3, cur = step & 1;
            if (x == 0) {
                const int r0 = 32 * ks + 8 * g + q;
#pragma unroll
                for (int y = 0; y < NB; ++y)
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl)
                        bf[y][pl] = frag(Bimg, P::boff(pl, r0, wave * NB + y, 8 * p4), P::boff(pl, r0 + 4, wave * NB + y, 8 * p4));
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int y = 0; y < NB; ++y) {
                f32x4 c = acc[x][y];
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[cur][2], bf[y][0], c, 0, 0, 0);   // the small terms first
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[cur][0], bf[y][2], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[cur][1], bf[y][1], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[cur][1], bf[y][0], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[cur][0], bf[y][1], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[cur][0], bf[y][0], c, 0, 0, 0);
                acc[x][y] = c;
                if (y == 0) {
                    if (step + 1 < 2 * NA) load_a(step + 1, af[cur ^ 1]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (gb && x == wave) {
                bacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[cur][2], ones, bacc, 0, 0, 0);
                bacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[cur][1], ones, bacc, 0, 0, 0);
                bacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[cur][0], ones, bacc, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // one 16-byte store per lane and block (a wave writes 1 KB contiguous)
#pragma unroll
        for (int x = 0; x < NA; ++x)
#pragma unroll
            for (int y = 0; y < NB; ++y) *slab_ptr(p, x, y) = acc[x][y];
        if (gb && s == 0) {   // column 0 of the all-ones product: features 16 (4p + wave) + 4g + r in register r
            f32x4* qd = reinterpret_cast<f32x4*>(gb + (p * NA + wave) * 16 + 4 * g);
            *qd = accum ? *qd + bacc : bacc;
        }
    }
    PIME16_BARRIER();   // the images may be rewritten
}

template <int TA, int TB, int RT>
__host__ __device__ constexpr int dw16_lds_floats() { return RT * 16 * (Dw16Plan<TA, TB>::PA + Dw16Plan<TA, TB>::PB); }

// publishers: a tile in accumulator layout (lane (s, g): features 16t + 4g + r) -> its row of the sample-major image
template <int TT>
struct PubAcc16 {
    const f32x4 (&v)[TT];
    int g;
    __device__ __forceinline__ void operator()(float* row) const {
#pragma unroll
        for (int t = 0; t < TT; ++t) *reinterpret_cast<f32x4*>(row + t * 16 + 4 * g) = v[t];
    }
};
// the raw state: lane (s, g) holds x[s][4 ks + g]; columns >= Dp are zero-filled up to the tile edge
struct PubX16 {
    const float (&xr)[8];
    int g, KS0, cols;
    __device__ __forceinline__ void operator()(float* row) const {
#pragma unroll
        for (int ks = 0; ks < 8; ++ks)
            if (ks < KS0) row[4 * ks + g] = xr[ks];
        for (int c = 4 * KS0 + g; c < cols; c += 4) row[c] = 0.f;
    }
};

// ---- LDS map ---------------------------------------------------------------------------------------------------------
struct Lds16 {
    int w0, b0, b1, b2, w3, b3, wsum, region, total;
};
template <int T>
__host__ __device__ inline Lds16 lds16(int D, bool grad) {
    const int md = T * 16;
    const Layout16 L = layout16(D, md);
    Lds16 S{};
    int o = 0;
    auto seg = [&](int& f, int n) { f = o; o += (n + 3) & ~3; };
    seg(S.w0, L.KS0 * T * 64);
    seg(S.b0, md); seg(S.b1, md); seg(S.b2, md); seg(S.w3, md); seg(S.b3, 4);
    seg(S.wsum, k16Waves * 6 * 2);
    int region = layer16_lds_floats<T>();
    if (grad) {
        constexpr int RT = T <= 8 ? 4 : 2;
        int dwf = dw16_lds_floats<T, T, RT>();
        if constexpr (T == 4 * k16Waves) {
            dwf = dwf > Dw16Sliced<T>::FLOATS ? dwf : Dw16Sliced<T>::FLOATS;
            dwf = dwf > Dw16SlicedB3<T>::FLOATS ? dwf : Dw16SlicedB3<T>::FLOATS;   // (bf16x3 variant: 120 KB; one workgroup per CU at this width anyway)
        }
        region = region > dwf ? region : dwf;
        const int hacc = k16Waves * md;
        region = region > hacc ? region : hacc;
        if constexpr (T >= 8) region = region > layer16_b3_lds_floats<T>() ? region : layer16_b3_lds_floats<T>();   // (never larger today)
    }
    seg(S.region, region);
    S.total = o;
    return S;
}

template <int T>
__device__ __forceinline__ void stage_small16(float* lds, const Lds16& S, const float* __restrict__ img, const Layout16& L, int tid) {
    const int md = T * 16;
    const int n0 = L.KS0 * T * 64;
    for (int e = tid; e < n0 / 4; e += k16Threads) reinterpret_cast<float4*>(lds + S.w0)[e] = reinterpret_cast<const float4*>(img + L.w0)[e];
    for (int e = tid; e < md; e += k16Threads) {
        lds[S.b0 + e] = img[L.b0 + e];
        lds[S.b1 + e] = img[L.b1 + e];
        lds[S.b2 + e] = img[L.b2 + e];
        lds[S.w3 + e] = img[L.w3 + e];
    }
    if (tid < 4) lds[S.b3 + tid] = img[L.b3 + tid];
}

__device__ __forceinline__ void gather_x16(const float* __restrict__ xrow, int D, int KS0, int g, float (&xr)[8]) {
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
        const int c = 4 * ks + g;
        xr[ks] = (ks < KS0 && c < D) ? xrow[c] : 0.f;
    }
}

// ---- forward only: value pass / policy mean --------------------------------------------------------------------------
template <int T, int ACT>
__global__ __launch_bounds__(k16Threads, T <= 8 ? 2 : 1) void mlp16_forward_kernel(const float* __restrict__ x, int M, int D,
                                                                                    const float* __restrict__ img,
                                                                                    float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int md = T * 16;
    const Layout16 L = layout16(D, md);
    const Lds16 S = lds16<T>(D, false);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, s = lane & 15, g = lane >> 4;
    stage_small16<T>(lds, S, img, L, tid);
    __syncthreads();
    const int ngroups = (M + k16Group - 1) / k16Group;
#pragma unroll 1
    for (int group = blockIdx.x; group < ngroups; group += gridDim.x) {
        const int m = group * k16Group + wave * 16 + s;
        float xr[8];
        gather_x16(x + (size_t)(m < M ? m : M - 1) * D, D, L.KS0, g, xr);
        f32x4 h1[T], h2[T];
        first16<T, ACT>(lds + S.w0, lds + S.b0, L.KS0, lane, xr, h1);
        PIME_NO_HOIST();
        layer16<T, ACT, true>(img + L.w1, lds + S.b1, lds + S.region, lane, tid, h1, h2);
        PIME_NO_HOIST();
        layer16<T, ACT, true>(img + L.w2, lds + S.b2, lds + S.region, lane, tid, h2, h1);
        const float y = head16<T>(lds + S.w3, lds[S.b3], lane, h1);
        if (g == 0 && m < M) out[m] = y;
    }
}

// ---- PPO minibatch gradients of one net ------------------------------------------------------------------------------
// PIME_FUSED_TRACE=<workgroup>: wall-clock marks (100 MHz) of that workgroup's first two groups -- a tuning aid
#define PIME16_MARK(i)                                                                                          \
    do {                                                                                                        \
        if (a.trace && blockIdx.x == a.trace_wg && threadIdx.x == 0 && mark0 + (i) < 32) a.trace[mark0 + (i)] = wall_clock64(); \
    } while (0)

// B3: the four streamed layers (two forward, two dX) on bf16 matrix instructions, every f32 operand as three bf16 pieces
// (layer16r_b3; PIME_GRAD_BF16X3=1).  The weight gradients, the first layer and the head stay on the f32 path.
template <int T, bool ACTOR, bool B3 = false>
__global__ __launch_bounds__(k16Threads, T <= 8 ? 2 : 1) void ppo16_kernel(PpoArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int ACT = ACTOR ? 1 : 0;
    constexpr int md = T * 16;
    constexpr int RT = T <= 8 ? 4 : 2;   // tiles per weight-gradient round (dw16)
    const Layout16 L = layout16(a.D, md);
    const LayoutB16 Lb = layoutb16(md);
    const LayoutB3_16 L3 = layout_b3_16(md);
    const float* const b3 = a.img_bwd + Lb.total;   // the bf16x3 planes behind the transposed image (B3 only)
    const Lds16 S = lds16<T>(a.D, true);
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* const region = lds + S.region;
    double* const wsum = reinterpret_cast<double*>(lds + S.wsum);   // [wave][6]
    const float asl = ACTOR ? a.a_std_log[0] : 0.f;
    if (a.trace_span && tid == 0 && blockIdx.x < 512) a.trace_span[2 * blockIdx.x] = wall_clock64();
    stage_small16<T>(lds, S, a.img_fwd, L, tid);
    if (tid < k16Waves * 6) wsum[tid] = 0.0;
    __syncthreads();
    const int ngroups = (a.B + k16Group - 1) / k16Group;
    const float invB = 1.0f / (float)a.B;
    float* const sl = a.slab + (size_t)blockIdx.x * a.slab_stride;   // this workgroup's partial gradients

#pragma unroll 1
    for (int group = blockIdx.x; group < ngroups; group += gridDim.x) {
        int lane = tid & 63;
        asm volatile("" : "+v"(lane));   // keep per-lane offsets inside the loop (hipcc hoists and spills them otherwise)
        const int s = lane & 15, g = lane >> 4;
        const bool accum = group != (int)blockIdx.x;
        const int mark0 = accum ? 12 : 0;
        PIME16_MARK(0);
        const int pos = group * k16Group + wave * 16 + s;
        const bool valid = pos < a.B;
        const int64_t* const idx = a.indices + (a.index_row ? (size_t)a.index_row[0] * a.B : 0);
        const long long row = idx[valid ? pos : a.B - 1];
        float xr[8];
        gather_x16(a.state + (size_t)row * a.D, a.D, L.KS0, g, xr);
        const float in_rsum = ACTOR ? 0.f : a.r_sum[row];
        const float in_action = ACTOR ? a.action[row] : 0.f;
        const float in_logprob = ACTOR ? a.logprob[row] : 0.f;
        const float in_adv = ACTOR ? a.adv[row] : 0.f;

        // ------------------------------------------------------------------------------------------ forward
        f32x4 h2[T], h3[T];
        {
            f32x4 h1[T];
            first16<T, ACT>(lds + S.w0, lds + S.b0, L.KS0, lane, xr, h1);
            PIME_NO_HOIST();
            PIME16_MARK(1);
            chain16<T, T, ACT, true, B3>(a.img_fwd + L.w1, b3 + L3.w1, lds + S.b1, region, lane, tid, h1, h2);
        }
        PIME16_MARK(2);
        PIME_NO_HOIST();
        chain16<T, T, ACT, true, B3>(a.img_fwd + L.w2, b3 + L3.w2, lds + S.b2, region, lane, tid, h2, h3);
        PIME16_MARK(3);
        const float y = head16<T>(lds + S.w3, lds[S.b3], lane, h3);

        // ------------------------------------------------------------------------------------------ loss gradient
        float dout = 0.f, s0 = 0.f, s1 = 0.f, gstd = 0.f;
        double m1 = 0.0, m2 = 0.0;
        if (valid) {
            if constexpr (!ACTOR) {
                const float d = y - in_rsum, ad = fabsf(d);         // SmoothL1, beta = 1 (agent.py:567,649)
                const float l = ad < 1.f ? 0.5f * d * d : ad - 0.5f;
                const float gl = ad < 1.f ? d : (d > 0.f ? 1.f : -1.f);
                dout = gl * invB;   // unscaled: the slab reduction applies 1/(std+1e-5) (agent.py:652)
                if (g == 0) { s0 = l; m1 = (double)in_rsum; m2 = (double)in_rsum * (double)in_rsum; }
            } else {
                const float inv_sigma = __expf(-asl);   // asl: loaded at kernel entry
                const float z = (y - in_action) * inv_sigma;
                const float logp = -(asl + kLogSqrt2Pi + 0.5f * z * z);           // compute_logprob
                const float ratio = __expf(logp - in_logprob);
                const float lo = 1.f - a.ratio_clip, hi = 1.f + a.ratio_clip;
                const float clamped = fminf(fmaxf(ratio, lo), hi);
                const float u = in_adv * ratio, c = in_adv * clamped;             // agent.py:639-641
                const float w_u = u < c ? 1.f : (u == c ? 0.5f : 0.f);            // torch.min backward (ties split)
                const float w_c = c < u ? 1.f : (u == c ? 0.5f : 0.f);
                const bool in_range = ratio >= lo && ratio <= hi;
                const float g_sur = w_u * u + (in_range ? w_c * u : 0.f);
                const float p = __expf(logp);
                const float g_logp = (-g_sur + a.lambda_entropy * p * (logp + 1.f)) * invB;
                dout = g_logp * (-z * inv_sigma);
                if (g == 0) { gstd = g_logp * (z * z - 1.f); s0 = -fminf(u, c); s1 = p * logp; }
            }
        }
        {
            float ghb = g == 0 ? dout : 0.f;   // head bias gradient
            s0 = wave_total_dpp(s0); ghb = wave_total_dpp(ghb);   // totals valid in lane 63
            if constexpr (!ACTOR) {
                for (int o = 32; o > 0; o >>= 1) { m1 += __shfl_xor(m1, o); m2 += __shfl_xor(m2, o); }
            } else {
                s1 = wave_total_dpp(s1); gstd = wave_total_dpp(gstd);
            }
            if (lane == 63) {
                double* w = wsum + wave * 6;
                w[0] += s0; w[1] += s1; w[2] += gstd; w[3] += ghb; w[4] += m1; w[5] += m2;
            }
        }
        // head weight gradient: sum over the tile's samples of dOut h3 (DPP row sums), per wave into the (free) region,
        // then over the waves in wave order; dZ3 = (w3 dOut) . act'(h3) replaces h3
        {
#pragma unroll
            for (int t = 0; t < T; ++t) {
                const f32x4 w = *reinterpret_cast<const f32x4*>(lds + S.w3 + t * 16 + 4 * g);
                f32x4 hs;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    hs[r] = row_sum16(dout * h3[t][r]);
                    h3[t][r] = w[r] * dout * act_grad_from_output<ACT>(h3[t][r]);
                }
                if (s == 15) *reinterpret_cast<f32x4*>(region + wave * md + t * 16 + 4 * g) = hs;
            }
            PIME16_BARRIER();
            if (tid < md) {
                float t8 = region[tid];
                for (int w = 1; w < k16Waves; ++w) t8 += region[w * md + tid];
                float* qd = &sl[a.poff[6] + tid];
                *qd = accum ? *qd + t8 : t8;
            }
        }
        // ------------------------------------------------------------------------------------------ backward
        PIME16_MARK(4);
        PIME_NO_HOIST();
        if constexpr (T == 4 * k16Waves && B3)
            dw16_sliced_b3<T>(region, lane, wave, h3, h2, sl + a.poff[4], sl + a.poff[5], accum);                           // net.4
        else if constexpr (T == 4 * k16Waves)
            dw16_sliced<T>(region, lane, wave, h3, h2, sl + a.poff[4], sl + a.poff[5], accum,
                           (a.trace && blockIdx.x == a.trace_wg && !accum) ? a.trace + 40 : nullptr);
        else
            dw16<T, T, RT>(region, lane, wave, PubAcc16<T>{h3, g}, PubAcc16<T>{h2, g}, sl + a.poff[4], sl + a.poff[5], accum);
        f32x4 d2[T];
        PIME16_BARRIER();
        PIME16_MARK(5);
        chain16<T, T, 2, false, B3>(a.img_bwd + Lb.w2t, b3 + L3.w2t, nullptr, region, lane, tid, h3, d2);
        PIME16_MARK(6);
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) d2[t][r] *= act_grad_from_output<ACT>(h2[t][r]);                                                 // dZ2
        f32x4(&h1)[T] = h2;   // h2 is dead: its registers take the recomputed first-layer activation
        PIME_NO_HOIST();
        first16<T, ACT>(lds + S.w0, lds + S.b0, L.KS0, lane, xr, h1);
        if constexpr (T == 4 * k16Waves && B3)
            dw16_sliced_b3<T>(region, lane, wave, d2, h1, sl + a.poff[2], sl + a.poff[3], accum);                           // net.2
        else if constexpr (T == 4 * k16Waves)
            dw16_sliced<T>(region, lane, wave, d2, h1, sl + a.poff[2], sl + a.poff[3], accum);
        else
            dw16<T, T, RT>(region, lane, wave, PubAcc16<T>{d2, g}, PubAcc16<T>{h1, g}, sl + a.poff[2], sl + a.poff[3], accum);
        f32x4(&d1)[T] = h3;   // dZ3 is dead
        PIME16_BARRIER();
        PIME16_MARK(7);
        chain16<T, T, 2, false, B3>(a.img_bwd + Lb.w1t, b3 + L3.w1t, nullptr, region, lane, tid, d2, d1);
        PIME16_MARK(8);
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) d1[t][r] *= act_grad_from_output<ACT>(h1[t][r]);                                                 // dZ1
        PIME_NO_HOIST();
        if (L.Dp <= 16)
            dw16<T, 1, RT>(region, lane, wave, PubAcc16<T>{d1, g}, PubX16{xr, g, L.KS0, 16}, sl + a.poff[0], sl + a.poff[1], accum);
        else
            dw16<T, 2, RT>(region, lane, wave, PubAcc16<T>{d1, g}, PubX16{xr, g, L.KS0, 32}, sl + a.poff[0], sl + a.poff[1], accum);
        PIME16_BARRIER();   // the next group's first chain layer writes the region
        PIME16_MARK(9);
    }

    if (a.trace_span && tid == 0 && blockIdx.x < 512) a.trace_span[2 * blockIdx.x + 1] = wall_clock64();
    // ---- workgroup totals of the scalar sums, in a fixed order (only the logged loss sums use atomics)
    __syncthreads();
    if (tid == 0) {
        double t[6] = {0, 0, 0, 0, 0, 0};
        for (int w = 0; w < k16Waves; ++w)
            for (int k = 0; k < 6; ++k) t[k] += wsum[w * 6 + k];
        sl[a.poff[7]] = (float)t[3];                  // head bias
        if constexpr (!ACTOR) {
            atomicAdd(&a.loss_sums[2], (float)t[0]);
            double* mo = reinterpret_cast<double*>(sl + a.poff[8]);
            mo[0] = t[4]; mo[1] = t[5];
        } else {
            atomicAdd(&a.loss_sums[0], (float)t[0]);
            atomicAdd(&a.loss_sums[1], (float)t[1]);
            sl[a.poff[8]] = (float)t[2];              // d loss / d a_std_log
        }
    }
}

// ==================================================================================================== modular actor
// ActorResidualIntegratorModularPPO (/root/reference/elegantrl/net_residual.py:138-205) in the 16-tile family, for the width the
// LDS-resident kernels cannot hold (net_dim 256: run_watertank_changing.sh:11-18):
//   o1 = tanh(Wo0 x[:, :Do] + bo0)  (md)     o2 = tanh(Wo2 o1 + bo2)  (md / 2)
//   i1 = tanh(Wi0 x[:, Do:] + bi0)  (md)     i2 = tanh(Wi2 i1 + bi2)  (md / 2)
//   n0 = tanh(Wn0 [o2 | i2] + bn0)  (md)     mean = wn2 . n0 + bn2
// The towers' second layers are rectangular chain layers (layer16r<T, T/2>), their transposes layer16r<T/2, T>; `cat` is the two
// half-width activations side by side in one T-tile register array, so net.0 and its weight gradient are the square-layer code.
// Weight gradients of the tower layers: dw16<T/2, T> (rectangular patches), first layers dw16<T, 1> on the tower's own columns.
struct Layout16M {
    int T, KS0o, KS0i;
    int w0o, b0o, w1o, b1o, w0i, b0i, w1i, b1i, wn, bn, w3, b3, total;   // forward image (float offsets)
};
struct LayoutB16M {
    int wnt, w1ot, w1it, total;   // backward image: Wn0^T (md -> md), Wo2^T, Wi2^T (md/2 -> md) as chain-order layers
};
__host__ __device__ inline Layout16M layout16m(int D, int Di, int md) {
    Layout16M L{};
    const int Do = D - Di;
    L.T = md / 16;
    L.KS0o = (Do + 3) / 4;
    L.KS0i = (Di + 3) / 4;
    int o = 0;
    auto seg = [&](int& f, int n) { f = o; o += (n + 3) & ~3; };
    seg(L.w0o, L.KS0o * L.T * 64); seg(L.b0o, md);
    seg(L.w1o, md * md / 2); seg(L.b1o, md / 2);
    seg(L.w0i, L.KS0i * L.T * 64); seg(L.b0i, md);
    seg(L.w1i, md * md / 2); seg(L.b1i, md / 2);
    seg(L.wn, md * md); seg(L.bn, md);
    seg(L.w3, md); seg(L.b3, 4);
    L.total = o;
    return L;
}
__host__ __device__ inline LayoutB16M layoutb16m(int md) {
    LayoutB16M L{};
    L.wnt = 0; L.w1ot = md * md; L.w1it = md * md + md * md / 2; L.total = 2 * md * md;
    return L;
}

// params (nn.Linear W, b pairs): other_net.0, other_net.2, integrator_net.0, integrator_net.2, net.0, net.2
__global__ void pack16m_kernel(PackArgs a, float* __restrict__ fwd, float* __restrict__ bwd) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x, nthr = gridDim.x * blockDim.x;
    const int md = a.md, hd = md / 2, Do = a.D - a.Di;
    if (fwd) {
        const Layout16M L = layout16m(a.D, a.Di, md);
        pack16_layer(fwd + L.w0o, a.p[0], md, Do, md, Do, L.KS0o, true, false, tid, nthr);
        pack16_layer(fwd + L.w1o, a.p[2], hd, md, hd, md, md / 4, false, false, tid, nthr);
        pack16_layer(fwd + L.w0i, a.p[4], md, a.Di, md, a.Di, L.KS0i, true, false, tid, nthr);
        pack16_layer(fwd + L.w1i, a.p[6], hd, md, hd, md, md / 4, false, false, tid, nthr);
        pack16_layer(fwd + L.wn, a.p[8], md, md, md, md, md / 4, false, false, tid, nthr);
        for (int i = tid; i < md; i += nthr) {
            fwd[L.b0o + i] = a.p[1][i];
            fwd[L.b0i + i] = a.p[5][i];
            fwd[L.bn + i] = a.p[9][i];
            fwd[L.w3 + i] = a.p[10][i];
            if (i < hd) { fwd[L.b1o + i] = a.p[3][i]; fwd[L.b1i + i] = a.p[7][i]; }
        }
        if (tid < 4) fwd[L.b3 + tid] = tid == 0 ? a.p[11][0] : 0.f;
    }
    if (bwd) {
        const LayoutB16M L = layoutb16m(md);
        pack16_layer(bwd + L.wnt, a.p[8], md, md, md, md, md / 4, false, true, tid, nthr);
        // V = W^T of a [hd x md] matrix: md outputs, hd inputs (k-steps hd / 4); W's row length is md
        pack16_layer(bwd + L.w1ot, a.p[2], hd, md, md, hd, hd / 4, false, true, tid, nthr);
        pack16_layer(bwd + L.w1it, a.p[6], hd, md, md, hd, hd / 4, false, true, tid, nthr);
    }
}

struct Lds16M {
    int w0o, b0o, b1o, w0i, b0i, b1i, bn, w3, b3, wsum, region, total;
};
template <int T>
__host__ __device__ inline Lds16M lds16m(int D, int Di, bool grad) {
    const int md = T * 16;
    const Layout16M L = layout16m(D, Di, md);
    Lds16M S{};
    int o = 0;
    auto seg = [&](int& f, int n) { f = o; o += (n + 3) & ~3; };
    seg(S.w0o, L.KS0o * T * 64); seg(S.b0o, md); seg(S.b1o, md / 2);
    seg(S.w0i, L.KS0i * T * 64); seg(S.b0i, md); seg(S.b1i, md / 2);
    seg(S.bn, md); seg(S.w3, md); seg(S.b3, 4);
    seg(S.wsum, k16Waves * 6 * 2);
    auto mx = [](int a, int b) { return a > b ? a : b; };
    int region = mx(layer16_lds_floats<T, T>(), mx(layer16_lds_floats<T, T / 2>(), layer16_lds_floats<T / 2, T>()));
    if (grad) {
        constexpr int RT = T <= 8 ? 4 : 2;
        int dwf = mx(dw16_lds_floats<T, T, RT>(), mx(dw16_lds_floats<T / 2, T, RT>(), dw16_lds_floats<T, 1, RT>()));
        if constexpr (T == 4 * k16Waves) dwf = mx(dwf, mx(Dw16Sliced<T>::FLOATS, Dw16SlicedB3<T>::FLOATS));
        region = mx(region, mx(dwf, k16Waves * md));
        if constexpr (T >= 8)
            region = mx(region, mx(layer16_b3_lds_floats<T, T>(), mx(layer16_b3_lds_floats<T, T / 2>(), layer16_b3_lds_floats<T / 2, T>())));
    }
    seg(S.region, region);
    S.total = o;
    return S;
}

template <int T>
__device__ __forceinline__ void stage_small16m(float* lds, const Lds16M& S, const float* __restrict__ img, const Layout16M& L, int tid) {
    const int md = T * 16;
    for (int e = tid; e < L.KS0o * T * 16; e += k16Threads) reinterpret_cast<float4*>(lds + S.w0o)[e] = reinterpret_cast<const float4*>(img + L.w0o)[e];
    for (int e = tid; e < L.KS0i * T * 16; e += k16Threads) reinterpret_cast<float4*>(lds + S.w0i)[e] = reinterpret_cast<const float4*>(img + L.w0i)[e];
    for (int e = tid; e < md; e += k16Threads) {
        lds[S.b0o + e] = img[L.b0o + e];
        lds[S.b0i + e] = img[L.b0i + e];
        lds[S.bn + e] = img[L.bn + e];
        lds[S.w3 + e] = img[L.w3 + e];
        if (e < md / 2) { lds[S.b1o + e] = img[L.b1o + e]; lds[S.b1i + e] = img[L.b1i + e]; }
    }
    if (tid < 4) lds[S.b3 + tid] = img[L.b3 + tid];
}

// the towers' inputs in the first layer's natural k order: lane (s, g) holds column 4 ks + g of its tower's slice of the state
__device__ __forceinline__ void gather_x16m(const float* __restrict__ xrow, int D, int Di, int g, float (&xo)[8], float (&xi)[8]) {
    const int Do = D - Di;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
        const int c = 4 * ks + g;
        xo[ks] = c < Do ? xrow[c] : 0.f;
        xi[ks] = c < Di ? xrow[Do + c] : 0.f;
    }
}

// policy mean of the modular actor (forward only): pime_mlp_forward at width 256
template <int T>
__global__ __launch_bounds__(k16Threads, T <= 8 ? 2 : 1) void mlp16m_forward_kernel(const float* __restrict__ x, int M, int D, int Di,
                                                                                     const float* __restrict__ img,
                                                                                     float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int H = T / 2;
    const Layout16M L = layout16m(D, Di, T * 16);
    const Lds16M S = lds16m<T>(D, Di, false);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, s = lane & 15, g = lane >> 4;
    stage_small16m<T>(lds, S, img, L, tid);
    __syncthreads();
    const int ngroups = (M + k16Group - 1) / k16Group;
#pragma unroll 1
    for (int group = blockIdx.x; group < ngroups; group += gridDim.x) {
        const int m = group * k16Group + wave * 16 + s;
        float xo[8], xi[8];
        gather_x16m(x + (size_t)(m < M ? m : M - 1) * D, D, Di, g, xo, xi);
        f32x4 cat[T], n0[T];
        {
            f32x4 t1[T];
            first16<T, 1>(lds + S.w0o, lds + S.b0o, L.KS0o, lane, xo, t1);
            PIME_NO_HOIST();
            layer16r<T, H, 1, true>(img + L.w1o, lds + S.b1o, lds + S.region, lane, tid, t1, *reinterpret_cast<f32x4(*)[H]>(&cat[0]));
            PIME_NO_HOIST();
            first16<T, 1>(lds + S.w0i, lds + S.b0i, L.KS0i, lane, xi, t1);
            PIME_NO_HOIST();
            layer16r<T, H, 1, true>(img + L.w1i, lds + S.b1i, lds + S.region, lane, tid, t1, *reinterpret_cast<f32x4(*)[H]>(&cat[H]));
        }
        PIME_NO_HOIST();
        layer16<T, 1, true>(img + L.wn, lds + S.bn, lds + S.region, lane, tid, cat, n0);
        const float y = head16<T>(lds + S.w3, lds[S.b3], lane, n0);
        if (g == 0 && m < M) out[m] = y;
    }
}

// PPO minibatch gradients of the modular actor: ppo16_kernel's structure (no activation stash, slabs in accumulator order)
template <int T, bool B3 = false>
__global__ __launch_bounds__(k16Threads, T <= 8 ? 2 : 1) void ppo16m_kernel(PpoArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int ACT = 1, md = T * 16, H = T / 2;
    constexpr int RT = T <= 8 ? 4 : 2;
    const Layout16M L = layout16m(a.D, a.Di, md);
    const LayoutB16M Lb = layoutb16m(md);
    const LayoutB3_16M L3 = layout_b3_16m(md);
    const float* const b3 = a.img_bwd + Lb.total;   // the bf16x3 planes behind the transposed image (B3 only)
    const Lds16M S = lds16m<T>(a.D, a.Di, true);
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* const region = lds + S.region;
    double* const wsum = reinterpret_cast<double*>(lds + S.wsum);   // [wave][6]
    const float asl = a.a_std_log[0];
    stage_small16m<T>(lds, S, a.img_fwd, L, tid);
    if (tid < k16Waves * 6) wsum[tid] = 0.0;
    __syncthreads();
    const int ngroups = (a.B + k16Group - 1) / k16Group;
    const float invB = 1.0f / (float)a.B;
    float* const sl = a.slab + (size_t)blockIdx.x * a.slab_stride;   // this workgroup's partial gradients
    // slab positions (slab_layout16m): 0/1 other.0 W,b  2/3 other.2  4/5 integrator.0  6/7 integrator.2  8/9 net.0  10/11 net.2  12 scalars

#pragma unroll 1
    for (int group = blockIdx.x; group < ngroups; group += gridDim.x) {
        int lane = tid & 63;
        asm volatile("" : "+v"(lane));   // keep per-lane offsets inside the loop
        const int s = lane & 15, g = lane >> 4;
        const bool accum = group != (int)blockIdx.x;
        const int pos = group * k16Group + wave * 16 + s;
        const bool valid = pos < a.B;
        const int64_t* const idx = a.indices + (a.index_row ? (size_t)a.index_row[0] * a.B : 0);
        const long long row = idx[valid ? pos : a.B - 1];
        float xo[8], xi[8];
        gather_x16m(a.state + (size_t)row * a.D, a.D, a.Di, g, xo, xi);
        const float in_action = a.action[row], in_logprob = a.logprob[row], in_adv = a.adv[row];

        // ------------------------------------------------------------------------------------------ forward
        f32x4 cat[T], n0[T];
        {
            f32x4 t1[T];
            first16<T, ACT>(lds + S.w0o, lds + S.b0o, L.KS0o, lane, xo, t1);
            PIME_NO_HOIST();
            chain16<T, H, ACT, true, B3>(a.img_fwd + L.w1o, b3 + L3.w1o, lds + S.b1o, region, lane, tid, t1, *reinterpret_cast<f32x4(*)[H]>(&cat[0]));
            PIME_NO_HOIST();
            first16<T, ACT>(lds + S.w0i, lds + S.b0i, L.KS0i, lane, xi, t1);
            PIME_NO_HOIST();
            chain16<T, H, ACT, true, B3>(a.img_fwd + L.w1i, b3 + L3.w1i, lds + S.b1i, region, lane, tid, t1, *reinterpret_cast<f32x4(*)[H]>(&cat[H]));
        }
        PIME_NO_HOIST();
        chain16<T, T, ACT, true, B3>(a.img_fwd + L.wn, b3 + L3.wn, lds + S.bn, region, lane, tid, cat, n0);
        const float y = head16<T>(lds + S.w3, lds[S.b3], lane, n0);

        // ------------------------------------------------------------------------------------------ loss gradient (agent.py:637-645)
        float dout = 0.f, s0 = 0.f, s1 = 0.f, gstd = 0.f;
        if (valid) {
            const float inv_sigma = __expf(-asl);
            const float z = (y - in_action) * inv_sigma;
            const float logp = -(asl + kLogSqrt2Pi + 0.5f * z * z);           // compute_logprob
            const float ratio = __expf(logp - in_logprob);
            const float lo = 1.f - a.ratio_clip, hi = 1.f + a.ratio_clip;
            const float clamped = fminf(fmaxf(ratio, lo), hi);
            const float u = in_adv * ratio, c = in_adv * clamped;
            const float w_u = u < c ? 1.f : (u == c ? 0.5f : 0.f);            // torch.min backward (ties split)
            const float w_c = c < u ? 1.f : (u == c ? 0.5f : 0.f);
            const bool in_range = ratio >= lo && ratio <= hi;
            const float g_sur = w_u * u + (in_range ? w_c * u : 0.f);
            const float p = __expf(logp);
            const float g_logp = (-g_sur + a.lambda_entropy * p * (logp + 1.f)) * invB;
            dout = g_logp * (-z * inv_sigma);
            if (g == 0) { gstd = g_logp * (z * z - 1.f); s0 = -fminf(u, c); s1 = p * logp; }
        }
        {
            float ghb = g == 0 ? dout : 0.f;   // head bias gradient
            s0 = wave_total_dpp(s0); ghb = wave_total_dpp(ghb);   // totals valid in lane 63
            s1 = wave_total_dpp(s1); gstd = wave_total_dpp(gstd);
            if (lane == 63) {
                double* w = wsum + wave * 6;
                w[0] += s0; w[1] += s1; w[2] += gstd; w[3] += ghb;
            }
        }
        // head weight gradient (DPP row sums, per wave into the free region, then over the waves); dZn0 replaces n0
        {
#pragma unroll
            for (int t = 0; t < T; ++t) {
                const f32x4 w = *reinterpret_cast<const f32x4*>(lds + S.w3 + t * 16 + 4 * g);
                f32x4 hs;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    hs[r] = row_sum16(dout * n0[t][r]);
                    n0[t][r] = w[r] * dout * act_grad_from_output<ACT>(n0[t][r]);
                }
                if (s == 15) *reinterpret_cast<f32x4*>(region + wave * md + t * 16 + 4 * g) = hs;
            }
            PIME16_BARRIER();
            if (tid < md) {
                float t8 = region[tid];
                for (int w = 1; w < k16Waves; ++w) t8 += region[w * md + tid];
                float* qd = &sl[a.poff[10] + tid];
                *qd = accum ? *qd + t8 : t8;
            }
        }
        // ------------------------------------------------------------------------------------------ backward
        PIME_NO_HOIST();
        if constexpr (T == 4 * k16Waves && B3)
            dw16_sliced_b3<T>(region, lane, wave, n0, cat, sl + a.poff[8], sl + a.poff[9], accum);                                   // net.0
        else if constexpr (T == 4 * k16Waves)
            dw16_sliced<T>(region, lane, wave, n0, cat, sl + a.poff[8], sl + a.poff[9], accum);
        else
            dw16<T, T, RT>(region, lane, wave, PubAcc16<T>{n0, g}, PubAcc16<T>{cat, g}, sl + a.poff[8], sl + a.poff[9], accum);
        f32x4 dcat[T];
        PIME16_BARRIER();
        chain16<T, T, 2, false, B3>(a.img_bwd + Lb.wnt, b3 + L3.wnt, nullptr, region, lane, tid, n0, dcat);
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) dcat[t][r] *= act_grad_from_output<ACT>(cat[t][r]);                                          // [dZo2 | dZi2]
        // the two towers, one after the other: cat / n0 are dead, their registers take the recomputed first-layer activation and dZ1
#pragma unroll
        for (int br = 0; br < 2; ++br) {
            f32x4(&t1)[T] = cat;     // tower's first-layer activation (recomputed)
            f32x4(&d1)[T] = n0;      // dZ of the tower's first layer
            const f32x4(&dz2)[H] = *reinterpret_cast<const f32x4(*)[H]>(&dcat[br * H]);
            PIME_NO_HOIST();
            if (br == 0) first16<T, ACT>(lds + S.w0o, lds + S.b0o, L.KS0o, lane, xo, t1);
            else first16<T, ACT>(lds + S.w0i, lds + S.b0i, L.KS0i, lane, xi, t1);
            dw16<H, T, RT>(region, lane, wave, PubAcc16<H>{dz2, g}, PubAcc16<T>{t1, g}, sl + a.poff[br ? 6 : 2], sl + a.poff[br ? 7 : 3], accum);
            PIME16_BARRIER();
            chain16<H, T, 2, false, B3>(a.img_bwd + (br ? Lb.w1it : Lb.w1ot), b3 + (br ? L3.w1it : L3.w1ot), nullptr, region, lane, tid, dz2, d1);
#pragma unroll
            for (int t = 0; t < T; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) d1[t][r] *= act_grad_from_output<ACT>(t1[t][r]);                                         // dZ1
            PIME_NO_HOIST();
            if (br == 0) dw16<T, 1, RT>(region, lane, wave, PubAcc16<T>{d1, g}, PubX16{xo, g, L.KS0o, 16}, sl + a.poff[0], sl + a.poff[1], accum);
            else dw16<T, 1, RT>(region, lane, wave, PubAcc16<T>{d1, g}, PubX16{xi, g, L.KS0i, 16}, sl + a.poff[4], sl + a.poff[5], accum);
            PIME16_BARRIER();   // the next chain layer / the next group writes the region
        }
    }

    // ---- workgroup totals of the scalar sums, in a fixed order (only the logged loss sums use atomics)
    __syncthreads();
    if (tid == 0) {
        double t[6] = {0, 0, 0, 0, 0, 0};
        for (int w = 0; w < k16Waves; ++w)
            for (int k = 0; k < 6; ++k) t[k] += wsum[w * 6 + k];
        sl[a.poff[11]] = (float)t[3];                 // head bias
        atomicAdd(&a.loss_sums[0], (float)t[0]);
        atomicAdd(&a.loss_sums[1], (float)t[1]);
        sl[a.poff[12]] = (float)t[2];                 // d loss / d a_std_log
    }
}

// ==================================================================================================== fused rollout, width 256
// The one-launch-per-episode rollout (csrc/rollout.hip: policy forward + exploration noise + residual composition + env step +
// in-kernel auto-reset + trajectory writes, replacing agent_residual.py:52-69) for the width the LDS-resident kernel cannot hold:
// a workgroup = four waves = 64 env lanes (one 16-lane tile per wave, every lane's env arithmetic replicated over the four
// feature groups g of the 16x16x4 layout), the weight images streamed through LDS once per env step by the chain layers above.
// 4 096 lanes are 64 workgroups: a quarter of the chip, each a serial chain of ~2 300 MFMAs per step -- latency-bound like the
// 32x32 rollout, but one launch per episode instead of 200 x [mlp16 forward + env step] launches and their gaps (round 2:
// 18 ms of the 134 ms width-256 step).
constexpr uint32_t STREAM_EXPLORE16 = 2;   // = rollout.hip's STREAM_EXPLORE

template <int D>
__device__ __forceinline__ float pick_col(const float (&obs)[D], int c0, int g) {   // obs[c0 + g], 0 beyond the row (c0 compile-time)
    const float v0 = c0 + 0 < D ? obs[c0 + 0 < D ? c0 + 0 : 0] : 0.f, v1 = c0 + 1 < D ? obs[c0 + 1 < D ? c0 + 1 : 0] : 0.f;
    const float v2 = c0 + 2 < D ? obs[c0 + 2 < D ? c0 + 2 : 0] : 0.f, v3 = c0 + 3 < D ? obs[c0 + 3 < D ? c0 + 3 : 0] : 0.f;
    return g == 0 ? v0 : (g == 1 ? v1 : (g == 2 ? v2 : v3));
}

// QUAD (launches of <= 4 096 lanes): the four waves share ONE 16-lane tile, each a quad of every layer's output tiles (layer16q; two
// tiles of the modular actor's tower layers: layer16rq) -- 256 workgroups instead of 64 for the reference script's 4 096 lanes, a quarter of the MFMA chain per env step.
template <int T, int KIND, int ENV, int STACK, bool QUAD = false>
__global__ __launch_bounds__(k16Threads, 1) void rollout16_kernel(RolloutArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int D = ENV == 0 ? 3 : (ENV == 1 ? 4 : 3 * STACK), Di = ENV == 2 ? 0 : 1, Do = D - Di, H = T / 2;
    constexpr bool MODULAR = KIND == MLP_MODULAR_ACTOR;
    static_assert(!MODULAR || ENV != 2, "the Stacking observation has no integrator column: plain actors only");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, sl = lane & 15, g = lane >> 4;
    const Layout16 L = layout16(D, T * 16);
    const Lds16 S = lds16<T>(D, false);
    const Layout16M Lm = layout16m(D, Di > 0 ? Di : 1, T * 16);
    const Lds16M Sm = lds16m<T>(D, Di > 0 ? Di : 1, false);
    if constexpr (MODULAR) stage_small16m<T>(lds, Sm, a.img, Lm, tid);
    else stage_small16<T>(lds, S, a.img, L, tid);
    __syncthreads();
    float* const region = lds + (MODULAR ? Sm.region : S.region);
    const int N = a.n;
    const int m = QUAD ? blockIdx.x * 16 + sl : blockIdx.x * k16Group + wave * 16 + sl;
    const bool valid = m < N;
    const int i = valid ? m : N - 1;   // idle lanes shadow the last env (compute, never store)
    const bool writer = valid && g == 0 && (!QUAD || wave == 0);
    [[maybe_unused]] float* const xb = lds + (MODULAR ? Sm.total : S.total);   // QUAD: the activation exchange buffer behind the map
    const uint32_t gid = a.env_offset + (uint32_t)i;
    const bool evaluating = a.eval_mode != 0;   // wave-uniform
    const float sigma = evaluating ? 0.f : __expf(a.a_std_log[0]);

    PhLane<float> E{};
    WtLane<float> W{};
    if constexpr (ENV == 0) ph_lane_load<float>(a.p, a.st, i, E);
    else wt_lane_load<float>(a.wp, a.wst, i, W);
    float obs[D];
    if (evaluating) {   // the lanes' current observation from the state (what pime_env_observe writes)
        if constexpr (ENV == 0) {
            obs[0] = (float)ph_lookup<float>(a.p, a.st.table, E.C, E.x); obs[1] = E.r; obs[2] = E.I;
        } else if constexpr (ENV == 1) {
            obs[0] = W.h1; obs[1] = W.h2; obs[2] = W.r; obs[3] = W.I;
        } else {
            const int head = a.wst.head[i];   // frame ring, oldest first
#pragma unroll
            for (int f = 0; f < STACK; ++f) {
                const int slot = (head + f) % STACK;
#pragma unroll
                for (int c = 0; c < 3; ++c) obs[3 * f + c] = a.wst.frames[(size_t)(3 * slot + c) * a.n + i];
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < D; ++j) obs[j] = a.state[(size_t)D * i + j];
    }
    double ret = 0.0;
    for (int t = 0; t < a.n_steps; ++t) {
        PIME_NO_HOIST();
        if (evaluating && a.seg_len > 0 && t % a.seg_len == 0) {   // segment boundary of a step-response protocol (wave-uniform), as
            const float sp = (float)a.setpoint[t / a.seg_len];     // rollout_eval_kernel: new set-point, I = 0, clock 0, next noise episode
            const int bump = t > 0 ? 1 : 0;
            if constexpr (ENV == 0) { E.r = sp; E.I = 0.f; E.t = 0; E.episode += bump; obs[1] = sp; obs[2] = 0.f; }
            else {
                W.r = sp; W.I = 0.f; W.t = 0; W.episode += bump;
                if constexpr (ENV == 1) { obs[2] = sp; obs[3] = 0.f; }
                else {   // Stacking: what a reset leaves -- every frame the current one
#pragma unroll
                    for (int f = 0; f < STACK; ++f) { obs[3 * f] = W.h1; obs[3 * f + 1] = W.h2; obs[3 * f + 2] = sp; }
                }
            }
        }
        float a_avg;
        if constexpr (MODULAR) {
            float xo[8], xi[8];
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const float vo = pick_col<D>(obs, 4 * ks, g);
                xo[ks] = 4 * ks + g < Do ? vo : 0.f;
                // the integrator tower's columns start at Do: column Do + 4 ks + g
                float vi = 0.f;
#pragma unroll
                for (int c = 0; c < Di; ++c)
                    if (c / 4 == ks) vi = (c & 3) == g ? obs[Do + c] : vi;
                xi[ks] = vi;
            }
            f32x4 cat[T], n0[T];
            {
                f32x4 t1[T];
                first16<T, 1>(lds + Sm.w0o, lds + Sm.b0o, Lm.KS0o, lane, xo, t1);
                PIME_NO_HOIST();
                if constexpr (QUAD) layer16rq<T, H, 1>(a.img + Lm.w1o, lds + Sm.b1o, region, xb, 0, lane, tid, t1);
                else layer16r<T, H, 1, true>(a.img + Lm.w1o, lds + Sm.b1o, region, lane, tid, t1, *reinterpret_cast<f32x4(*)[H]>(&cat[0]));
                PIME_NO_HOIST();
                first16<T, 1>(lds + Sm.w0i, lds + Sm.b0i, Lm.KS0i, lane, xi, t1);
                PIME_NO_HOIST();
                if constexpr (QUAD) layer16rq<T, H, 1>(a.img + Lm.w1i, lds + Sm.b1i, region, xb, H, lane, tid, t1);
                else layer16r<T, H, 1, true>(a.img + Lm.w1i, lds + Sm.b1i, region, lane, tid, t1, *reinterpret_cast<f32x4(*)[H]>(&cat[H]));
            }
            if constexpr (QUAD) {   // both towers' halves are in the exchange buffer: publish, read the concatenation back
                PIME16_BARRIER();
#pragma unroll
                for (int t = 0; t < T; ++t) cat[t] = *reinterpret_cast<const f32x4*>(xb + (t * 64 + lane) * 4);
            }
            PIME_NO_HOIST();
            if constexpr (QUAD) layer16q<T, 1>(a.img + Lm.wn, lds + Sm.bn, region, xb, lane, tid, cat, n0);
            else layer16<T, 1, true>(a.img + Lm.wn, lds + Sm.bn, region, lane, tid, cat, n0);
            a_avg = head16<T>(lds + Sm.w3, lds[Sm.b3], lane, n0);
        } else {
            float xr[8];
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) xr[ks] = pick_col<D>(obs, 4 * ks, g);
            f32x4 h1[T], h2[T];
            first16<T, 1>(lds + S.w0, lds + S.b0, L.KS0, lane, xr, h1);
            PIME_NO_HOIST();
            if constexpr (QUAD) {
                layer16q<T, 1>(a.img + L.w1, lds + S.b1, region, xb, lane, tid, h1, h2);
                PIME_NO_HOIST();
                layer16q<T, 1>(a.img + L.w2, lds + S.b2, region, xb, lane, tid, h2, h1);
            } else {
                layer16<T, 1, true>(a.img + L.w1, lds + S.b1, region, lane, tid, h1, h2);
                PIME_NO_HOIST();
                layer16<T, 1, true>(a.img + L.w2, lds + S.b2, region, lane, tid, h2, h1);
            }
            a_avg = head16<T>(lds + S.w3, lds[S.b3], lane, h1);
        }
        // from here on: rollout.hip's step, one env lane per (wave, lane & 15)
        float eps = 0.f;
        if (!evaluating) {
            double ua, ub;
            philox_pair(a.noise_seed, gid, a.noise_epoch, (uint32_t)t, STREAM_EXPLORE16, ua, ub);
            eps = (float)(sqrt(-2.0 * log(1.0 - ua)) * cos(6.283185307179586476925286766559 * ub));
        }
        const float a_pre = a_avg + eps * sigma;                                   // net_residual.py:179 (evaluation: the mean, run.py:608)
        double dot = 0.0;                                                          // agent_residual.py:61
#pragma unroll
        for (int j = 0; j < D; ++j) dot += (double)obs[j] * a.K.k[j];
        const double a_env = residual_tanh(a_pre) + dot;
        float nxt[D], rew;
        bool d;
        double tr0 = 0, tr1 = 0, tr2 = 0;
        if constexpr (ENV == 0) {
            if (a.trace) { tr0 = (double)ph_lookup<float>(a.p, a.st.table, E.C, E.x); tr1 = (double)E.r; tr2 = (double)E.I; }
            float o3[3];
            d = ph_lane_step<float>(a.p, a.st.table, a_env, E, o3, rew);
            if (d && !evaluating) ph_lane_reset<float>(a.p, a.st.table, gid, nullptr, E, o3);     // in-kernel auto-reset
            nxt[0] = o3[0]; nxt[1] = o3[1]; nxt[2] = o3[2];
        } else {
            double z1n, z2n;
            wt_lane_noise<float>(a.wp, gid, W, nullptr, z1n, z2n);
            d = wt_lane_step<float>(a.wp, a_env, z1n, z2n, W, rew);
            const bool rst = d && !evaluating;
            if (rst) wt_lane_reset<float>(a.wp, gid, nullptr, W);
            if constexpr (ENV == 1) {
                nxt[0] = W.h1; nxt[1] = W.h2; nxt[2] = W.r; nxt[D - 1] = W.I;
            } else {   // deque(maxlen=S).append (:1143-1144), or after a reset every frame = the first one (:1181-1183)
#pragma unroll
                for (int j = 0; j < D - 3; ++j) nxt[j] = rst ? (j % 3 == 0 ? W.h1 : (j % 3 == 1 ? W.h2 : W.r)) : obs[j + 3];
                nxt[D - 3] = W.h1; nxt[D - 2] = W.h2; nxt[D - 1] = W.r;
            }
        }
        ret += (double)rew;
        if (evaluating && a.trace && writer) {   // the evaluation kernel's trace layout (rollout_eval.hip)
            double* q = a.trace + (size_t)t * 6 * N + i;
            if constexpr (ENV == 0) {
                q[0] = tr0; q[(size_t)N] = tr1; q[2 * (size_t)N] = tr2; q[3 * (size_t)N] = a_env; q[4 * (size_t)N] = (double)rew;
                q[5 * (size_t)N] = E.x;
            } else {
                q[0] = (double)W.h1; q[(size_t)N] = (double)W.h2; q[2 * (size_t)N] = (double)W.r; q[3 * (size_t)N] = (double)W.I;
                q[4 * (size_t)N] = (double)rew; q[5 * (size_t)N] = a_env;
            }
        }
        const size_t k = (size_t)t * N + i;
        if (writer && !evaluating) {
            a.action[k] = a_pre;
            a.noise[k] = eps;
            a.done[k] = (uint8_t)d;
            a.reward[k] = rew;
            float* sp = a.state + ((size_t)(t + 1) * N + i) * D;
#pragma unroll
            for (int j = 0; j < D; ++j) sp[j] = nxt[j];
        }
#pragma unroll
        for (int j = 0; j < D; ++j) obs[j] = nxt[j];
    }
    if (writer) {
        if constexpr (ENV == 0) ph_lane_store<float>(a.p, a.st, i, E);
        else wt_lane_store<float>(a.wp, a.wst, i, W);
        if constexpr (ENV == 2) {   // the frame ring of the step-per-launch kernels: slot j = frame j, oldest at slot 0
#pragma unroll
            for (int j = 0; j < D; ++j) a.wst.frames[(size_t)j * N + i] = obs[j];
            a.wst.head[i] = 0;
        }
        if (evaluating && a.ret) a.ret[i] += ret;
    }
}

template <int KIND, int ENV, int STACK, bool QUAD>
static int launch_rollout16_q(const RolloutArgs& a, hipStream_t s) {
    constexpr int T = 16, D = ENV == 0 ? 3 : (ENV == 1 ? 4 : 3 * STACK);
    const size_t lds_bytes = sizeof(float) * ((size_t)(KIND == MLP_MODULAR_ACTOR ? lds16m<T>(D, 1, false).total : lds16<T>(D, false).total) +
                                              (QUAD ? T * 64 * 4 : 0));
    static LdsLimit lds_limit;  // per instantiation
    PIME_RAISE_LDS(lds_limit, (rollout16_kernel<T, KIND, ENV, STACK, QUAD>), 160 * 1024);
    const int per_wg = QUAD ? 16 : k16Group;
    hipLaunchKernelGGL((rollout16_kernel<T, KIND, ENV, STACK, QUAD>), dim3((a.n + per_wg - 1) / per_wg), dim3(k16Threads), lds_bytes, s, a);
    PIME_HIP_TRY(hipGetLastError());
    return PIME_OK;
}
template <int KIND, int ENV, int STACK>
static int launch_rollout16_t(const RolloutArgs& a, hipStream_t s) {
    bool quad = a.n <= 4096;   // at most one tile per compute unit: split it over the workgroup's waves (as csrc/rollout.hip: tiling)
    if (const char* e = std::getenv("PIME_ROLLOUT_NARROW")) quad = std::atoi(e) == 2;
    if (quad) return launch_rollout16_q<KIND, ENV, STACK, true>(a, s);
    return launch_rollout16_q<KIND, ENV, STACK, false>(a, s);
}

// width 256 (csrc/rollout.hip dispatches here); binary16 rows (PIME_STATE_MIXED16) are not served at this width
int launch_rollout16(int kind, const RolloutArgs& a, hipStream_t s) {
    PIME_REQUIRE(a.I16 == nullptr, "the width-256 fused rollout serves PIME_STATE_MIXED handles");
    if (kind == MLP_MODULAR_ACTOR) {
        if (a.env == 0) return launch_rollout16_t<MLP_MODULAR_ACTOR, 0, 0>(a, s);
        if (a.env == 1) return launch_rollout16_t<MLP_MODULAR_ACTOR, 1, 0>(a, s);
    } else if (kind == MLP_PLAIN_ACTOR) {
        if (a.env == 0) return launch_rollout16_t<MLP_PLAIN_ACTOR, 0, 0>(a, s);
        if (a.env == 1) return launch_rollout16_t<MLP_PLAIN_ACTOR, 1, 0>(a, s);
        if (a.env == 2 && a.wp.num_stack == 1) return launch_rollout16_t<MLP_PLAIN_ACTOR, 2, 1>(a, s);
        if (a.env == 2 && a.wp.num_stack == 4) return launch_rollout16_t<MLP_PLAIN_ACTOR, 2, 4>(a, s);
        if (a.env == 2 && a.wp.num_stack == 10) return launch_rollout16_t<MLP_PLAIN_ACTOR, 2, 10>(a, s);
    }
    set_error("no width-256 fused rollout for env %d kind %d", a.env, kind);
    return PIME_ERR_ARG;
}

// ==================================================================================================== host side
bool fused_fits(int kind, int D, int Di, int md);

// Which nets this family serves.  Forward images (pime_mlp_pack / pime_mlp_forward; the fused rollout reads the 32x32x2
// layout): width 256 only.  Gradient path (pime_ppo_*): width 256, and widths 64 / 128 when the observation is too wide for the
// LDS-resident kernel's map (the stacked water tank, 30 floats) -- instead of the split net + dW pipeline and its float atomics,
// so that those gradients are reproducible too.  PIME_MLP16=1 (A/B knob) routes every plain net of width 64 / 128 here.
static bool forced16() {
    static const bool forced = std::getenv("PIME_MLP16") != nullptr;
    return forced;
}
bool family16(int kind, int md) {
    if (kind == MLP_MODULAR_ACTOR) return md == 256;   // the modular actor here only at the width the LDS-resident kernels cannot hold
    if (md == 256) return true;
    return forced16() && (md == 64 || md == 128);
}
bool family16_grad(int kind, int md, int D, int Di) {
    if (family16(kind, md)) return true;
    if (kind == MLP_MODULAR_ACTOR) return md == 128 && forced16();   // (A/B and the bf16x3 variant at the headline width)
    if (md != 64 && md != 128) return false;
    return !fused_fits(kind, D, Di, md);
}

int64_t packed16_floats(int kind, int D, int Di, int md) {
    return kind == MLP_MODULAR_ACTOR ? layout16m(D, Di, md).total : layout16(D, md).total;
}
// PIME_GRAD_BF16X3=1 (opt-in): the gradient kernels of this family run their streamed layers on bf16 matrix instructions with every
// f32 operand split into three bf16 pieces (layer16r_b3) -- a different rounding of the same f32 products, not bit-equal to the f32
// MFMA chain.  Widths 128 and 256.
static bool b3_enabled() {
    static const bool on = std::getenv("PIME_GRAD_BF16X3") != nullptr && std::atoi(std::getenv("PIME_GRAD_BF16X3")) != 0;
    return on;
}
bool b3_grad(int kind, int md, int D, int Di) { return b3_enabled() && (md == 128 || md == 256) && family16_grad(kind, md, D, Di); }
int64_t b3_floats(int kind, int md) { return kind == MLP_MODULAR_ACTOR ? layout_b3_16m(md).total : layout_b3_16(md).total; }

// the f32 part of the transposed image; with b3_grad the planes follow it (the caller sizes the image with ppo_bwd_image_floats)
int64_t bwd16_floats(int kind, int md) { return kind == MLP_MODULAR_ACTOR ? layoutb16m(md).total : layoutb16(md).total; }

int launch_pack16(const PackArgs& a, float* fwd, float* bwd, hipStream_t s) {
    if (a.kind == MLP_MODULAR_ACTOR) hipLaunchKernelGGL(pack16m_kernel, dim3(128), dim3(256), 0, s, a, fwd, bwd);
    else hipLaunchKernelGGL(pack16_kernel, dim3(128), dim3(256), 0, s, a, fwd, bwd);
    if (bwd && b3_grad(a.kind, a.md, a.D, a.Di))
        hipLaunchKernelGGL(pack16_b3_kernel, dim3(256), dim3(256), 0, s, a, bwd + bwd16_floats(a.kind, a.md));
    PIME_HIP_TRY(hipGetLastError());
    return PIME_OK;
}
// the planes alone (after an optimizer step that kept the f32 images current through the image map)
int launch_pack16_b3(const PackArgs& a, float* bwd, hipStream_t s) {
    hipLaunchKernelGGL(pack16_b3_kernel, dim3(256), dim3(256), 0, s, a, bwd + bwd16_floats(a.kind, a.md));
    PIME_HIP_TRY(hipGetLastError());
    return PIME_OK;
}

template <int T>
static int wgs_per_cu(size_t lds_bytes) {
    const int by_lds = (int)((160 * 1024) / lds_bytes);
    const int by_regs = T <= 8 ? 2 : 1;
    return by_lds < by_regs ? (by_lds < 1 ? 1 : by_lds) : by_regs;
}

int grid16(int kind, int B, int md, int D, int Di) {
    const int ngroups = (B + k16Group - 1) / k16Group;
    size_t lds = 0;
    int per = 1;
    if (kind == MLP_MODULAR_ACTOR && md == 128) { lds = sizeof(float) * lds16m<8>(D, Di, true).total; per = wgs_per_cu<8>(lds); }
    else if (kind == MLP_MODULAR_ACTOR) { lds = sizeof(float) * lds16m<16>(D, Di, true).total; per = wgs_per_cu<16>(lds); }
    else if (md == 64) { lds = sizeof(float) * lds16<4>(D, true).total; per = wgs_per_cu<4>(lds); }
    else if (md == 128) { lds = sizeof(float) * lds16<8>(D, true).total; per = wgs_per_cu<8>(lds); }
    else { lds = sizeof(float) * lds16<16>(D, true).total; per = wgs_per_cu<16>(lds); }
    const int cap = 256 * per;
    return ngroups < cap ? ngroups : cap;
}

template <int T, int ACT>
static int launch_fwd16(const float* x, int M, int D, const float* img, float* out, hipStream_t s) {
    const size_t lds_bytes = sizeof(float) * (size_t)lds16<T>(D, false).total;
    static LdsLimit lds_limit;  // per instantiation
    PIME_RAISE_LDS(lds_limit, (mlp16_forward_kernel<T, ACT>), 160 * 1024);
    const int ngroups = (M + k16Group - 1) / k16Group;
    const int cap = 256 * wgs_per_cu<T>(lds_bytes);
    hipLaunchKernelGGL((mlp16_forward_kernel<T, ACT>), dim3(ngroups < cap ? ngroups : cap), dim3(k16Threads), lds_bytes, s, x, M,
                       D, img, out);
    PIME_HIP_TRY(hipGetLastError());
    return PIME_OK;
}

int launch_forward16(int kind, const float* x, int M, int D, int Di, int md, const float* img, float* out, hipStream_t s) {
    const int T = md / 16;
    if (kind == MLP_MODULAR_ACTOR) {
        PIME_REQUIRE(T == 16, "the 16-tile modular actor serves width 256, got %d", md);
        const size_t lds_bytes = sizeof(float) * (size_t)lds16m<16>(D, Di, false).total;
        static LdsLimit lds_limit;
        PIME_RAISE_LDS(lds_limit, (mlp16m_forward_kernel<16>), 160 * 1024);
        const int ngroups = (M + k16Group - 1) / k16Group;
        const int cap = 256 * wgs_per_cu<16>(lds_bytes);
        hipLaunchKernelGGL((mlp16m_forward_kernel<16>), dim3(ngroups < cap ? ngroups : cap), dim3(k16Threads), lds_bytes, s, x, M, D, Di,
                           img, out);
        PIME_HIP_TRY(hipGetLastError());
        return PIME_OK;
    }
#define PIME_F16(TT) \
    if (T == TT) return kind == MLP_CRITIC ? launch_fwd16<TT, 0>(x, M, D, img, out, s) : launch_fwd16<TT, 1>(x, M, D, img, out, s);
    PIME_F16(4) PIME_F16(8) PIME_F16(16)
#undef PIME_F16
    set_error("no 16-tile forward for width %d", md);
    return PIME_ERR_ARG;
}

template <int T, bool ACTOR, bool B3 = false>
static int launch_grad16(const PpoArgs& a, hipStream_t s) {
    const size_t lds_bytes = sizeof(float) * (size_t)lds16<T>(a.D, true).total;
    PIME_REQUIRE(lds_bytes <= 160 * 1024, "16-tile PPO kernel needs %zu B of LDS", lds_bytes);
    static LdsLimit lds_limit;  // per instantiation
    PIME_RAISE_LDS(lds_limit, (ppo16_kernel<T, ACTOR, B3>), 160 * 1024);
    hipLaunchKernelGGL((ppo16_kernel<T, ACTOR, B3>), dim3(grid16(ACTOR ? MLP_PLAIN_ACTOR : MLP_CRITIC, a.B, T * 16, a.D, 0)), dim3(k16Threads), lds_bytes, s, a);
    PIME_HIP_TRY(hipGetLastError());
    return PIME_OK;
}
template <int T, bool B3>
static int launch_grad16m(const PpoArgs& a, hipStream_t s) {
    const size_t lds_bytes = sizeof(float) * (size_t)lds16m<T>(a.D, a.Di, true).total;
    PIME_REQUIRE(lds_bytes <= 160 * 1024, "16-tile modular PPO kernel needs %zu B of LDS", lds_bytes);
    static LdsLimit lds_limit;  // per instantiation
    PIME_RAISE_LDS(lds_limit, (ppo16m_kernel<T, B3>), 160 * 1024);
    hipLaunchKernelGGL((ppo16m_kernel<T, B3>), dim3(grid16(MLP_MODULAR_ACTOR, a.B, T * 16, a.D, a.Di)), dim3(k16Threads), lds_bytes, s, a);
    PIME_HIP_TRY(hipGetLastError());
    return PIME_OK;
}

int launch_ppo16(int kind, int md, const PpoArgs& a, hipStream_t s) {
    const int T = md / 16;
    const bool b3 = b3_grad(kind, md, a.D, a.Di);
    if (kind == MLP_MODULAR_ACTOR) {
        PIME_REQUIRE(T == 16 || T == 8, "the 16-tile modular actor serves widths 128 and 256, got %d", md);
        if (T == 8) return b3 ? launch_grad16m<8, true>(a, s) : launch_grad16m<8, false>(a, s);
        return b3 ? launch_grad16m<16, true>(a, s) : launch_grad16m<16, false>(a, s);
    }
    if (b3 && T == 8) return kind == MLP_CRITIC ? launch_grad16<8, false, true>(a, s) : launch_grad16<8, true, true>(a, s);
    if (b3 && T == 16) return kind == MLP_CRITIC ? launch_grad16<16, false, true>(a, s) : launch_grad16<16, true, true>(a, s);
#define PIME_G16(TT) \
    if (T == TT) return kind == MLP_CRITIC ? launch_grad16<TT, false>(a, s) : launch_grad16<TT, true>(a, s);
    PIME_G16(4) PIME_G16(8) PIME_G16(16)
#undef PIME_G16
    set_error("no 16-tile PPO kernel for width %d", md);
    return PIME_ERR_ARG;
}

}  // namespace pime
