// Device-side building blocks shared by the fused MLP forward (mlp_mfma.hip) and the fused PPO minibatch
// gradient kernels (ppo_train.hip): packed-image layout, MFMA layer chain, first layer / head, tanh.
#pragma once
#include "pime_common.hpp"
#include "ppo_train.hpp"

namespace pime {

using f32x16 = __attribute__((ext_vector_type(16))) float;

enum { MLP_CRITIC = 0, MLP_PLAIN_ACTOR = 1, MLP_MODULAR_ACTOR = 2 };

// feature index inside a 32-wide tile held by accumulator register s of lane-half h (C/D map of 32x32 MFMA)
__host__ __device__ __forceinline__ constexpr int feat32(int s, int h) { return (s & 3) + 8 * (s >> 2) + 4 * h; }

// ---- packed image layout (identical in HBM and LDS) -------------------------------------------------------
// FIRST  layer (D_in -> 32*OT):   [D_in+1][2][OT][16]        column j of W for feature feat(ot,r,h); row D_in = bias
//                                 (j outermost; lane half h next, so that the 16 values a lane reads per output tile are
//                                 contiguous: four ds_read_b128 instead of sixteen ds_read_b32 -- round 3; LDS instructions cost
//                                 SIMD issue slots, not bytes)
// MFMA   layer (32*KT -> 32*OT):  [KT*16][64][OT]            W[ot*32 + (lane&31)][kt*32 + feat32(s, lane>>5)]
// VEC    (bias / head weights):   [2][OT][16]                v[ot*32 + feat32(r,h)] at h*OT*16 + ot*16 + r
struct MlpLayout {
    int T;          // md / 32
    int off[12];    // float offsets of the segments
    int total;      // floats (multiple of 4)
};

__host__ __device__ inline int align4(int v) { return (v + 3) & ~3; }

__host__ __device__ inline MlpLayout mlp_layout(int kind, int D, int Di, int md) {
    MlpLayout L{};
    const int T = md / 32;
    L.T = T;
    int o = 0;
    auto seg = [&](int idx, int floats) { L.off[idx] = o; o = align4(o + floats); };
    if (kind == MLP_MODULAR_ACTOR) {
        const int Do = D - Di, H = T / 2;
        seg(0, T * 32 * (Do + 1));     // other_net.0 (FIRST)
        seg(1, T * 16 * 64 * H);       // other_net.2 (MFMA md -> md/2)
        seg(2, H * 32);                // other_net.2 bias
        seg(3, T * 32 * (Di + 1));     // integrator_net.0 (FIRST)
        seg(4, T * 16 * 64 * H);       // integrator_net.2
        seg(5, H * 32);
        seg(6, T * 16 * 64 * T);       // net.0 (MFMA md -> md)
        seg(7, T * 32);                // net.0 bias
        seg(8, T * 32);                // net.2 weights (HEAD)
        seg(9, 4);                     // net.2 bias
    } else {
        seg(0, T * 32 * (D + 1));      // net.0 (FIRST)
        seg(1, T * 16 * 64 * T);       // net.2
        seg(2, T * 32);
        seg(3, T * 16 * 64 * T);       // net.4
        seg(4, T * 32);
        seg(5, T * 32);                // net.6 weights (HEAD)
        seg(6, 4);                     // net.6 bias
    }
    L.total = o;
    return L;
}

__device__ inline void pack_first(float* dst, const float* W, const float* b, int Din, int ldw, int col0, int OT,
                                  int tid, int nthr) {
    const int per = OT * 32, n = per * (Din + 1);
    for (int idx = tid; idx < n; idx += nthr) {
        const int j = idx / per, q = idx % per;
        const int h = q / (OT * 16), r = q & 15, ot = (q >> 4) % OT;
        const int f = ot * 32 + feat32(r, h);
        dst[idx] = j < Din ? W[(size_t)f * ldw + col0 + j] : b[f];
    }
}

__device__ inline void pack_mfma(float* dst, const float* W, int KT, int OT, int tid, int nthr) {
    const int K = KT * 32, n = KT * 16 * 64 * OT;
    for (int idx = tid; idx < n; idx += nthr) {
        const int ot = idx % OT, lane = (idx / OT) & 63, ks = idx / (OT * 64);
        const int kt = ks >> 4, s = ks & 15;
        dst[idx] = W[(size_t)(ot * 32 + (lane & 31)) * K + kt * 32 + feat32(s, lane >> 5)];
    }
}

// The same image for the TRANSPOSED matrix: the backward chain dH_in = W^T dZ is a forward layer whose weight is
// V = W^T ([K_in x O] seen as out x in).  W is nn.Linear [O][K]; KT = O/32 input tiles, OT = K/32 output tiles.
__device__ inline void pack_mfma_t(float* dst, const float* W, int KT, int OT, int tid, int nthr) {
    const int Kin = OT * 32, n = KT * 16 * 64 * OT;  // W row length = K_in of the forward layer
    for (int idx = tid; idx < n; idx += nthr) {
        const int ot = idx % OT, lane = (idx / OT) & 63, ks = idx / (OT * 64);
        const int kt = ks >> 4, s = ks & 15;
        // V[ot*32 + i][kt*32 + feat] = W[kt*32 + feat][ot*32 + i]
        dst[idx] = W[(size_t)(kt * 32 + feat32(s, lane >> 5)) * Kin + ot * 32 + (lane & 31)];
    }
}

__device__ inline void pack_vec(float* dst, const float* v, int OT, int tid, int nthr) {
    for (int idx = tid; idx < OT * 32; idx += nthr) {
        const int h = idx / (OT * 16), r = idx & 15, ot = (idx >> 4) % OT;
        dst[idx] = v[ot * 32 + feat32(r, h)];
    }
}

// nn.Linear layout -> forward image (mlp_layout order)
__device__ inline void pack_forward_image(const PackArgs& a, float* __restrict__ out, int tid, int nthr) {
    const MlpLayout L = mlp_layout(a.kind, a.D, a.Di, a.md);
    const int T = L.T;
    if (a.kind == MLP_MODULAR_ACTOR) {
        const int Do = a.D - a.Di, H = T / 2;
        pack_first(out + L.off[0], a.p[0], a.p[1], Do, Do, 0, T, tid, nthr);
        pack_mfma(out + L.off[1], a.p[2], T, H, tid, nthr);
        pack_vec(out + L.off[2], a.p[3], H, tid, nthr);
        pack_first(out + L.off[3], a.p[4], a.p[5], a.Di, a.Di, 0, T, tid, nthr);
        pack_mfma(out + L.off[4], a.p[6], T, H, tid, nthr);
        pack_vec(out + L.off[5], a.p[7], H, tid, nthr);
        pack_mfma(out + L.off[6], a.p[8], T, T, tid, nthr);
        pack_vec(out + L.off[7], a.p[9], T, tid, nthr);
        pack_vec(out + L.off[8], a.p[10], T, tid, nthr);
        if (tid == 0) out[L.off[9]] = a.p[11][0];
    } else {
        pack_first(out + L.off[0], a.p[0], a.p[1], a.D, a.D, 0, T, tid, nthr);
        pack_mfma(out + L.off[1], a.p[2], T, T, tid, nthr);
        pack_vec(out + L.off[2], a.p[3], T, tid, nthr);
        pack_mfma(out + L.off[3], a.p[4], T, T, tid, nthr);
        pack_vec(out + L.off[4], a.p[5], T, tid, nthr);
        pack_vec(out + L.off[5], a.p[6], T, tid, nthr);
        if (tid == 0) out[L.off[6]] = a.p[7][0];
    }
}

// hipcc treats the (read-only after staging) LDS image as loop invariant and hoists bias/head/first-layer reads
// out of the persistent tile loop, then spills them (1.2 KB/lane of scratch).  A compiler-only memory barrier per
// tile and per layer keeps each read next to its use.
#define PIME_NO_HOIST() asm volatile("" ::: "memory")

// ---- forward --------------------------------------------------------------------------------------------------
// Copy a packed image (n4 float4s) from HBM/L2 into LDS with every load of a batch in flight before the first LDS
// write.  The naive `dst[i] = src[i]` loop compiles to load -> wait -> ds_write per iteration, i.e. 16 exposed L2
// round trips (~16 us) for a 131 KB image with 512 threads; batching 8 loads per thread leaves two.
template <int K>
__device__ __forceinline__ void stage_batches(float4* __restrict__ dst, const float4* __restrict__ src, int n4, int nthr, int& i) {
    for (; i + (K - 1) * nthr < n4; i += nthr * K) {   // full batches only: no guards, v[] stays in registers
        float4 v[K];
#pragma unroll
        for (int k = 0; k < K; ++k) v[k] = src[i + k * nthr];
#pragma unroll
        for (int k = 0; k < K; ++k) dst[i + k * nthr] = v[k];
    }
}
__device__ __forceinline__ void stage_image(float* __restrict__ lds, const float* __restrict__ image, int n4) {
    const float4* src = reinterpret_cast<const float4*>(image);
    float4* dst = reinterpret_cast<float4*>(lds);
    const int nthr = blockDim.x;
    int i = threadIdx.x;
    // batches of 8, then the remainder in batches of 4 / 2 / 1 (a guarded partial batch of 8 made hipcc keep v[] on the stack:
    // 144 B of scratch per lane in every kernel that stages an image, and scratch reloads wait on the shared vmcnt counter)
    stage_batches<8>(dst, src, n4, nthr, i);
    stage_batches<4>(dst, src, n4, nthr, i);
    stage_batches<2>(dst, src, n4, nthr, i);
    stage_batches<1>(dst, src, n4, nthr, i);
}

// tanh for the hidden layers in 5 VALU ops, two of them transcendental (ocml's tanhf inlined 64x per layer drove the
// kernel to the 256-VGPR cap and is ~40 ops): 1 - 2/(e^{2x}+1) via v_exp_f32 / v_rcp_f32.  The formula holds for either sign
// (x -> -inf: e -> 0, t -> -1; x > 44: e = inf, rcp -> 0, t -> 1), so no |x| / copysign pair around it (rounds 1-2 had them: 7
// ops; with f32 MFMAs every vector instruction is time on the SIMD -- the modular actor's gradient kernel evaluates 514 of
// these per lane).  Absolute error < 2e-7 everywhere; the RELATIVE error grows below |x| ~ 1e-3 (cancellation against 1),
// which is immaterial for a hidden unit that feeds a dot product (an earlier version avoided it with a 9th-order polynomial
// branch for |x| < 0.25: +6 VALU ops per activation, ~10 us per actor gradient kernel, no measurable change in any parity test).
// The ENV action tanh (agent_residual.py:61) does not use this: it is the float64 tanh rounded once (env_device.hpp).
__device__ __forceinline__ float fast_tanh(float x) {
    const float e = __builtin_amdgcn_exp2f(x * 2.885390081777927f);  // e^{2x}
    return fmaf(-2.0f, __builtin_amdgcn_rcpf(e + 1.0f), 1.0f);   // (explicit fma: the library is built with -ffp-contract=off)
}

typedef float f32x2_t __attribute__((ext_vector_type(2)));   // packed f32 pairs (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32)

template <int ACT>
__device__ __forceinline__ float activate(float v) {
    if constexpr (ACT == 0) return v > 0.f ? v : 0.f;  // nn.ReLU
    else if constexpr (ACT == 1) return fast_tanh(v);   // nn.Tanh
    else return v;                                      // identity (backward chains)
}

// the 16 values of lane half h for output tile ot of a VEC / FIRST row (16-byte aligned: segments are multiples of 4 floats)
__device__ __forceinline__ void load16(const float* __restrict__ p, float (&v)[16]) {
    const float4* q = reinterpret_cast<const float4*>(p);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const float4 w = q[g];
        v[4 * g] = w.x; v[4 * g + 1] = w.y; v[4 * g + 2] = w.z; v[4 * g + 3] = w.w;
    }
}
template <int OT>
__device__ __forceinline__ const float* vec_at(const float* __restrict__ v, int ot, int h) { return v + h * (OT * 16) + ot * 16; }

// y[ot][r] = act( b[f] + sum_j x[m][col0 + j] * W[f][j] ),  f = ot*32 + feat32(r, h)
template <int OT, int ACT>
__device__ __forceinline__ void layer_first(const float* __restrict__ w0, const float* __restrict__ xrow, int Din,
                                            int h, f32x16 (&out)[OT]) {
    const float* wb = w0 + Din * (OT * 32);  // bias row
#pragma unroll
    for (int ot = 0; ot < OT; ++ot) {
        float v[16];
        load16(vec_at<OT>(wb, ot, h), v);
#pragma unroll
        for (int r = 0; r < 16; ++r) out[ot][r] = v[r];
    }
    for (int j = 0; j < Din; ++j) {
        const float xj = xrow[j];
        const float* wj = w0 + j * (OT * 32);
#pragma unroll
        for (int ot = 0; ot < OT; ++ot) {
            float v[16];
            load16(vec_at<OT>(wj, ot, h), v);
#pragma unroll
            for (int r = 0; r < 16; ++r) out[ot][r] = fmaf(xj, v[r], out[ot][r]);
        }
    }
#pragma unroll
    for (int ot = 0; ot < OT; ++ot)
#pragma unroll
        for (int r = 0; r < 16; ++r) out[ot][r] = activate<ACT>(out[ot][r]);
}

template <int OT>
struct WFrag;
template <>
struct WFrag<4> { using type = float4; };
template <>
struct WFrag<2> { using type = float2; };
template <>
struct WFrag<1> { using type = float; };

__device__ __forceinline__ float wfrag_get(const float4& v, int i) { return i == 0 ? v.x : i == 1 ? v.y : i == 2 ? v.z : v.w; }
__device__ __forceinline__ float wfrag_get(const float2& v, int i) { return i == 0 ? v.x : v.y; }
__device__ __forceinline__ float wfrag_get(const float& v, int) { return v; }

// out = act( W * in + b ) on the matrix cores.  in: KT tiles of 32 features, out: OT tiles.
template <int KT, int OT, int ACT, bool HAS_BIAS = true>
__device__ __forceinline__ void layer_mfma(const float* __restrict__ wp, const float* __restrict__ bp, int lane,
                                           const f32x16 (&in)[KT], f32x16 (&out)[OT]) {
    using Frag = typename WFrag<OT>::type;
    const int h = lane >> 5;
#pragma unroll
    for (int ot = 0; ot < OT; ++ot) {
        float bv[16];
        if constexpr (HAS_BIAS) load16(vec_at<OT>(bp, ot, h), bv);
#pragma unroll
        for (int r = 0; r < 16; ++r) out[ot][r] = HAS_BIAS ? bv[r] : 0.f;
    }
    const Frag* wl = reinterpret_cast<const Frag*>(wp) + lane;
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            if ((s & 3) == 0) PIME_NO_HOIST();  // bound the W-fragment prefetch depth to 4 k-steps (<= 16 VGPRs)
            const Frag w = wl[(kt * 16 + s) * 64];
            const float b = in[kt][s];
#pragma unroll
            for (int ot = 0; ot < OT; ++ot)
                out[ot] = __builtin_amdgcn_mfma_f32_32x32x2f32(wfrag_get(w, ot), b, out[ot], 0, 0, 0);
        }
    }
#pragma unroll
    for (int ot = 0; ot < OT; ++ot)
#pragma unroll
        for (int r = 0; r < 16; ++r) out[ot][r] = activate<ACT>(out[ot][r]);
}

// out = act_out( W * act_in(in) + b ): the same layer with the INPUT's activation applied inside the k-loop, in place
// (afterwards `in` holds the activated values).  A tanh costs ~50 VALU cycles; done in the producing layer's epilogue
// the 64 of them are exposed, done here each one runs in the shadow of the previous k-step's OT MFMAs (64 cycles each).
template <int KT, int OT, int ACT_OUT, int ACT_IN, bool HAS_BIAS = true>
__device__ __forceinline__ void layer_mfma_in(const float* __restrict__ wp, const float* __restrict__ bp, int lane,
                                              f32x16 (&in)[KT], f32x16 (&out)[OT]) {
    using Frag = typename WFrag<OT>::type;
    const int h = lane >> 5;
#pragma unroll
    for (int ot = 0; ot < OT; ++ot) {
        float bv[16];
        if constexpr (HAS_BIAS) load16(vec_at<OT>(bp, ot, h), bv);
#pragma unroll
        for (int r = 0; r < 16; ++r) out[ot][r] = HAS_BIAS ? bv[r] : 0.f;
    }
    const Frag* wl = reinterpret_cast<const Frag*>(wp) + lane;
    in[0][0] = activate<ACT_IN>(in[0][0]);
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            if ((s & 3) == 0) PIME_NO_HOIST();  // bound the W-fragment prefetch depth to 4 k-steps (<= 16 VGPRs)
            const Frag w = wl[(kt * 16 + s) * 64];
            const float b = in[kt][s];
#pragma unroll
            for (int ot = 0; ot < OT; ++ot)
                out[ot] = __builtin_amdgcn_mfma_f32_32x32x2f32(wfrag_get(w, ot), b, out[ot], 0, 0, 0);
            // the next k-step's input, activated while this k-step's MFMAs run
            constexpr int kLast = KT * 16 - 1;
            const int nx = kt * 16 + s + 1;
            if (nx <= kLast) in[nx >> 4][nx & 15] = activate<ACT_IN>(in[nx >> 4][nx & 15]);
        }
    }
#pragma unroll
    for (int ot = 0; ot < OT; ++ot)
#pragma unroll
        for (int r = 0; r < 16; ++r) out[ot][r] = activate<ACT_OUT>(out[ot][r]);
}

// Backward chain step into a FIRST layer: out = (W * in) (.) act'(h1), where h1 = act(v) is the first-layer
// activation whose pre-activations v the caller recomputed (layer_first<.., 2>).  The OT*16 activations (a tanh is ~50
// VALU cycles) are evaluated inside the k-loop, in the shadow of the MFMAs, instead of after it.
// V_IS_ACT: v already holds the ACTIVATED h1 (read back from a stash) -- only act' = 1 - h1^2 / [h1 > 0] is formed here.
template <int KT, int OT, int ACT, bool V_IS_ACT = false>
__device__ __forceinline__ void layer_mfma_gate(const float* __restrict__ wp, int lane, const f32x16 (&in)[KT],
                                                f32x16 (&out)[OT], f32x16 (&v)[OT]) {
    using Frag = typename WFrag<OT>::type;
    constexpr int STEPS = KT * 16, NV = OT * 16, PERK = (NV + STEPS - 1) / STEPS;
#pragma unroll
    for (int ot = 0; ot < OT; ++ot)
#pragma unroll
        for (int r = 0; r < 16; ++r) out[ot][r] = 0.f;
    const Frag* wl = reinterpret_cast<const Frag*>(wp) + lane;
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            if ((s & 3) == 0) PIME_NO_HOIST();
            const Frag w = wl[(kt * 16 + s) * 64];
            const float b = in[kt][s];
#pragma unroll
            for (int ot = 0; ot < OT; ++ot)
                out[ot] = __builtin_amdgcn_mfma_f32_32x32x2f32(wfrag_get(w, ot), b, out[ot], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < PERK; ++q) {
                const int idx = (kt * 16 + s) * PERK + q;
                if (idx < NV) {
                    const float hq = V_IS_ACT ? v[idx >> 4][idx & 15] : activate<ACT>(v[idx >> 4][idx & 15]);
                    v[idx >> 4][idx & 15] = ACT == 0 ? (hq > 0.f ? 1.f : 0.f) : fmaf(-hq, hq, 1.f);
                }
            }
        }
    }
#pragma unroll
    for (int ot = 0; ot < OT; ++ot)
#pragma unroll
        for (int r = 0; r < 16; ++r) out[ot][r] *= v[ot][r];
}

template <int NT, int ACT>
__device__ __forceinline__ void activate_tiles(f32x16 (&v)[NT]) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) v[t][r] = activate<ACT>(v[t][r]);
}

template <int KT>
__device__ __forceinline__ float layer_head(const float* __restrict__ w, float bias, int lane, const f32x16 (&in)[KT]) {
    const int h = lane >> 5;
    float acc = 0.f;
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
        float wv[16];
        load16(vec_at<KT>(w, kt, h), wv);
#pragma unroll
        for (int r = 0; r < 16; ++r) acc = fmaf(in[kt][r], wv[r], acc);
    }
    acc += __shfl_xor(acc, 32);  // the other lane half holds the other 16 features of every tile
    return acc + bias;
}

}  // namespace pime
