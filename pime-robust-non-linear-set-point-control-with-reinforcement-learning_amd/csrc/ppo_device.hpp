// Device helpers shared by the PPO gradient kernels (ppo_train.hip: split net + dW kernels; ppo_fused.hip: one
// kernel per net with the weight gradients formed on chip).
#pragma once
#include "mlp_device.hpp"

namespace pime {

constexpr float kLogSqrt2Pi = 0.91893853320467274178f;

// ---- backward image ------------------------------------------------------------------------------------------------
struct BwdLayout {
    int T, off[8], total;
};

__host__ __device__ inline BwdLayout bwd_layout(int kind, int D, int Di, int md) {
    BwdLayout L{};
    const int T = md / 32;
    L.T = T;
    int o = 0;
    auto seg = [&](int idx, int floats) { L.off[idx] = o; o = align4(o + floats); };
    if (kind == MLP_MODULAR_ACTOR) {
        const int Do = D - Di, H = T / 2;
        seg(0, T * 32 * (Do + 1));   // other_net.0 (FIRST, recomputed for act')
        seg(1, T * 32 * (Di + 1));   // integrator_net.0 (FIRST)
        seg(2, T * 32);              // net.2 weights (head, VEC)
        seg(3, T * 16 * 64 * T);     // net.0 transposed: dZn0 (T tiles) -> dcat (T tiles)
        seg(4, H * 16 * 64 * T);     // other_net.2 transposed: dZo2 (H tiles) -> dh_o1 (T tiles)
        seg(5, H * 16 * 64 * T);     // integrator_net.2 transposed
    } else {
        seg(0, T * 32 * (D + 1));    // net.0 (FIRST)
        seg(1, T * 32);              // net.6 weights (head, VEC)
        seg(2, T * 16 * 64 * T);     // net.4 transposed: dZ3 -> dH2
        seg(3, T * 16 * 64 * T);     // net.2 transposed: dZ2 -> dH1
    }
    L.total = o;
    return L;
}

// nn.Linear layout -> backward image (bwd_layout order)
__device__ inline void pack_backward_image(const PackArgs& a, float* __restrict__ out, int tid, int nthr) {
    const BwdLayout L = bwd_layout(a.kind, a.D, a.Di, a.md);
    const int T = L.T;
    if (a.kind == MLP_MODULAR_ACTOR) {
        const int Do = a.D - a.Di, H = T / 2;
        pack_first(out + L.off[0], a.p[0], a.p[1], Do, Do, 0, T, tid, nthr);
        pack_first(out + L.off[1], a.p[4], a.p[5], a.Di, a.Di, 0, T, tid, nthr);
        pack_vec(out + L.off[2], a.p[10], T, tid, nthr);
        pack_mfma_t(out + L.off[3], a.p[8], T, T, tid, nthr);
        pack_mfma_t(out + L.off[4], a.p[2], H, T, tid, nthr);
        pack_mfma_t(out + L.off[5], a.p[6], H, T, tid, nthr);
    } else {
        pack_first(out + L.off[0], a.p[0], a.p[1], a.D, a.D, 0, T, tid, nthr);
        pack_vec(out + L.off[1], a.p[6], T, tid, nthr);
        pack_mfma_t(out + L.off[2], a.p[4], T, T, tid, nthr);
        pack_mfma_t(out + L.off[3], a.p[2], T, T, tid, nthr);
    }
}

// ---- workspace ------------------------------------------------------------------------------------------------------
// stash[(tile * NT + t) * 1024 + r * 64 + lane]: register r of lane `lane` of stashed tile t of sample-tile `tile`.
// critic / plain actor, NT = 5T:  H2 [0,T)  H3 [T,2T)  dZ3 [2T,3T)  dZ2 [3T,4T)  dZ1 [4T,5T)
// modular actor,        NT = 6T:  cat [0,T) n0 [T,2T)  dZn0 [2T,3T) dZcat [3T,4T) dZo1 [4T,5T) dZi1 [5T,6T)
__host__ __device__ inline int stash_tiles(int kind, int T) { return kind == MLP_MODULAR_ACTOR ? 6 * T : 5 * T; }

template <int NTL>
__device__ __forceinline__ void stash_store(float* __restrict__ base, int lane, const f32x16 (&a)[NTL]) {
#pragma unroll
    for (int t = 0; t < NTL; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) base[(t * 16 + r) * 64 + lane] = a[t][r];
}
template <int NTL>
__device__ __forceinline__ void stash_load(const float* __restrict__ base, int lane, f32x16 (&a)[NTL]) {
#pragma unroll
    for (int t = 0; t < NTL; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) a[t][r] = base[(t * 16 + r) * 64 + lane];
}

template <int ACT>
__device__ __forceinline__ float act_grad_from_output(float h) {
    if constexpr (ACT == 0) return h > 0.f ? 1.f : 0.f;  // ReLU'
    else return fmaf(-h, h, 1.f);                         // tanh' = 1 - h^2 (explicit fma: -ffp-contract=off)
}

template <int NTL, int ACT>
__device__ __forceinline__ void times_act_grad(f32x16 (&d)[NTL], const f32x16 (&h)[NTL]) {
#pragma unroll
    for (int t = 0; t < NTL; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) d[t][r] *= act_grad_from_output<ACT>(h[t][r]);
}

// dH_L = w_head (x) dOut in accumulator layout
template <int NTL>
__device__ __forceinline__ void head_backward(const float* __restrict__ w, int lane, float dout, f32x16 (&d)[NTL]) {
    const int h = lane >> 5;
#pragma unroll
    for (int t = 0; t < NTL; ++t) {
        float wv[16];
        load16(vec_at<NTL>(w, t, h), wv);
#pragma unroll
        for (int r = 0; r < 16; ++r) d[t][r] = wv[r] * dout;
    }
}

// Sum over the 64 lanes on the DPP path (6 VALU ops, no LDS crossbar); the total is valid in LANE 63 only.
// (__shfl_xor compiles to ds_bpermute: six dependent LDS-pipe round trips per sum, ~0.5 us per sum and wave.)
__device__ __forceinline__ float wave_total_dpp(float v) {
#define PIME_DPP_ADD(x, ctrl, row_mask) \
    x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), ctrl, row_mask, 0xf, true))
    PIME_DPP_ADD(v, 0x111, 0xf);  // row_shr:1
    PIME_DPP_ADD(v, 0x112, 0xf);  // row_shr:2
    PIME_DPP_ADD(v, 0x114, 0xf);  // row_shr:4
    PIME_DPP_ADD(v, 0x118, 0xf);  // row_shr:8   -> lane 15 of every row holds the row sum
    PIME_DPP_ADD(v, 0x142, 0xa);  // row_bcast:15 into rows 1 and 3 -> lanes 31 / 63 hold the half sums
    PIME_DPP_ADD(v, 0x143, 0xc);  // row_bcast:31 into rows 2 and 3 -> lane 63 holds the total
#undef PIME_DPP_ADD
    return v;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

}  // namespace pime
