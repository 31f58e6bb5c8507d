// The residual policy's mean for one 32-lane tile with the observation in registers: shared by the fused rollout kernel
// (rollout.hip) and the fused evaluation kernel (rollout_eval.hip), so both run literally the same forward.
// replaces: ActorResidualPPO / ActorPPO .net and ActorResidualIntegratorModularPPO .mean (/root/reference/elegantrl/
// net_residual.py:19-22,153-160) for the lanes of one wave.
#pragma once
#include "mlp_device.hpp"

namespace pime {

// first layer from a register-resident input of compile-time width
template <int OT, int ACT, int DIN>
__device__ __forceinline__ void layer_first_regs(const float* __restrict__ w0, const float* x, int h, f32x16 (&out)[OT]) {
    const float* wb = w0 + DIN * (OT * 32);
#pragma unroll
    for (int ot = 0; ot < OT; ++ot) {
        float v[16];
        load16(vec_at<OT>(wb, ot, h), v);
#pragma unroll
        for (int r = 0; r < 16; ++r) out[ot][r] = v[r];
    }
#pragma unroll
    for (int j = 0; j < DIN; ++j) {
        const float* wj = w0 + j * (OT * 32);
#pragma unroll
        for (int ot = 0; ot < OT; ++ot) {
            float v[16];
            load16(vec_at<OT>(wj, ot, h), v);
#pragma unroll
            for (int r = 0; r < 16; ++r) out[ot][r] = fmaf(x[j], v[r], out[ot][r]);
        }
    }
#pragma unroll
    for (int ot = 0; ot < OT; ++ot)
#pragma unroll
        for (int r = 0; r < 16; ++r) out[ot][r] = activate<ACT>(out[ot][r]);
}

// a_avg = mean(obs) for this lane's sample (both lane halves carry the same 32 samples).  lds: the packed forward image
// (mlp_layout order); D = observation width, Di = integrator columns (modular actor), T = width / 32.
template <int T, int KIND, int D, int Di>
__device__ __forceinline__ float policy_forward(const float* __restrict__ lds, const MlpLayout& L, const float (&obs)[D], int lane) {
    constexpr int Do = D - Di, H = T / 2 > 0 ? T / 2 : 1;
    const int h = lane >> 5;
    if constexpr (KIND == MLP_MODULAR_ACTOR) {
        f32x16 cat[T];
        {
            f32x16 a0[T];
            layer_first_regs<T, 2, Do>(lds + L.off[0], obs, h, a0);   // activations are applied by the consuming layer
            PIME_NO_HOIST();
            layer_mfma_in<T, H, 2, 1>(lds + L.off[1], lds + L.off[2], lane, a0, *reinterpret_cast<f32x16(*)[H]>(&cat[0]));
        }
        {
            f32x16 a0[T];
            PIME_NO_HOIST();
            layer_first_regs<T, 2, Di>(lds + L.off[3], obs + Do, h, a0);
            PIME_NO_HOIST();
            layer_mfma_in<T, H, 2, 1>(lds + L.off[4], lds + L.off[5], lane, a0, *reinterpret_cast<f32x16(*)[H]>(&cat[H]));
        }
        f32x16 n0[T];
        PIME_NO_HOIST();
        layer_mfma_in<T, T, 1, 1>(lds + L.off[6], lds + L.off[7], lane, cat, n0);
        PIME_NO_HOIST();
        return layer_head<T>(lds + L.off[8], lds[L.off[9]], lane, n0);
    } else {   // plain actor (Tanh) or the TD3 Actor, whose shape and ReLUs are CriticAdv's (net.py:96-110 / :274-277): KIND MLP_CRITIC
        constexpr int ACT = KIND == MLP_CRITIC ? 0 : 1;
        f32x16 a0[T], a1[T];
        layer_first_regs<T, 2, D>(lds + L.off[0], obs, h, a0);
        PIME_NO_HOIST();
        layer_mfma_in<T, T, 2, ACT>(lds + L.off[1], lds + L.off[2], lane, a0, a1);
        PIME_NO_HOIST();
        layer_mfma_in<T, T, ACT, ACT>(lds + L.off[3], lds + L.off[4], lane, a1, a0);
        PIME_NO_HOIST();
        return layer_head<T>(lds + L.off[5], lds[L.off[6]], lane, a0);
    }
}

// ---- the same policy for 16-lane tiles (v_mfma_f32_16x16x4_f32) on the SAME packed image -----------------------------------------
// A 32-lane tile per wave gives 512 waves for 16 384 lanes: half of the chip's 1 024 SIMDs idle while every wave is latency-bound on
// its own serial MFMA chain.  With 16 lanes per wave every SIMD has a wave and a layer's chain is half as
// long (16 x 128 x 128 MACs at the same 32 MAC/cycle).  No second image: the 32x32x2 layout is re-addressed.  Accumulator layout of
// 16x16x4: lane (c = lane & 15, g = lane >> 4) holds sample c, output rows 4 g + r (r = 0..3) of every 16-row tile ot', i.e. feature
// 16 ot' + 4 g + r -- which is also what the lane supplies as B operand of k-step (ot_k, r_k): k = g <-> feature 16 ot_k + 4 g + r_k.
// Its A value for output tile ot' is W[16 ot' + c][16 ot_k + 4 g + r_k]; in pack_mfma's image (mlp_device.hpp) that element sits at
//   float-OT word ((kt 16 + s) 64 + lane32),  kt = ot_k >> 1,  s = r_k + 8 (ot_k & 1) + 4 (g >> 1),  lane32 = 16 (ot' & 1) + c + 32 (g & 1),
// component ot' >> 1: two OT-wide LDS reads per k-step (ot' even / odd) feed all the step's MFMAs.  Vectors (pack_vec / pack_first
// rows): the lane's four features of tile ot' are the 16-byte word at (g & 1) OT 16 + (ot' >> 1) 16 + 8 (ot' & 1) + 4 (g >> 1).
typedef float f32x4_t __attribute__((ext_vector_type(4)));

template <int OT32>
__device__ __forceinline__ int vec16_off(int ot, int g) { return (g & 1) * (OT32 * 16) + (ot >> 1) * 16 + 8 * (ot & 1) + 4 * (g >> 1); }

template <int ACT>
__device__ __forceinline__ void activate4(f32x4_t& v) {
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = activate<ACT>(v[r]);
}

// first layer (vector ALUs) from a register-resident input: out = act(W x + b), OT32 * 2 tiles of 16 features
template <int OT32, int ACT, int DIN>
__device__ __forceinline__ void first16_on32(const float* __restrict__ w0, const float* x, int g, f32x4_t (&out)[OT32 * 2]) {
#pragma unroll
    for (int ot = 0; ot < OT32 * 2; ++ot) out[ot] = *reinterpret_cast<const f32x4_t*>(w0 + DIN * (OT32 * 32) + vec16_off<OT32>(ot, g));
#pragma unroll
    for (int j = 0; j < DIN; ++j)
#pragma unroll
        for (int ot = 0; ot < OT32 * 2; ++ot) {
            const f32x4_t w = *reinterpret_cast<const f32x4_t*>(w0 + j * (OT32 * 32) + vec16_off<OT32>(ot, g));
#pragma unroll
            for (int r = 0; r < 4; ++r) out[ot][r] = fmaf(x[j], w[r], out[ot][r]);
        }
#pragma unroll
    for (int ot = 0; ot < OT32 * 2; ++ot) activate4<ACT>(out[ot]);
}

// out = act(W in + b) on the matrix cores: in = KT32 * 2 tiles of 16 features, out = OT32 * 2 tiles
template <int KT32, int OT32, int ACT>
__device__ __forceinline__ void layer16_on32(const float* __restrict__ wp, const float* __restrict__ bp, int lane,
                                             const f32x4_t (&in)[KT32 * 2], f32x4_t (&out)[OT32 * 2]) {
    using Frag = typename WFrag<OT32>::type;
    const int c = lane & 15, g = lane >> 4;
#pragma unroll
    for (int ot = 0; ot < OT32 * 2; ++ot) out[ot] = *reinterpret_cast<const f32x4_t*>(bp + vec16_off<OT32>(ot, g));
    const Frag* wl = reinterpret_cast<const Frag*>(wp) + c + 32 * (g & 1) + 256 * (g >> 1);
#pragma unroll
    for (int ok = 0; ok < KT32 * 2; ++ok) {
        PIME_NO_HOIST();   // bound the fragment prefetch depth
#pragma unroll
        for (int rk = 0; rk < 4; ++rk) {
            const int word = ((ok >> 1) * 16 + rk + 8 * (ok & 1)) * 64;
            const Frag w0 = wl[word], w1 = wl[word + 16];
            const float b = in[ok][rk];
#pragma unroll
            for (int o = 0; o < OT32; ++o) {
                out[2 * o] = __builtin_amdgcn_mfma_f32_16x16x4f32(wfrag_get(w0, o), b, out[2 * o], 0, 0, 0);
                out[2 * o + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wfrag_get(w1, o), b, out[2 * o + 1], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int ot = 0; ot < OT32 * 2; ++ot) activate4<ACT>(out[ot]);
}

template <int KT32>
__device__ __forceinline__ float head16_on32(const float* __restrict__ w, float bias, int lane, const f32x4_t (&in)[KT32 * 2]) {
    const int g = lane >> 4;
    float acc = 0.f;
#pragma unroll
    for (int ot = 0; ot < KT32 * 2; ++ot) {
        const f32x4_t wv = *reinterpret_cast<const f32x4_t*>(w + vec16_off<KT32>(ot, g));
#pragma unroll
        for (int r = 0; r < 4; ++r) acc = fmaf(in[ot][r], wv[r], acc);
    }
    acc += __shfl_xor(acc, 16);   // the four lane groups hold disjoint quarters of the features
    acc += __shfl_xor(acc, 32);
    return acc + bias;
}

// a_avg for this lane's sample (lane & 15; the four lane groups carry the same 16 samples)
template <int T, int KIND, int D, int Di>
__device__ __forceinline__ float policy_forward16(const float* __restrict__ lds, const MlpLayout& L, const float (&obs)[D], int lane) {
    constexpr int Do = D - Di, H = T / 2 > 0 ? T / 2 : 1;
    static_assert(T >= 2 && T % 2 == 0, "whole 32-feature tiles in both towers");
    const int g = lane >> 4;
    if constexpr (KIND == MLP_MODULAR_ACTOR) {
        f32x4_t cat[T * 2];
        {
            f32x4_t a0[T * 2];
            first16_on32<T, 1, Do>(lds + L.off[0], obs, g, a0);
            PIME_NO_HOIST();
            layer16_on32<T, H, 1>(lds + L.off[1], lds + L.off[2], lane, a0, *reinterpret_cast<f32x4_t(*)[H * 2]>(&cat[0]));
        }
        {
            f32x4_t a0[T * 2];
            PIME_NO_HOIST();
            first16_on32<T, 1, Di>(lds + L.off[3], obs + Do, g, a0);
            PIME_NO_HOIST();
            layer16_on32<T, H, 1>(lds + L.off[4], lds + L.off[5], lane, a0, *reinterpret_cast<f32x4_t(*)[H * 2]>(&cat[H * 2]));
        }
        f32x4_t n0[T * 2];
        PIME_NO_HOIST();
        layer16_on32<T, T, 1>(lds + L.off[6], lds + L.off[7], lane, cat, n0);
        PIME_NO_HOIST();
        return head16_on32<T>(lds + L.off[8], lds[L.off[9]], lane, n0);
    } else {
        constexpr int ACT = KIND == MLP_CRITIC ? 0 : 1;
        f32x4_t a0[T * 2], a1[T * 2];
        first16_on32<T, ACT, D>(lds + L.off[0], obs, g, a0);
        PIME_NO_HOIST();
        layer16_on32<T, T, ACT>(lds + L.off[1], lds + L.off[2], lane, a0, a1);
        PIME_NO_HOIST();
        layer16_on32<T, T, ACT>(lds + L.off[3], lds + L.off[4], lane, a1, a0);
        PIME_NO_HOIST();
        return head16_on32<T>(lds + L.off[5], lds[L.off[6]], lane, a0);
    }
}

}  // namespace pime
