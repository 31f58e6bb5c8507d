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

}  // namespace pime
