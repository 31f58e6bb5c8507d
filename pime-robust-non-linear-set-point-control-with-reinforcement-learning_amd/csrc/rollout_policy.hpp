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

// ---- QUAD: one 16-lane tile per WORKGROUP, a layer's output tiles split over its four waves ----------------------------------------
// For launches of at most 4 096 lanes (the water-tank configurations: 256 tiles of 16 lanes) even 16-lane tiles leave three of
// four SIMDs idle.  Here the four waves of a workgroup share ONE tile: each computes a quarter of a layer's output tiles (a quarter
// of the serial MFMA chain), the quarters meet in an 8 KB LDS buffer and every wave reads the whole activation back as the next
// layer's operand.  Two exchanges per env step (towers / first hidden layer, then the last hidden layer); two buffers, so that one
// LDS-only barrier per exchange orders both the publication and the reuse.  First layers and the head are computed by every wave
// (vector ALUs, a few hundred cycles).  The env arithmetic runs in all four waves on the same 16 lanes; wave 0 stores.
#define PIME_XCHG_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

constexpr int kQuadWaves = 4;
template <int T>
__host__ __device__ constexpr int quad_xchg_floats() { return 2 * (T * 2) * 64 * 4; }   // two buffers of T * 2 tiles x 64 lanes x 4

// this wave's share of out = act(W in + b): output tiles [t0, t0 + NT) of the layer's OT32 * 2
template <int KT32, int OT32, int NT, int ACT>
__device__ __forceinline__ void layer16_on32_part(const float* __restrict__ wp, const float* __restrict__ bp, int lane, int t0,
                                                  const f32x4_t (&in)[KT32 * 2], f32x4_t (&out)[NT]) {
    static_assert(NT == 1 || NT == 2, "one tile, or an (even, odd) pair");
    const int c = lane & 15, g = lane >> 4;
    // tile ot' = component ot' >> 1 of the OT32-wide word at lane32 = 16 (ot' & 1) + c + 32 (g & 1); NT == 2: t0 is even
    const int comp = t0 >> 1;
    const float* wl = wp + (size_t)(c + 32 * (g & 1) + 256 * (g >> 1) + (NT == 1 ? 16 * (t0 & 1) : 0)) * OT32 + comp;
#pragma unroll
    for (int n = 0; n < NT; ++n)
        out[n] = *reinterpret_cast<const f32x4_t*>(bp + (g & 1) * (OT32 * 16) + ((t0 + n) >> 1) * 16 + 8 * ((t0 + n) & 1) + 4 * (g >> 1));
#pragma unroll
    for (int ok = 0; ok < KT32 * 2; ++ok) {
        PIME_NO_HOIST();
#pragma unroll
        for (int rk = 0; rk < 4; ++rk) {
            const int word = ((ok >> 1) * 16 + rk + 8 * (ok & 1)) * 64;
            const float b = in[ok][rk];
            out[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(wl[(size_t)word * OT32], b, out[0], 0, 0, 0);
            if constexpr (NT == 2)
                out[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wl[(size_t)(word + 16) * OT32], b, out[1], 0, 0, 0);
        }
    }
#pragma unroll
    for (int n = 0; n < NT; ++n) activate4<ACT>(out[n]);
}

template <int NTILES>
__device__ __forceinline__ void xchg_read(const float* __restrict__ buf, int lane, f32x4_t (&v)[NTILES]) {
#pragma unroll
    for (int t = 0; t < NTILES; ++t) v[t] = *reinterpret_cast<const f32x4_t*>(buf + (t * 64 + lane) * 4);
}
__device__ __forceinline__ void xchg_write(float* __restrict__ buf, int lane, int tile, const f32x4_t& v) {
    *reinterpret_cast<f32x4_t*>(buf + (tile * 64 + lane) * 4) = v;
}

// a_avg for this lane's sample; every wave of the workgroup returns the same value.  xbuf: quad_xchg_floats<T>() floats of LDS.
template <int T, int KIND, int D, int Di>
__device__ __forceinline__ float policy_forward16q(const float* __restrict__ lds, float* __restrict__ xbuf, const MlpLayout& L,
                                                   const float (&obs)[D], int lane, int wave) {
    constexpr int Do = D - Di, H = T / 2 > 0 ? T / 2 : 1, NT = T * 2;   // NT tiles of 16 features per full-width activation
    static_assert(T == 4 || T == 2, "widths 128 and 64");
    constexpr int PER = NT / kQuadWaves;                                 // output tiles per wave of a full-width layer: 2 (1 at width 64)
    const int g = lane >> 4;
    float* const buf0 = xbuf;
    float* const buf1 = xbuf + NT * 64 * 4;
    f32x4_t act[NT];
    if constexpr (KIND == MLP_MODULAR_ACTOR) {
        constexpr int PT = (H * 2) / kQuadWaves > 0 ? (H * 2) / kQuadWaves : 1;   // tiles per wave of a tower's H * 2
        static_assert(H * 2 >= kQuadWaves || T == 2, "towers of at least four tiles, or width 64 (two tiles: waves 0-1 and 2-3 split the towers)");
        {
            f32x4_t a0[NT], o[PT];
            first16_on32<T, 1, Do>(lds + L.off[0], obs, g, a0);
            PIME_NO_HOIST();
            if constexpr (T == 4) {
                layer16_on32_part<T, H, PT, 1>(lds + L.off[1], lds + L.off[2], lane, wave * PT, a0, o);
                xchg_write(buf0, lane, wave * PT, o[0]);
                PIME_NO_HOIST();
                first16_on32<T, 1, Di>(lds + L.off[3], obs + Do, g, a0);
                PIME_NO_HOIST();
                layer16_on32_part<T, H, PT, 1>(lds + L.off[4], lds + L.off[5], lane, wave * PT, a0, o);
                xchg_write(buf0, lane, H * 2 + wave * PT, o[0]);
            } else {   // width 64: each tower has two output tiles; waves 0-1 take other_net's, waves 2-3 integrator_net's
                const int tw = wave >> 1, tile = wave & 1;
                if (tw == 1) first16_on32<T, 1, Di>(lds + L.off[3], obs + Do, g, a0);
                layer16_on32_part<T, H, 1, 1>(lds + L.off[tw ? 4 : 1], lds + L.off[tw ? 5 : 2], lane, tile, a0, o);
                xchg_write(buf0, lane, tw * (H * 2) + tile, o[0]);
            }
        }
        PIME_XCHG_BARRIER();
        xchg_read<NT>(buf0, lane, act);   // cat
        {
            f32x4_t o[PER];
            PIME_NO_HOIST();
            layer16_on32_part<T, T, PER, 1>(lds + L.off[6], lds + L.off[7], lane, wave * PER, act, o);
#pragma unroll
            for (int n = 0; n < PER; ++n) xchg_write(buf1, lane, wave * PER + n, o[n]);
        }
        PIME_XCHG_BARRIER();
        xchg_read<NT>(buf1, lane, act);   // n0
        PIME_NO_HOIST();
        return head16_on32<T>(lds + L.off[8], lds[L.off[9]], lane, act);
    } else {
        constexpr int ACT = KIND == MLP_CRITIC ? 0 : 1;
        f32x4_t o[PER];
        {
            f32x4_t a0[NT];
            first16_on32<T, ACT, D>(lds + L.off[0], obs, g, a0);
            PIME_NO_HOIST();
            layer16_on32_part<T, T, PER, ACT>(lds + L.off[1], lds + L.off[2], lane, wave * PER, a0, o);
        }
#pragma unroll
        for (int n = 0; n < PER; ++n) xchg_write(buf0, lane, wave * PER + n, o[n]);
        PIME_XCHG_BARRIER();
        xchg_read<NT>(buf0, lane, act);
        PIME_NO_HOIST();
        layer16_on32_part<T, T, PER, ACT>(lds + L.off[3], lds + L.off[4], lane, wave * PER, act, o);
#pragma unroll
        for (int n = 0; n < PER; ++n) xchg_write(buf1, lane, wave * PER + n, o[n]);
        PIME_XCHG_BARRIER();
        xchg_read<NT>(buf1, lane, act);
        PIME_NO_HOIST();
        return head16_on32<T>(lds + L.off[5], lds[L.off[6]], lane, act);
    }
}

}  // namespace pime
