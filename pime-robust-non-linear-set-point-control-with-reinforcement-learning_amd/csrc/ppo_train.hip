// Fused PPO minibatch gradients on the gfx950 f32 matrix cores: the loss gradient and the whole fwd+bwd of the small
// actor / critic MLPs for one minibatch in three launches (critic net, actor net, weight gradients), replacing
// ~150 PyTorch/rocBLAS launches per optimizer step (profiles/r01_a_*: the skinny K=65536 rocBLAS GEMMs run at
// 200-250 us each and made the update 99 % of the headline metric's time).
//
// replaces (reference, /root/reference/elegantrl/agent.py:629-657): minibatch gather, ActorResidual*.compute_logprob
// (net_residual.py:48-54,182-190), clipped surrogate + entropy proxy (:637-645), CriticAdv forward + SmoothL1 (:648-649),
// `obj_united.backward()` (:654-655).  The optimizer step itself stays in PyTorch (torch.optim.Adam).
//
// Structure (DESIGN.md "ppo_minibatch_grad"):
//  net kernel, phase A (forward image in LDS):  per 32-sample tile, one wave: gather the rows by index, run the net with
//     activations in registers (mlp_device.hpp), stash the hidden activations the backward needs in HBM in fragment
//     order (every store is one contiguous 256-B segment), evaluate the loss and its gradient w.r.t. the net output.
//  net kernel, phase B (transposed image in LDS): the same wave re-reads its stash and chains dZ_l -> dH_{l-1} through
//     the matrix cores exactly like a forward (the weight operand is W^T), stashing every dZ_l.
//  dW kernel: for every layer the TN GEMM dW_l = dZ_l^T H_{l-1} over the sample axis.  The stashes are sample-on-lane;
//     the MFMA operands need feature-on-lane, so each 32x32 tile is transposed through a 33-float-pitch LDS image
//     (conflict-free both ways).  First-layer activations are recomputed from the 3-float state instead of stashed.
//     Partial products are combined with float atomics into the (zeroed) gradient tensors.
#include "ppo_device.hpp"
#include "ppo_train.hpp"

namespace pime {


using PackBwdArgs = PackArgs;

__global__ void mlp_pack_bwd_kernel(PackBwdArgs a, float* __restrict__ out) {
    pack_backward_image(a, out, blockIdx.x * blockDim.x + threadIdx.x, gridDim.x * blockDim.x);
}

constexpr int kTrainThreads = 512;

template <int T, int KIND>
__global__ __launch_bounds__(kTrainThreads) void ppo_fwd_bwd_kernel(PpoArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr bool MODULAR = KIND == MLP_MODULAR_ACTOR;
    constexpr bool CRITIC = KIND == MLP_CRITIC;
    constexpr int ACT = CRITIC ? 0 : 1;
    constexpr int NT = MODULAR ? 6 * T : 5 * T;
    constexpr int H = T / 2 > 0 ? T / 2 : 1;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, waves = blockDim.x >> 6, h = lane >> 5;
    const int ntiles = (a.B + 31) / 32;
    const float invB = 1.0f / (float)a.B;

    // ------------------------------------------------------------------ phase A: forward + loss gradient
    const MlpLayout L = mlp_layout(KIND, a.D, a.Di, T * 32);
    stage_image(lds, a.img_fwd, L.total / 4);
    __syncthreads();
    // The two waves that share a SIMD (w and w+4) run the same program; in lock-step their MFMA segments collide on
    // the matrix pipe and their VALU / memory segments leave it idle (PMC: pipe busy 47 %, waves issue-stalled 67 %).
    // Delaying the second group by about one MFMA segment makes the segments complementary.
    const int stagger = __builtin_amdgcn_readfirstlane(wave >= 4 ? a.stagger : 0);
    for (int k = 0; k < stagger; ++k) __builtin_amdgcn_s_sleep(127);  // 127 * 64 cycles each
    float s0 = 0.f, s1 = 0.f, gstd = 0.f;
    double m1 = 0.0, m2 = 0.0;
    for (int tile = blockIdx.x * waves + wave; tile < ntiles; tile += gridDim.x * waves) {
        PIME_NO_HOIST();
        const int pos = tile * 32 + (lane & 31);
        const bool valid = pos < a.B;
        const int64_t* const idx = a.indices + (a.index_row ? (size_t)a.index_row[0] * a.B : 0);
        const long long row = idx[valid ? pos : a.B - 1];
        const float* xrow = a.state + (size_t)row * a.D;
        float* st = a.stash + (size_t)tile * NT * 1024;
        // per-sample loss inputs: issued now so that their (index-dependent) latency hides behind the layers
        const float in_rsum = CRITIC ? a.r_sum[row] : 0.f;
        const float in_action = CRITIC ? 0.f : a.action[row];
        const float in_logprob = CRITIC ? 0.f : a.logprob[row];
        const float in_adv = CRITIC ? 0.f : a.adv[row];
        if (h == 0)  // contiguous copy of the gathered states for the dW kernel (its first-layer operands)
            for (int c = 0; c < a.D; ++c) a.xg[(size_t)pos * a.D + c] = xrow[c];
        float y;
        if constexpr (MODULAR) {
            const int Do = a.D - a.Di;
            f32x16 cat[T];
            {
                f32x16 a0[T];
                layer_first<T, 1>(lds + L.off[0], xrow, Do, h, a0);
                PIME_NO_HOIST();
                layer_mfma<T, H, 1>(lds + L.off[1], lds + L.off[2], lane, a0, *reinterpret_cast<f32x16(*)[H]>(&cat[0]));
            }
            {
                f32x16 a0[T];
                PIME_NO_HOIST();
                layer_first<T, 1>(lds + L.off[3], xrow + Do, a.Di, h, a0);
                PIME_NO_HOIST();
                layer_mfma<T, H, 1>(lds + L.off[4], lds + L.off[5], lane, a0, *reinterpret_cast<f32x16(*)[H]>(&cat[H]));
            }
            stash_store<T>(st, lane, cat);
            f32x16 n0[T];
            PIME_NO_HOIST();
            layer_mfma<T, T, 1>(lds + L.off[6], lds + L.off[7], lane, cat, n0);
            stash_store<T>(st + T * 1024, lane, n0);
            PIME_NO_HOIST();
            y = layer_head<T>(lds + L.off[8], lds[L.off[9]], lane, n0);
        } else {
            f32x16 a0[T], a1[T];
            layer_first<T, ACT>(lds + L.off[0], xrow, a.D, h, a0);
            PIME_NO_HOIST();
            layer_mfma<T, T, ACT>(lds + L.off[1], lds + L.off[2], lane, a0, a1);
            stash_store<T>(st, lane, a1);
            PIME_NO_HOIST();
            layer_mfma<T, T, ACT>(lds + L.off[3], lds + L.off[4], lane, a1, a0);
            stash_store<T>(st + T * 1024, lane, a0);
            PIME_NO_HOIST();
            y = layer_head<T>(lds + L.off[5], lds[L.off[6]], lane, a0);
        }
        // loss and d(loss)/d(net output) for this sample
        float dout = 0.f;
        if (valid) {
            if constexpr (CRITIC) {
                const float d = y - in_rsum, ad = fabsf(d);         // SmoothL1, beta = 1 (agent.py:567,649)
                const float l = ad < 1.f ? 0.5f * d * d : ad - 0.5f;
                const float g = ad < 1.f ? d : (d > 0.f ? 1.f : -1.f);
                dout = g * invB;  // unscaled: critic_scale_kernel applies 1/(std+1e-5) to the finished gradients
                if (h == 0) {
                    s0 += l;
                    m1 += (double)in_rsum;
                    m2 += (double)in_rsum * (double)in_rsum;
                }
            } else {
                const float asl = a.a_std_log[0], inv_sigma = __expf(-asl);
                const float z = (y - in_action) * inv_sigma;
                const float logp = -(asl + kLogSqrt2Pi + 0.5f * z * z);           // compute_logprob
                const float ratio = __expf(logp - in_logprob);
                const float lo = 1.f - a.ratio_clip, hi = 1.f + a.ratio_clip;
                const float clamped = fminf(fmaxf(ratio, lo), hi);
                const float adv = in_adv;
                const float u = adv * ratio, c = adv * clamped;                   // agent.py:639-641
                // torch.min backward: the smaller operand gets the gradient, ties split it; clamp passes it inside [lo,hi]
                const float w_u = u < c ? 1.f : (u == c ? 0.5f : 0.f);
                const float w_c = c < u ? 1.f : (u == c ? 0.5f : 0.f);
                const bool in_range = ratio >= lo && ratio <= hi;
                const float g_sur = w_u * u + (in_range ? w_c * u : 0.f);         // d min / d logp  (d ratio/d logp = ratio)
                const float p = __expf(logp);
                const float ent = p * logp;                                       // entropy proxy (:643)
                const float g_logp = (-g_sur + a.lambda_entropy * p * (logp + 1.f)) * invB;
                dout = g_logp * (-z * inv_sigma);                                 // d logp / d a_avg
                if (h == 0) {
                    gstd += g_logp * (z * z - 1.f);                               // d logp / d a_std_log
                    s0 += -fminf(u, c);
                    s1 += ent;
                }
            }
        }
        if (h == 0 && pos < ntiles * 32) a.dout[pos] = dout;
    }
    s0 = wave_sum(s0); s1 = wave_sum(s1); gstd = wave_sum(gstd);
    if constexpr (CRITIC) {
        for (int o = 32; o > 0; o >>= 1) { m1 += __shfl_xor(m1, o); m2 += __shfl_xor(m2, o); }
    }
    if (lane == 0) {
        if constexpr (CRITIC) {
            atomicAdd(&a.loss_sums[2], s0);
            atomicAdd(&a.moments[0], m1);
            atomicAdd(&a.moments[1], m2);
        }
        else { atomicAdd(&a.loss_sums[0], s0); atomicAdd(&a.loss_sums[1], s1); atomicAdd(a.g_std, gstd); }
    }

    // ------------------------------------------------------------------ phase B: backward chain
    __syncthreads();  // every wave is done reading the forward image
    const BwdLayout Lb = bwd_layout(KIND, a.D, a.Di, T * 32);
    stage_image(lds, a.img_bwd, Lb.total / 4);
    __syncthreads();
    for (int k = 0; k < stagger; ++k) __builtin_amdgcn_s_sleep(127);
    for (int tile = blockIdx.x * waves + wave; tile < ntiles; tile += gridDim.x * waves) {
        PIME_NO_HOIST();
        const int pos = tile * 32 + (lane & 31);
        const int64_t* const idx = a.indices + (a.index_row ? (size_t)a.index_row[0] * a.B : 0);
        const long long row = idx[pos < a.B ? pos : a.B - 1];
        const float* xrow = a.state + (size_t)row * a.D;
        float* st = a.stash + (size_t)tile * NT * 1024;
        const float dout = a.dout[pos];  // written by this very wave in phase A
        if constexpr (MODULAR) {
            const int Do = a.D - a.Di;
            f32x16 dcat[T];
            {
                f32x16 dn0[T], n0[T];
                head_backward<T>(lds + Lb.off[2], lane, dout, dn0);
                stash_load<T>(st + T * 1024, lane, n0);
                times_act_grad<T, 1>(dn0, n0);
                stash_store<T>(st + 2 * T * 1024, lane, dn0);                    // dZn0
                PIME_NO_HOIST();
                layer_mfma<T, T, 2, false>(lds + Lb.off[3], nullptr, lane, dn0, dcat);
            }
            {
                f32x16 cat[T];
                stash_load<T>(st, lane, cat);
                times_act_grad<T, 1>(dcat, cat);
                stash_store<T>(st + 3 * T * 1024, lane, dcat);                   // dZcat = [dZo2 | dZi2]
            }
            {
                f32x16 d1[T], h1[T];
                PIME_NO_HOIST();
                layer_mfma<H, T, 2, false>(lds + Lb.off[4], nullptr, lane, *reinterpret_cast<f32x16(*)[H]>(&dcat[0]), d1);
                PIME_NO_HOIST();
                layer_first<T, 1>(lds + Lb.off[0], xrow, Do, h, h1);             // recompute h_o1
                times_act_grad<T, 1>(d1, h1);
                stash_store<T>(st + 4 * T * 1024, lane, d1);                     // dZo1
            }
            {
                f32x16 d1[T], h1[T];
                PIME_NO_HOIST();
                layer_mfma<H, T, 2, false>(lds + Lb.off[5], nullptr, lane, *reinterpret_cast<f32x16(*)[H]>(&dcat[H]), d1);
                PIME_NO_HOIST();
                layer_first<T, 1>(lds + Lb.off[1], xrow + Do, a.Di, h, h1);      // recompute h_i1
                times_act_grad<T, 1>(d1, h1);
                stash_store<T>(st + 5 * T * 1024, lane, d1);                     // dZi1
            }
        } else {
            f32x16 d[T], hh[T], d2[T];
            head_backward<T>(lds + Lb.off[1], lane, dout, d);
            stash_load<T>(st + T * 1024, lane, hh);                              // H3
            times_act_grad<T, ACT>(d, hh);
            stash_store<T>(st + 2 * T * 1024, lane, d);                          // dZ3
            stash_load<T>(st, lane, hh);                                         // H2: in flight behind the next layer
            PIME_NO_HOIST();
            layer_mfma<T, T, 2, false>(lds + Lb.off[2], nullptr, lane, d, d2);   // dH2 = W(net.4)^T dZ3
            times_act_grad<T, ACT>(d2, hh);
            stash_store<T>(st + 3 * T * 1024, lane, d2);                         // dZ2
            PIME_NO_HOIST();
            layer_mfma<T, T, 2, false>(lds + Lb.off[3], nullptr, lane, d2, d);   // dH1 = W(net.2)^T dZ2
            PIME_NO_HOIST();
            layer_first<T, ACT>(lds + Lb.off[0], xrow, a.D, h, hh);              // recompute H1
            times_act_grad<T, ACT>(d, hh);
            stash_store<T>(st + 4 * T * 1024, lane, d);                          // dZ1
        }
    }
}

// ==================================================================================================== dW kernel
constexpr int kDwThreads = 256;
constexpr int kDwPitch = 33;                 // 32 samples + 1: conflict-free for row writes and column reads
constexpr int kDwTile = 32 * kDwPitch;
constexpr int kDwMaxDin = kMaxObsDim;        // first-layer fan-in
// LDS: operand tiles [AT + BT][32][33], then (b_kind 1) first-layer weights [32*BT][Din+1], then the tile's states [32][Din]
__host__ __device__ inline int dw_lds_floats(int Din) { return 8 * kDwTile + 128 * (Din + 1) + 32 * Din; }

// One float4 of a stashed tile per thread: elements 4*tid .. 4*tid+3 of [16 regs][64 lanes] -> feature-major LDS rows
__device__ __forceinline__ void lds_put_frag4(float* tile, int tid, const float4& v) {
    const int r = tid >> 4, ln = (tid & 15) * 4;
    float* p = tile + feat32(r, ln >> 5) * kDwPitch + (ln & 31);
    p[0] = v.x; p[1] = v.y; p[2] = v.z; p[3] = v.w;
}

template <int AT, int BT>
__device__ void dw_job(const DwJob& j, int tiles_per_wg, int dbg, float* lds) {
    constexpr int NQ = AT * BT;
    constexpr int PER_WAVE = (NQ + 3) / 4;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, li = lane & 31;
    float* A = lds;                    // [AT][32 feature rows][33]
    float* Bm = lds + AT * kDwTile;    // [BT][32 feature rows][33]
    float* W1 = lds + 8 * kDwTile;     // b_kind 1: [32*BT][Din] weights then [32*BT] bias
    float* xs = W1 + 128 * (j.Din + 1);  // [32][Din] states of the current sample tile
    const int ntiles = (j.B + 31) / 32;
    const int tile0 = blockIdx.x * tiles_per_wg;
    const int tile_end = min(tile0 + tiles_per_wg, ntiles);
    const bool head = j.a_stash == nullptr;
    if (j.b_kind == 1) {  // first-layer weights stay in LDS for the whole job
        const int nf = 32 * BT;
        for (int e = tid; e < nf * j.Din; e += kDwThreads) W1[e] = j.W[e];
        for (int e = tid; e < nf; e += kDwThreads) W1[nf * j.Din + e] = j.bias[e];
    }
    f32x16 acc[PER_WAVE];
#pragma unroll
    for (int q = 0; q < PER_WAVE; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    float dbsum = 0.f;  // thread t < 32*AT owns feature t's bias gradient

    // register prefetch of the next sample tile: global loads stay in flight behind the current tile's MFMAs
    float4 pa[AT], pb[BT];
    float px[4];
    auto prefetch = [&](int tile) {
        if (!head) {
            const float4* src = reinterpret_cast<const float4*>(j.a_stash + ((size_t)tile * j.a_nt + j.a_t0) * 1024);
#pragma unroll
            for (int t = 0; t < AT; ++t) pa[t] = src[t * 256 + tid];
        } else {
            pa[0].x = tid < 32 ? j.dout[tile * 32 + tid] : 0.f;
        }
        if (j.b_kind == 0) {
            const float4* src = reinterpret_cast<const float4*>(j.b_stash + ((size_t)tile * j.b_nt + j.b_t0) * 1024);
#pragma unroll
            for (int t = 0; t < BT; ++t) pb[t] = src[t * 256 + tid];
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int e = tid + k * kDwThreads;  // (sample, column) of the gathered state copy xg [B_pad][D]
                px[k] = e < 32 * j.Din ? j.xg[((size_t)tile * 32 + e / j.Din) * j.D + j.col0 + e % j.Din] : 0.f;
            }
        }
    };
    if (tile0 < tile_end) prefetch(tile0);
    for (int tile = tile0; tile < tile_end; ++tile) {
        __syncthreads();  // previous tile's operands fully consumed
        if (!head) {
#pragma unroll
            for (int t = 0; t < AT; ++t) lds_put_frag4(A + t * kDwTile, tid, pa[t]);
        } else {
            for (int e = tid; e < 1024; e += kDwThreads) A[(e >> 5) * kDwPitch + (e & 31)] = 0.f;
        }
        if (j.b_kind == 0) {
#pragma unroll
            for (int t = 0; t < BT; ++t) lds_put_frag4(Bm + t * kDwTile, tid, pb[t]);
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int e = tid + k * kDwThreads;
                if (e < 32 * j.Din) xs[e] = px[k];
            }
        }
        __syncthreads();
        if (head && tid < 32) A[tid] = pa[0].x;  // feature row 0 = dOut
        if (j.b_kind != 0) {  // B tile from the states: thread = (sample m, feature group)
            const int m = tid & 31;
            const float* xm = xs + m * j.Din;
            if (j.b_kind == 2) {
                for (int f = tid >> 5; f < 32; f += 8) Bm[f * kDwPitch + m] = (f < j.Din && !(dbg & 4)) ? xm[f] : 0.f;
            } else {
                const int nf = 32 * BT;
                for (int f = tid >> 5; f < nf; f += 8) {
                    float s = W1[nf * j.Din + f];
                    for (int c = 0; c < j.Din; ++c) s = fmaf(xm[c], W1[f * j.Din + c], s);
                    const float v = j.act == 0 ? (s > 0.f ? s : 0.f) : fast_tanh(s);
                    Bm[(f >> 5) * kDwTile + (f & 31) * kDwPitch + m] = (dbg & 4) ? 0.f : v;
                }
            }
        }
        if (head || j.b_kind != 0) __syncthreads();
        if (tile + 1 < tile_end) prefetch(tile + 1);
        if (j.db && tid < 32 * AT) {
            const float* rowp = A + (tid >> 5) * kDwTile + (tid & 31) * kDwPitch;
            float s = 0.f;
#pragma unroll
            for (int m = 0; m < 32; ++m) s += rowp[m];
            dbsum += s;
        }
        // ---- 16 k-steps of two samples each.  All operand reads of a half (8 k-steps) are issued before its first
        // MFMA, and the second half's reads before the first half's MFMAs, so the LDS latency hides behind matrix
        // work (the naive per-tile loop compiled to read -> lgkmcnt(0) -> 2 MFMAs, exposing it every 128 cycles).
        // kt = tq % BT is the same for every tile of a wave (4 % BT == 0), so B is read once per k-step.
        if (wave < NQ && !(dbg & 2)) {
            const int kt = wave % BT;
            const float* bp = Bm + kt * kDwTile + li * kDwPitch + h;
            float av[2][PER_WAVE][8], bv[2][8];
#pragma unroll
            for (int half = 0; half < 2; ++half) {
#pragma unroll
                for (int s = 0; s < 8; ++s) bv[half][s] = bp[2 * (8 * half + s)];
#pragma unroll
                for (int q = 0; q < PER_WAVE; ++q)
                    if (wave + 4 * q < NQ) {
                        const float* ap = A + ((wave + 4 * q) / BT) * kDwTile + li * kDwPitch + h;
#pragma unroll
                        for (int s = 0; s < 8; ++s) av[half][q][s] = ap[2 * (8 * half + s)];
                    }
            }
#pragma unroll
            for (int half = 0; half < 2; ++half)
#pragma unroll
                for (int s = 0; s < 8; ++s)
#pragma unroll
                    for (int q = 0; q < PER_WAVE; ++q)
                        if (wave + 4 * q < NQ)
                            acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[half][q][s], bv[half][s], acc[q], 0, 0, 0);
        }
    }
    // ---- combine: D[i = a feature][j = b feature], col on the lane, rows in the registers
#pragma unroll
    for (int q = 0; q < PER_WAVE; ++q) {
        const int tq = wave + 4 * q;
        if (tq < NQ) {
            const int ot = tq / BT, kt = tq % BT;
            const int col = kt * 32 + li;
            if (col < j.out_cols && !(dbg & 1)) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rowi = ot * 32 + feat32(r, h);
                    if (rowi < j.out_rows) atomicAdd(&j.dW[(size_t)rowi * j.ldw + col], acc[q][r]);
                }
            }
        }
    }
    if (j.db && tid < 32 * AT && tid < j.out_rows) atomicAdd(&j.db[tid], dbsum);
}

__global__ __launch_bounds__(kDwThreads) void ppo_dw_kernel(DwArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const DwJob& j = a.job[blockIdx.y];
    const int key = j.a_tiles * 8 + j.b_tiles;
    switch (key) {
        case 2 * 8 + 4: dw_job<2, 4>(j, a.tiles_per_wg, a.debug_skip, lds); break;
        case 4 * 8 + 1: dw_job<4, 1>(j, a.tiles_per_wg, a.debug_skip, lds); break;
        case 1 * 8 + 4: dw_job<1, 4>(j, a.tiles_per_wg, a.debug_skip, lds); break;
        case 2 * 8 + 2: dw_job<2, 2>(j, a.tiles_per_wg, a.debug_skip, lds); break;
        case 1 * 8 + 2: dw_job<1, 2>(j, a.tiles_per_wg, a.debug_skip, lds); break;
        case 2 * 8 + 1: dw_job<2, 1>(j, a.tiles_per_wg, a.debug_skip, lds); break;
        default: break;
    }
}

// ==================================================================================================== critic scale
// obj_united = obj_actor + obj_critic / (r_sum[idx].std() + 1e-5)  (agent.py:652).  The critic kernel back-propagates
// the UNSCALED SmoothL1 and accumulates sum / sum-of-squares of the minibatch targets it gathers anyway (float64
// atomics); this kernel turns the moments into torch's unbiased std and scales the critic's gradient tensors.
struct ScaleArgs {
    float* grad[8];
    int n[8];
    const double* moments;
    int B;
    float* scale_out;
    float* scale_sum;   // += scale (running sum over calls)
    int64_t* index_row; // NULL, or the index table's row cursor (advanced here, after every reader)
};

__global__ void critic_scale_kernel(ScaleArgs a) {
    const double s = a.moments[0], ss = a.moments[1], B = (double)a.B;
    const double var = a.B > 1 ? fmax((ss - s * s / B) / (B - 1.0), 0.0) : 0.0;
    const float scale = (float)(1.0 / ((double)(float)sqrt(var) + 1e-5));
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
        a.scale_out[0] = scale; a.scale_sum[0] += scale;
        const float csum = a.scale_sum[-1];   // loss_sums[2]; see ppo_grad_reduce_kernel
        a.scale_sum[1] += (csum - a.scale_sum[2]) * scale;
        a.scale_sum[2] = csum;
        if (a.index_row) a.index_row[0] += 1;
    }
    float* g = a.grad[blockIdx.y];
    const int n = a.n[blockIdx.y];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) g[i] *= scale;
}

int launch_critic_scale(int D, int md, float* const* grads, const double* moments, int B, float* scale_out,
                        float* scale_sum, int64_t* index_row, hipStream_t s) {
    ScaleArgs a{};
    const int sizes[8] = {md * D, md, md * md, md, md * md, md, md, 1};
    for (int i = 0; i < 8; ++i) { a.grad[i] = grads[i]; a.n[i] = sizes[i]; }
    a.moments = moments; a.B = B; a.scale_out = scale_out; a.scale_sum = scale_sum; a.index_row = index_row;
    hipLaunchKernelGGL(critic_scale_kernel, dim3(16, 8), dim3(256), 0, s, a);
    PIME_HIP_TRY(hipGetLastError());
    return PIME_OK;
}

// ==================================================================================================== Adam
// torch.optim.Adam (no weight decay, no amsgrad; agent.py:565-566,656-657) over ONE flat parameter / gradient
// tensor.  torch's fused multi-tensor Adam needs 27 us for the 22 small tensors of the two nets and 94 us when
// handed the flat 67k-element tensor (one 64k chunk = one workgroup); this is a plain grid-wide elementwise pass.
// One launch: every thread reads the OLD step count t0 and updates with t = t0 + 1; the workgroup that finishes last
// (a device-side arrival counter) stores t.  No workgroup can read step[] after that store: the store waits for all of
// them to have arrived, and they arrive after their reads.  (A separate 1-thread "tick" launch cost ~3 us per step.)
// The arrival counter is step[1] (caller-owned, one per optimizer), so optimizers on different streams do not share it.
struct AdamImages {   // optional: keep the packed images current (pime_adam_step_images)
    const int32_t* map;   // [2 n], see pime_ppo_image_map; NULL: parameters only
    float* img[2][2];     // [0 critic, 1 actor][0 forward, 1 transposed]
};
// Data parallel (pime_adam_step_dp): the critic's gradient arrives unscaled and rank-averaged, with the averaged target moments
// behind it; every workgroup recovers 1 / (std of the union minibatch + 1e-5) from them and applies it to the critic's elements.
struct AdamDp {
    const float* moments;   // NULL: plain step.  [0] mean over ranks of sum r, [1] of sum r^2, [2] B
    long long critic_off;
    int world;
};
__global__ void adam_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, long long n, float lr, float b1, float b2, float eps,
                            float* __restrict__ step, AdamImages im, AdamDp dp) {
    // the bias corrections (two float64 pow behind a dependent load of the step count) once per workgroup, not once per thread:
    // they are ~10x the work of an element's update
    __shared__ float consts[4];
    if (threadIdx.x == 0) {
        const float tn = step[0] + 1.0f;
        const double t = (double)tn;
        consts[0] = tn;
        consts[1] = lr / (float)(1.0 - pow((double)b1, t));
        consts[2] = (float)sqrt(1.0 - pow((double)b2, t));
        consts[3] = 1.0f;
        if (dp.moments) {   // torch's unbiased std over the union of the ranks' minibatches (ppo_grad_reduce_kernel's formula)
            const double G = (double)dp.world, N = G * (double)dp.moments[2];
            const double m1 = G * (double)dp.moments[0], m2 = G * (double)dp.moments[1];
            const double var = N > 1.0 ? fmax((m2 - m1 * m1 / N) / (N - 1.0), 0.0) : 0.0;
            consts[3] = (float)(1.0 / ((double)(float)sqrt(var) + 1e-5));
        }
    }
    __syncthreads();
    const float t_new = consts[0], step_size = consts[1], bc2_sqrt = consts[2], cscale = consts[3];
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        float gi = g[i];
        if (dp.moments && i >= dp.critic_off) {
            gi *= cscale;
            g[i] = gi;   // .grad holds the gradient of the united loss
        }
        const float mi = m[i] + (gi - m[i]) * (1.0f - b1);       // exp_avg.lerp_(grad, 1 - beta1)
        const float vi = v[i] * b2 + gi * gi * (1.0f - b2);      // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
        m[i] = mi;
        v[i] = vi;
        const float pn = p[i] - step_size * (mi / (sqrtf(vi) / bc2_sqrt + eps));
        p[i] = pn;
        if (im.map) {
            const int2 e = reinterpret_cast<const int2*>(im.map)[i];
            if (e.x >= 0) im.img[(e.x >> 28) & 1][0][e.x & 0x0fffffff] = pn;
            if (e.y >= 0) im.img[(e.y >> 28) & 1][1][e.y & 0x0fffffff] = pn;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned int* arrivals = reinterpret_cast<unsigned int*>(step + 1);
        const unsigned int prev = atomicAdd(arrivals, 1u);
        if (prev == gridDim.x - 1) {
            *arrivals = 0;           // ready for the next launch (launches of one optimizer are stream-ordered)
            step[0] = t_new;
        }
    }
}

int launch_adam(float* p, float* g, float* m, float* v, long long n, float lr, float b1, float b2, float eps,
                float* step, const int32_t* map, float* const (*img)[2], const float* dp_moments, long long critic_off, int world,
                hipStream_t s) {
    AdamImages im{};
    if (map) {
        im.map = map;
        for (int k = 0; k < 2; ++k)
            for (int w = 0; w < 2; ++w) im.img[k][w] = img[k][w];
    }
    const int block = 256;
    long long grid = (n + block - 1) / block;
    if (grid > 2048) grid = 2048;
    const AdamDp dp{dp_moments, critic_off, world};
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)grid), dim3(block), 0, s, p, g, m, v, n, lr, b1, b2, eps, step, im, dp);
    PIME_HIP_TRY(hipGetLastError());
    return PIME_OK;
}

// ==================================================================================================== host side
int mlp_check(int kind, int D, int Di, int md);

bool family16(int kind, int md);
bool family16_grad(int kind, int md, int D, int Di);
int64_t packed16_floats(int kind, int D, int Di, int md);
int64_t bwd16_floats(int kind, int md);
bool b3_grad(int kind, int md, int D, int Di);
int64_t b3_floats(int kind, int md);
int grid16(int kind, int B, int md, int D, int Di);
int launch_pack16(const PackArgs& a, float* fwd, float* bwd, hipStream_t s);

// the transposed image's f32 part (what the image map covers) and the whole image: + the bf16x3 planes under PIME_GRAD_BF16X3=1
int64_t ppo_bwd_image_f32_floats(int kind, int D, int Di, int md) {
    return family16_grad(kind, md, D, Di) ? bwd16_floats(kind, md) : (int64_t)bwd_layout(kind, D, Di, md).total;
}
int64_t ppo_bwd_image_floats(int kind, int D, int Di, int md) {
    return ppo_bwd_image_f32_floats(kind, D, Di, md) + (b3_grad(kind, md, D, Di) ? b3_floats(kind, md) : 0);
}
int64_t ppo_fwd_image_floats(int kind, int D, int Di, int md) {
    return family16_grad(kind, md, D, Di) ? packed16_floats(kind, D, Di, md) : (int64_t)mlp_layout(kind, D, Di, md).total;
}

int64_t fused_workspace_floats(int kind, int B, int D, int Di, int md);

int64_t ppo_workspace_floats(int kind, int B, int md) {
    int64_t f16 = 0;
    int poff[13], psize[12];
    if (kind != MLP_MODULAR_ACTOR) {   // the 16-tile family: gradient slabs only (no activation stash), bound over the state widths
        const int64_t stride = slab_layout16(kMaxObsDim, md, poff, psize);
        int g = grid16(kind, B, md, 1, 0);
        const int g2 = grid16(kind, B, md, kMaxObsDim, 0);
        g = g > g2 ? g : g2;
        f16 = (int64_t)g * stride;
    } else if (md == 256 || md == 128) {
        f16 = (int64_t)grid16(kind, B, md, 4, 1) * slab_layout16m(md, poff, psize);   // (the grid does not depend on the state width)
    }
    if (md == 256) return f16;
    const int64_t ntiles = ((B + 31) / 32 + 7) / 8 * 8;
    const int64_t split = ntiles * stash_tiles(kind, md / 32) * 1024 + ntiles * 32 + ntiles * 32 * kMaxObsDim;
    const int64_t fused = fused_workspace_floats(kind, B, kMaxObsDim, 1, md);  // slab size bound: widest state
    const int64_t old = split > fused ? split : fused;
    return old > f16 ? old : f16;   // the caller does not say the state width: room for whichever family serves it
}

int launch_pack_bwd(int kind, int D, int Di, int md, const float* const* params, float* out, hipStream_t s) {
    if (int rc = mlp_check(kind, D, Di, md)) return rc;
    PackBwdArgs a{};
    const int np = kind == MLP_MODULAR_ACTOR ? 12 : 8;
    for (int i = 0; i < np; ++i) {
        PIME_REQUIRE(params[i] != nullptr, "ppo pack: params[%d] is NULL", i);
        a.p[i] = params[i];
    }
    a.kind = kind; a.D = D; a.Di = Di; a.md = md;
    if (family16_grad(kind, md, D, Di)) return launch_pack16(a, nullptr, out, s);
    hipLaunchKernelGGL(mlp_pack_bwd_kernel, dim3(64), dim3(256), 0, s, a, out);
    PIME_HIP_TRY(hipGetLastError());
    return PIME_OK;
}

template <int T, int KIND>
static int launch_net(const PpoArgs& a, hipStream_t s) {
    const MlpLayout L = mlp_layout(KIND, a.D, a.Di, T * 32);
    const BwdLayout Lb = bwd_layout(KIND, a.D, a.Di, T * 32);
    const size_t lds_bytes = sizeof(float) * (size_t)(L.total > Lb.total ? L.total : Lb.total);
    PIME_REQUIRE(lds_bytes <= 160 * 1024, "PPO net image (%zu B) exceeds the 160 KB LDS", lds_bytes);
    static LdsLimit lds_limit;  // per instantiation
    PIME_RAISE_LDS(lds_limit, (ppo_fwd_bwd_kernel<T, KIND>), 160 * 1024);
    const int ntiles = (a.B + 31) / 32, waves = kTrainThreads / 64;
    int grid = (ntiles + waves - 1) / waves;
    if (grid > 256) grid = 256;
    hipLaunchKernelGGL((ppo_fwd_bwd_kernel<T, KIND>), dim3(grid), dim3(kTrainThreads), lds_bytes, s, a);
    PIME_HIP_TRY(hipGetLastError());
    return PIME_OK;
}

int launch_ppo_net(int kind, int md, const PpoArgs& a, hipStream_t s) {
    const int T = md / 32;
#define PIME_NET(TT, KK) \
    if (T == TT && kind == KK) return launch_net<TT, KK>(a, s);
    PIME_NET(2, MLP_CRITIC) PIME_NET(4, MLP_CRITIC)
    PIME_NET(2, MLP_PLAIN_ACTOR) PIME_NET(4, MLP_PLAIN_ACTOR)
    PIME_NET(2, MLP_MODULAR_ACTOR) PIME_NET(4, MLP_MODULAR_ACTOR)
#undef PIME_NET
    set_error("no fused PPO instantiation for kind %d width %d", kind, md);
    return PIME_ERR_ARG;
}

// Fills the dW jobs of one net.  params/grads: nn.Linear order as in pime_mlp_pack (W,b pairs).
int build_dw_jobs(int kind, int md, const PpoArgs& a, const float* const* params, float* const* grads, DwJob* jobs) {
    const int T = md / 32, NT = stash_tiles(kind, T);
    int n = 0;
    auto base = [&]() {
        DwJob j{};
        j.xg = a.xg; j.D = a.D; j.B = a.B; j.dout = a.dout;
        return j;
    };
    auto stash_a = [&](DwJob& j, int t0, int tiles) { j.a_stash = a.stash; j.a_nt = NT; j.a_t0 = t0; j.a_tiles = tiles; };
    auto stash_b = [&](DwJob& j, int t0, int tiles) { j.b_kind = 0; j.b_stash = a.stash; j.b_nt = NT; j.b_t0 = t0; j.b_tiles = tiles; };
    auto outp = [&](DwJob& j, int pi, int rows, int cols) {
        j.dW = grads[pi]; j.db = grads[pi + 1]; j.ldw = cols; j.out_rows = rows; j.out_cols = cols;
    };
    if (kind == MLP_MODULAR_ACTOR) {
        const int Do = a.D - a.Di, H = T / 2;
        { DwJob j = base(); j.a_stash = nullptr; j.a_tiles = 1; stash_b(j, T, T); outp(j, 10, 1, md); jobs[n++] = j; }            // net.2
        { DwJob j = base(); stash_a(j, 2 * T, T); stash_b(j, 0, T); outp(j, 8, md, md); jobs[n++] = j; }                            // net.0
        { DwJob j = base(); stash_a(j, 3 * T, H); j.b_kind = 1; j.b_tiles = T; j.W = params[0]; j.bias = params[1]; j.Din = Do;
          j.col0 = 0; j.act = 1; outp(j, 2, md / 2, md); jobs[n++] = j; }                                                           // other_net.2
        { DwJob j = base(); stash_a(j, 3 * T + H, H); j.b_kind = 1; j.b_tiles = T; j.W = params[4]; j.bias = params[5];
          j.Din = a.Di; j.col0 = Do; j.act = 1; outp(j, 6, md / 2, md); jobs[n++] = j; }                                            // integrator_net.2
        { DwJob j = base(); stash_a(j, 4 * T, T); j.b_kind = 2; j.b_tiles = 1; j.Din = Do; j.col0 = 0; outp(j, 0, md, Do); jobs[n++] = j; }   // other_net.0
        { DwJob j = base(); stash_a(j, 5 * T, T); j.b_kind = 2; j.b_tiles = 1; j.Din = a.Di; j.col0 = Do; outp(j, 4, md, a.Di); jobs[n++] = j; } // integrator_net.0
    } else {
        const int act = kind == MLP_CRITIC ? 0 : 1;
        { DwJob j = base(); j.a_stash = nullptr; j.a_tiles = 1; stash_b(j, T, T); outp(j, 6, 1, md); jobs[n++] = j; }               // net.6
        { DwJob j = base(); stash_a(j, 2 * T, T); stash_b(j, 0, T); outp(j, 4, md, md); jobs[n++] = j; }                            // net.4
        { DwJob j = base(); stash_a(j, 3 * T, T); j.b_kind = 1; j.b_tiles = T; j.W = params[0]; j.bias = params[1]; j.Din = a.D;
          j.col0 = 0; j.act = act; outp(j, 2, md, md); jobs[n++] = j; }                                                             // net.2
        { DwJob j = base(); stash_a(j, 4 * T, T); j.b_kind = 2; j.b_tiles = 1; j.Din = a.D; j.col0 = 0; outp(j, 0, md, a.D); jobs[n++] = j; }  // net.0
    }
    // A 128x128 gradient as one (4 x 4)-tile job needs 64 accumulator + 32 prefetch + 80 operand registers per lane,
    // which caps the kernel at two workgroups per CU and leaves it latency-bound (PMC: waves parked on memory 54 %,
    // matrix pipe busy 33 %).  Splitting it into two (2 x 4) jobs over the output rows re-reads the B tiles once more
    // but halves the registers, so twice as many workgroups hide each other's HBM latency.
    int m = n;
    for (int i = 0; i < n; ++i) {
        if (jobs[i].a_tiles == 4 && jobs[i].b_tiles == 4) {
            DwJob hi = jobs[i];
            jobs[i].a_tiles = 2; jobs[i].out_rows = 64;
            hi.a_tiles = 2; hi.a_t0 += 2; hi.out_rows = 64;
            hi.dW += (size_t)64 * hi.ldw; hi.db += 64;
            jobs[m++] = hi;
        }
    }
    return m;
}

int launch_dw(const DwArgs& args, int B, hipStream_t s) {
    static LdsLimit lds_limit;  // per instantiation
    int din_max = 1;
    for (int i = 0; i < args.njobs; ++i) din_max = args.job[i].Din > din_max ? args.job[i].Din : din_max;
    const size_t lds_bytes = sizeof(float) * (size_t)dw_lds_floats(din_max);
    PIME_RAISE_LDS(lds_limit, ppo_dw_kernel, (int)(sizeof(float) * dw_lds_floats(kDwMaxDin)));
    const int ntiles = (B + 31) / 32;
    const int chunks = (ntiles + args.tiles_per_wg - 1) / args.tiles_per_wg;
    hipLaunchKernelGGL(ppo_dw_kernel, dim3(chunks, args.njobs), dim3(kDwThreads), lds_bytes, s, args);
    PIME_HIP_TRY(hipGetLastError());
    return PIME_OK;
}

}  // namespace pime
