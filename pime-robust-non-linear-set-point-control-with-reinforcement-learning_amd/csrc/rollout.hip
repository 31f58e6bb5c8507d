// Fused on-device rollout for the pH and water-tank (Integrator and Stacking1/4/10 observation) envs: ONE launch advances every lane through `n_steps` steps --
// policy forward on the f32 matrix cores, exploration noise, residual action composition, env step (+ in-kernel
// auto-reset) and the trajectory-buffer writes -- with the env state and the observation in registers for the whole
// episode.  Replaces the per-step launch sequence of AgentResidual*.explore_env
// (/root/reference/elegantrl/agent_residual.py:52-69: select_action -> env.step(tanh(a)+s@priorK) -> append_buffer),
// i.e. ~8 launches x 50 steps per rollout.
//
// Tilings (template parameter NARROW; chosen per launch by tiling() below).  The packed image fills LDS, so a compute unit holds ONE
// workgroup, and every wave is latency-bound on its own serial MFMA chain plus the env arithmetic of its lanes:
//   0  one 32-lane tile per wave, two waves per workgroup (rounds 1-2): v_mfma_f32_32x32x2_f32, 512 x 64 cycles per env step; both
//      lane halves carry the same 32 envs (the MFMA layout needs the sample on lane & 31).  Leaves two SIMDs of every unit idle.
//   1  one 16-lane tile per wave, four waves per workgroup: v_mfma_f32_16x16x4_f32 on the same image (policy_forward16), 512 x 32
//      cycles; the four lane groups carry the same 16 envs.  The default above 4 096 lanes.
//   2  one 16-lane tile per WORKGROUP, each wave a quarter of every layer's output tiles (policy_forward16q), 128 x 32 cycles + two
//      LDS exchanges; the default up to 4 096 lanes (at most one tile per compute unit).  Bit-identical to tiling 1.
// The redundant copies of an env do the same arithmetic in every lane group / wave (cheap enough not to shuffle); one of them stores.
// The env arithmetic is env_device.hpp -- literally the code of ph_step_kernel / ph_reset_kernel.
#include <cstdlib>
#include "env_device.hpp"
#include "rollout.hpp"
#include "rollout_policy.hpp"

namespace pime {

constexpr uint32_t STREAM_EXPLORE = 2;

constexpr int kRolloutThreads = 128;
constexpr int kRolloutNarrowThreads = 256;

__device__ __forceinline__ PhPtrs<float, half_t> with_half_I(const PhPtrs<float>& s, half_t* I16) {
    PhPtrs<float, half_t> h{};
    h.x = s.x; h.A = s.A; h.B = s.B; h.C = s.C; h.qww = s.qww; h.qc = s.qc; h.r = s.r; h.last_a = s.last_a; h.I = I16;
    h.t = s.t; h.episode = s.episode; h.table = s.table;
    return h;
}
__device__ __forceinline__ WtPtrs<float, half_t> with_half_I(const WtPtrs<float>& s, half_t* I16) {
    WtPtrs<float, half_t> h{};
    h.h1 = s.h1; h.h2 = s.h2; h.r = s.r; h.a1 = s.a1; h.a2 = s.a2; h.kp = s.kp; h.I = I16; h.frames = s.frames; h.head = s.head;
    h.t = s.t; h.episode = s.episode;
    return h;
}
__device__ __forceinline__ float through_half(float v) { return (float)(half_t)v; }

// ENV 0: pH, obs [y, r, I];  1: water tank, Integrator obs [h1, h2, r, I];  2: water tank, Stacking obs = the last STACK frames
// [h1, h2, r], oldest first (nonlinear_watertank.py:1056-1208).  For ENV 2 the observation registers ARE the frame deque: a step
// shifts them by one frame and appends the new one, a reset fills every frame with the first (:1181-1183); the SoA ring in HBM
// is only written back when the launch ends.
// NARROW (the default): 16 lanes per wave (four workgroup waves, policy_forward16) -- every SIMD has a wave where the 32-lane tiles of a launch
// would cover only half of them, and a layer's serial MFMA chain is half as long.  Same image, same env code, same Philox keys (the
// noise of a lane does not depend on the tiling); the policy mean differs in the last bits (another summation order).
// QUAD (NARROW == 2; launches of <= 4 096 lanes): the four waves of a workgroup share ONE 16-lane tile and a quarter of every layer's
// output tiles each (policy_forward16q: a quarter of the MFMA chain, two LDS exchanges per env step); the same MFMA sequence per
// accumulator as NARROW, so the trajectories are bit-identical to it.
template <int T, int KIND, int ENV, int STACK, int NARROW = 0>
__global__ __launch_bounds__(NARROW ? kRolloutNarrowThreads : kRolloutThreads) void rollout_kernel(RolloutArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int D = ENV == 0 ? 3 : (ENV == 1 ? 4 : 3 * STACK), Di = ENV == 2 ? 0 : 1;
    static_assert(ENV != 2 || KIND == MLP_PLAIN_ACTOR, "the Stacking observation has no integrator column: plain actors only");
    const MlpLayout L = mlp_layout(KIND, D, Di, T * 32);
    stage_image(lds, a.img, L.total / 4);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5;
    const int N = a.n;
    constexpr int LW = NARROW ? 16 : 32, WAVES = (NARROW ? kRolloutNarrowThreads : kRolloutThreads) / 64;   // env lanes per wave
    constexpr bool QUAD = NARROW == 2;
    const int m = (QUAD ? blockIdx.x : blockIdx.x * WAVES + wave) * LW + (lane & (LW - 1));
    const bool valid = m < N;
    const int i = valid ? m : N - 1;  // idle lanes shadow the last env (compute, never store)
    // the lane groups (and, QUAD, the waves) carry copies of the same envs: one of them stores
    const bool writer = valid && (NARROW ? (lane >> 4) == 0 && (!QUAD || wave == 0) : h == 0);
    [[maybe_unused]] float* const xbuf = lds + L.total;   // QUAD: the activation exchange buffers behind the image
    const uint32_t gid = a.env_offset + (uint32_t)i;
    const float sigma = __expf(a.a_std_log[0]);

    PhLane<float> E;   // exactly one of the two lanes is live, selected at compile time
    WtLane<float> W;
    // binary16 storage (PIME_STATE_MIXED16, wave-uniform): I lives as binary16 in the handle, observation / reward rows are
    // binary16, and everything a step-per-launch *_h kernel would have stored and re-read is rounded through binary16 here
    const bool h16 = ENV != 2 && a.I16 != nullptr;
    if constexpr (ENV == 0) {
        if (h16) ph_lane_load<float>(a.p, with_half_I(a.st, a.I16), i, E);
        else ph_lane_load<float>(a.p, a.st, i, E);
    } else {
        if (h16) wt_lane_load<float>(a.wp, with_half_I(a.wst, a.I16), i, W);
        else wt_lane_load<float>(a.wp, a.wst, i, W);
    }
    float obs[D];
#pragma unroll
    for (int j = 0; j < D; ++j) obs[j] = h16 ? (float)a.state_h[(size_t)D * i + j] : a.state[(size_t)D * i + j];
    for (int t = 0; t < a.n_steps; ++t) {
        PIME_NO_HOIST();
        float a_avg;
        if constexpr (QUAD) a_avg = policy_forward16q<T, KIND, D, Di>(lds, xbuf, L, obs, lane, wave);
        else if constexpr (NARROW == 1) a_avg = policy_forward16<T, KIND, D, Di>(lds, L, obs, lane);
        else a_avg = policy_forward<T, KIND, D, Di>(lds, L, obs, lane);
        // exploration noise eps ~ N(0,1): Box-Muller on a Philox pair keyed by the global lane (net_residual.py:178)
        double ua, ub;
        philox_pair(a.noise_seed, gid, a.noise_epoch, (uint32_t)t, STREAM_EXPLORE, ua, ub);
        const float eps = (float)(sqrt(-2.0 * log(1.0 - ua)) * cos(6.283185307179586476925286766559 * ub));
        const float a_pre = a_avg + eps * sigma;                                   // net_residual.py:179
        double dot = 0.0;                                                          // agent_residual.py:61
#pragma unroll
        for (int j = 0; j < D; ++j) dot += (double)obs[j] * a.K.k[j];
        const double a_env = residual_tanh(a_pre) + dot;
        float nxt[D], rew;
        bool d;
        if constexpr (ENV == 0) {
            float o3[3];
            d = ph_lane_step<float>(a.p, a.st.table, a_env, E, o3, rew);
            if (d) ph_lane_reset<float>(a.p, a.st.table, gid, nullptr, E, o3);     // in-kernel auto-reset
            nxt[0] = o3[0]; nxt[1] = o3[1]; nxt[2] = o3[2];
        } else {
            double z1n, z2n;
            wt_lane_noise<float>(a.wp, gid, W, nullptr, z1n, z2n);
            d = wt_lane_step<float>(a.wp, a_env, z1n, z2n, W, rew);
            if (d) wt_lane_reset<float>(a.wp, gid, nullptr, W);
            if constexpr (ENV == 1) {
                nxt[0] = W.h1; nxt[1] = W.h2; nxt[2] = W.r; nxt[D - 1] = W.I;
            } else {   // deque(maxlen=S).append (:1143-1144), or after a reset every frame = the first one (:1181-1183)
#pragma unroll
                for (int j = 0; j < D - 3; ++j) nxt[j] = d ? (j % 3 == 0 ? W.h1 : (j % 3 == 1 ? W.h2 : W.r)) : obs[j + 3];
                nxt[D - 3] = W.h1; nxt[D - 2] = W.h2; nxt[D - 1] = W.r;
            }
        }
        const size_t k = (size_t)t * N + i;
        if (h16) {   // what the next step reads back is what binary16 storage kept
            if constexpr (ENV == 0) E.I = through_half(E.I);
            else W.I = through_half(W.I);
#pragma unroll
            for (int j = 0; j < D; ++j) nxt[j] = through_half(nxt[j]);
        }
        if (writer) {
            a.action[k] = a_pre;
            a.noise[k] = eps;
            a.done[k] = (uint8_t)d;
            if (h16) {
                a.reward_h[k] = (half_t)rew;
                half_t* s = a.state_h + ((size_t)(t + 1) * N + i) * D;
#pragma unroll
                for (int j = 0; j < D; ++j) s[j] = (half_t)nxt[j];
            } else {
                a.reward[k] = rew;
                float* s = a.state + ((size_t)(t + 1) * N + i) * D;
#pragma unroll
                for (int j = 0; j < D; ++j) s[j] = nxt[j];
            }
        }
#pragma unroll
        for (int j = 0; j < D; ++j) obs[j] = nxt[j];
    }
    if (writer) {
        if constexpr (ENV == 0) {
            if (h16) ph_lane_store<float>(a.p, with_half_I(a.st, a.I16), i, E);
            else ph_lane_store<float>(a.p, a.st, i, E);
        } else {
            if (h16) wt_lane_store<float>(a.wp, with_half_I(a.wst, a.I16), i, W);
            else wt_lane_store<float>(a.wp, a.wst, i, W);
        }
        if constexpr (ENV == 2) {   // the frame ring of the step-per-launch kernels: slot j = frame j, oldest at slot 0
#pragma unroll
            for (int j = 0; j < D; ++j) a.wst.frames[(size_t)j * N + i] = obs[j];
            a.wst.head[i] = 0;
        }
    }
}

int mlp_check(int kind, int D, int Di, int md);
int launch_rollout16(int kind, const RolloutArgs& a, hipStream_t s);   // width 256: the streamed 16-tile family (mlp16.hip)

// Tiling of a launch.  Default: 16-lane tiles (NARROW) -- the image fills LDS, so a compute unit holds ONE workgroup; four waves of 16
// lanes give each of its SIMDs a wave with half the serial chain of a 32-lane tile, where the 32-lane workgroup (two waves) leaves
// two SIMDs idle at any launch size.  Launches of <= 4 096 lanes (256 tiles: at most one per compute unit) split every tile over the
// four waves of its workgroup (QUAD) when the exchange buffers fit behind the image.  NARROW and QUAD give bit-identical
// trajectories, so a lane's results do not depend on how the lanes are sharded over ranks (tests/test_gpu_config4.py).
// PIME_ROLLOUT_NARROW=0 / 1 / 2 forces 32-lane tiles / NARROW / QUAD (A/B; the tests replay all of them).
static int tiling(int n, size_t quad_lds_bytes) {
    int mode = n <= 4096 ? 2 : 1;
    if (const char* e = std::getenv("PIME_ROLLOUT_NARROW")) mode = std::atoi(e);   // read per launch: tests flip it
    if (mode == 2 && quad_lds_bytes > 160 * 1024) mode = 1;                        // (the 30-float Stacking10 observation at width 128)
    return mode < 0 || mode > 2 ? 1 : mode;
}

template <int T, int KIND, int ENV, int STACK, int NARROW>
static int launch_rollout_n(const RolloutArgs& a, hipStream_t s) {
    const MlpLayout L = mlp_layout(KIND, ENV == 0 ? 3 : (ENV == 1 ? 4 : 3 * STACK), ENV == 2 ? 0 : 1, T * 32);
    const size_t lds_bytes = ((size_t)L.total + (NARROW == 2 ? quad_xchg_floats<T>() : 0)) * sizeof(float);
    static LdsLimit lds_limit;  // per instantiation
    PIME_RAISE_LDS(lds_limit, (rollout_kernel<T, KIND, ENV, STACK, NARROW>), 160 * 1024);
    constexpr int threads = NARROW ? kRolloutNarrowThreads : kRolloutThreads;
    constexpr int per_wg = NARROW == 2 ? 16 : threads / 64 * (NARROW ? 16 : 32);
    hipLaunchKernelGGL((rollout_kernel<T, KIND, ENV, STACK, NARROW>), dim3((a.n + per_wg - 1) / per_wg), dim3(threads), lds_bytes, s, a);
    PIME_HIP_TRY(hipGetLastError());
    return PIME_OK;
}
template <int T, int KIND, int ENV, int STACK>
static int launch_rollout_t(const RolloutArgs& a, hipStream_t s) {
    const MlpLayout L = mlp_layout(KIND, ENV == 0 ? 3 : (ENV == 1 ? 4 : 3 * STACK), ENV == 2 ? 0 : 1, T * 32);
    const int mode = tiling(a.n, ((size_t)L.total + quad_xchg_floats<T>()) * sizeof(float));
    if (mode == 2) return launch_rollout_n<T, KIND, ENV, STACK, 2>(a, s);
    if (mode == 1) return launch_rollout_n<T, KIND, ENV, STACK, 1>(a, s);
    return launch_rollout_n<T, KIND, ENV, STACK, 0>(a, s);
}

int launch_rollout(int kind, int md, const RolloutArgs& a, hipStream_t s) {
    const int D = a.env == 0 ? 3 : (a.env == 1 ? 4 : 3 * a.wp.num_stack);
    if (int rc = mlp_check(kind, D, kind == MLP_MODULAR_ACTOR ? 1 : 0, md)) return rc;
    PIME_REQUIRE(kind != MLP_CRITIC, "rollout needs an actor image");
    if (md == 256) return launch_rollout16(kind, a, s);
    const int T = md / 32;
#define PIME_RS(TT, SS) \
    if (T == TT && kind == MLP_PLAIN_ACTOR && a.env == 2 && a.wp.num_stack == SS) return launch_rollout_t<TT, MLP_PLAIN_ACTOR, 2, SS>(a, s);
    PIME_RS(4, 1) PIME_RS(4, 4) PIME_RS(4, 10) PIME_RS(2, 1) PIME_RS(2, 4) PIME_RS(2, 10)
#undef PIME_RS
#define PIME_RO(TT, KK, EE) \
    if (T == TT && kind == KK && a.env == EE) return launch_rollout_t<TT, KK, EE, 0>(a, s);
    PIME_RO(4, MLP_MODULAR_ACTOR, 0) PIME_RO(2, MLP_MODULAR_ACTOR, 0) PIME_RO(4, MLP_PLAIN_ACTOR, 0) PIME_RO(2, MLP_PLAIN_ACTOR, 0)
    PIME_RO(4, MLP_MODULAR_ACTOR, 1) PIME_RO(2, MLP_MODULAR_ACTOR, 1) PIME_RO(4, MLP_PLAIN_ACTOR, 1) PIME_RO(2, MLP_PLAIN_ACTOR, 1)
#undef PIME_RO
    set_error("no fused rollout instantiation for env %d kind %d width %d", a.env, kind, md);
    return PIME_ERR_ARG;
}

}  // namespace pime
