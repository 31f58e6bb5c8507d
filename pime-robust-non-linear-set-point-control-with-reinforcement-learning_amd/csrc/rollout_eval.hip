// Fused on-device EVALUATION: one launch advances every lane through `n_steps` steps under the DETERMINISTIC residual policy
// (a_env = tanh(mean(s)) + s @ priorK, no exploration noise, no auto-reset), optionally along a set-point schedule, with the
// env state and the observation in registers, and leaves per-lane returns (and, if asked, a float64 trace of every step).
//
// replaces, per launch:
//   * get_episode_return on every lane (/root/reference/elegantrl/run.py:600-619: reset -> max_step x [act(s) -> env.step]
//     -> sum of rewards), i.e. the evaluator's one-policy-forward + one-env-step launch pair per step with a host loop around it;
//   * the fixed set-point step-response protocols (utils/test.py:1369-1407 pH: r = 10,6,3,8,5 x 50 steps, plant state carried
//     over; :209-349 water tank: r = 3,6,9,4,2; utils/robust_test.py:4-46): a segment boundary does what the protocol's
//     `env.reset(); set_state(last); set_r(r)` leaves behind -- integrated error 0, step counter 0, new set-point, plant state
//     kept -- and the trace holds what the protocol appends per step.
// KIND = -1: the prior controller alone (get_linear_action, ph.py:227-231 / nonlinear_watertank.py:755-759 without its clip).
// S = float (PIME_STATE_MIXED) or double (PIME_STATE_F64: the golden-pinned protocol tests run here, 1e-11).
// The policy forward is rollout_policy.hpp -- the code of the rollout kernel; the env arithmetic is env_device.hpp.
#include <cstdlib>
#include "env_device.hpp"
#include "rollout_eval.hpp"
#include "rollout_policy.hpp"

namespace pime {

constexpr int kEvalThreads = 256;   // four waves of 16 lanes: one per SIMD (16-lane tiles, rollout_policy.hpp: policy_forward16)

// QUAD (launches of <= 4 096 lanes with a policy): one 16-lane tile per workgroup, split over its four waves (policy_forward16q)
template <int T, int KIND, int ENV, typename S, bool QUAD>
__global__ __launch_bounds__(kEvalThreads) void rollout_eval_kernel(EvalArgs<S> a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int D = ENV == 0 ? 3 : 4;
    constexpr bool POLICY = KIND >= 0;
    MlpLayout L{};
    if constexpr (POLICY) {
        L = mlp_layout(KIND, D, 1, T * 32);
        stage_image(lds, a.img, L.total / 4);
        __syncthreads();
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int N = a.n;
    const int m = (QUAD ? blockIdx.x : blockIdx.x * (kEvalThreads / 64) + wave) * 16 + (lane & 15);
    const bool valid = m < N;
    const int i = valid ? m : N - 1;  // idle lanes shadow the last env (compute, never store)
    const bool writer = valid && (lane >> 4) == 0 && (!QUAD || wave == 0);   // the lane groups (QUAD: and the waves) carry copies of the same 16 envs
    const uint32_t gid = a.env_offset + (uint32_t)i;

    PhLane<S> E{};
    WtLane<S> W{};
    float obs[D];
    if constexpr (ENV == 0) {
        ph_lane_load<S>(a.p, a.st, i, E);
        obs[0] = (float)ph_lookup<S>(a.p, a.st.table, E.C, E.x); obs[1] = (float)E.r; obs[2] = (float)E.I;
    } else {
        wt_lane_load<S>(a.wp, a.wst, i, W);
        obs[0] = (float)W.h1; obs[1] = (float)W.h2; obs[2] = (float)W.r; obs[3] = (float)W.I;
    }
    double ret = 0.0;
    for (int t = 0; t < a.n_steps; ++t) {
        PIME_NO_HOIST();
        if (a.seg_len > 0 && t % a.seg_len == 0) {   // segment boundary of a step-response protocol (wave-uniform)
            const double sp = a.setpoint[t / a.seg_len];
            // a boundary stands for the env.reset() the step-per-launch protocol does per segment (protocols.py): beyond the first
            // one it starts a new EPISODE -- the process noise is keyed on (episode, t), so without the bump every segment would
            // replay segment 0's noise sequence
            const int bump = t > 0 ? 1 : 0;
            if constexpr (ENV == 0) { E.r = (S)sp; E.I = S(0); E.t = 0; E.episode += bump; obs[1] = (float)E.r; obs[2] = 0.f; }
            else { W.r = (S)sp; W.I = S(0); W.t = 0; W.episode += bump; obs[2] = (float)W.r; obs[3] = 0.f; }
        }
        double a_env = 0.0;                                                        // agent_residual.py:61 without the noise
#pragma unroll
        for (int j = 0; j < D; ++j) a_env += (double)obs[j] * a.K.k[j];
        if constexpr (POLICY) {
            float mean;
            if constexpr (QUAD) mean = policy_forward16q<T, KIND, D, 1>(lds, lds + L.total, L, obs, lane, wave);
            else mean = policy_forward16<T, KIND, D, 1>(lds, L, obs, lane);
            a_env = residual_tanh(mean) + a_env;
        }
        double tr0 = 0, tr1 = 0, tr2 = 0, tr3 = 0, tr5 = 0;
        float rew;
        if constexpr (ENV == 0) {
            if (a.trace) { tr0 = (double)ph_lookup<S>(a.p, a.st.table, E.C, E.x); tr1 = (double)E.r; tr2 = (double)E.I; }
            float o3[3];
            ph_lane_step<S>(a.p, a.st.table, a_env, E, o3, rew);
            obs[0] = o3[0]; obs[1] = o3[1]; obs[2] = o3[2];
            tr3 = a_env; tr5 = E.x;
        } else {
            double z1n, z2n;
            wt_lane_noise<S>(a.wp, gid, W, nullptr, z1n, z2n);
            wt_lane_step<S>(a.wp, a_env, z1n, z2n, W, rew);
            obs[0] = (float)W.h1; obs[1] = (float)W.h2; obs[2] = (float)W.r; obs[3] = (float)W.I;
            tr0 = (double)W.h1; tr1 = (double)W.h2; tr2 = (double)W.r; tr3 = (double)W.I; tr5 = a_env;
        }
        ret += (double)rew;
        if (a.trace && writer) {   // [n_steps][6][N]: pH (y, r, I before the step | action, reward, x after);  tank (h1, h2, r, I after | reward, action)
            double* q = a.trace + (size_t)t * 6 * N + i;
            q[0] = tr0; q[(size_t)N] = tr1; q[2 * (size_t)N] = tr2; q[3 * (size_t)N] = tr3; q[4 * (size_t)N] = (double)rew;
            q[5 * (size_t)N] = tr5;
        }
    }
    if (writer) {
        if constexpr (ENV == 0) ph_lane_store<S>(a.p, a.st, i, E);
        else wt_lane_store<S>(a.wp, a.wst, i, W);
        if (a.ret) a.ret[i] += ret;
    }
}

int mlp_check(int kind, int D, int Di, int md);

template <int T, int KIND, int ENV, typename S, bool QUAD>
static int launch_eval_q(const EvalArgs<S>& a, hipStream_t s) {
    size_t lds_bytes = 0;
    if constexpr (KIND >= 0) {
        lds_bytes = ((size_t)mlp_layout(KIND, ENV == 0 ? 3 : 4, 1, T * 32).total + (QUAD ? quad_xchg_floats<T>() : 0)) * sizeof(float);
        static LdsLimit lds_limit;  // per instantiation
        PIME_RAISE_LDS(lds_limit, (rollout_eval_kernel<T, KIND, ENV, S, QUAD>), 160 * 1024);
    }
    const int per_wg = QUAD ? 16 : kEvalThreads / 64 * 16;
    hipLaunchKernelGGL((rollout_eval_kernel<T, KIND, ENV, S, QUAD>), dim3((a.n + per_wg - 1) / per_wg), dim3(kEvalThreads), lds_bytes, s, a);
    PIME_HIP_TRY(hipGetLastError());
    return PIME_OK;
}
template <int T, int KIND, int ENV, typename S>
static int launch_eval_t(const EvalArgs<S>& a, hipStream_t s) {
    if constexpr (KIND >= 0) {
        bool quad = a.n <= 4096;   // at most one tile per compute unit: split it over the workgroup's waves (csrc/rollout.hip: tiling)
        if (const char* e = std::getenv("PIME_ROLLOUT_NARROW")) quad = std::atoi(e) == 2;
        if (quad) return launch_eval_q<T, KIND, ENV, S, true>(a, s);
    }
    return launch_eval_q<T, KIND, ENV, S, false>(a, s);
}

template <typename S>
int launch_rollout_eval(int kind, int md, const EvalArgs<S>& a, hipStream_t s) {
    if (kind >= 0) {
        if (int rc = mlp_check(kind, a.env == 0 ? 3 : 4, kind == MLP_MODULAR_ACTOR ? 1 : 0, md)) return rc;
        PIME_REQUIRE(kind != MLP_CRITIC, "evaluation needs an actor image");
    }
    const int T = kind < 0 ? 0 : md / 32;
#define PIME_EV(TT, KK, EE) \
    if (T == TT && kind == KK && a.env == EE) return launch_eval_t<(TT == 0 ? 1 : TT), KK, EE, S>(a, s);
    PIME_EV(0, -1, 0) PIME_EV(0, -1, 1)
    PIME_EV(4, MLP_MODULAR_ACTOR, 0) PIME_EV(2, MLP_MODULAR_ACTOR, 0) PIME_EV(4, MLP_PLAIN_ACTOR, 0) PIME_EV(2, MLP_PLAIN_ACTOR, 0)
    PIME_EV(4, MLP_MODULAR_ACTOR, 1) PIME_EV(2, MLP_MODULAR_ACTOR, 1) PIME_EV(4, MLP_PLAIN_ACTOR, 1) PIME_EV(2, MLP_PLAIN_ACTOR, 1)
#undef PIME_EV
    set_error("no fused evaluation instantiation for env %d kind %d width %d", a.env, kind, md);
    return PIME_ERR_ARG;
}
template int launch_rollout_eval<float>(int, int, const EvalArgs<float>&, hipStream_t);
template int launch_rollout_eval<double>(int, int, const EvalArgs<double>&, hipStream_t);

}  // namespace pime
