// Fused on-device exploration for the off-policy agents (TD3 / residual TD3) on a vectorised env: ONE launch advances every lane
// `n_steps` lock-steps -- deterministic actor forward on the f32 matrix cores, clipped exploration noise, residual composition,
// env step with in-kernel auto-reset, and the transition (state, reward * scale, mask, action) written straight into the device
// ring buffer -- continuing the running episodes (off-policy exploration does not start from a reset).
//
// replaces, per lock-step: AgentBase.explore_env's body (/root/reference/elegantrl/agent.py:54-70) with AgentTD3.select_action
// (:300-306: a = (act(s) + N(0, explore_noise)).clamp(-1, 1)), env.step, buffer.append_buffer(state, (reward * scale, 0 if done
// else gamma, action)) (replay.py:290-300), and for the residual agent the prior term a_env = a + s @ priorK
// (agent_residual.py:61's composition with the TD3 pieces) -- ~20 PyTorch / HIP launches per lock-step in
// round 2 (42 ms per 200 lock-steps of 4 096 lanes, profiles/r03d_td3_kstats.txt).
// Policy forward = rollout_policy.hpp with the TD3 Actor's activations (three ReLU layers: CriticAdv's image kind); env arithmetic
// = env_device.hpp; exploration noise = the rollout kernel's Philox stream 2.
#include <cstdlib>
#include "env_device.hpp"
#include "rollout_offpolicy.hpp"
#include "rollout_policy.hpp"

namespace pime {

constexpr uint32_t STREAM_EXPLORE_OFFPOLICY = 2;   // = rollout.hip's STREAM_EXPLORE
constexpr int kOffThreads = 256;   // four waves of 16 lanes: one per SIMD (16-lane tiles, rollout_policy.hpp: policy_forward16)

// QUAD (launches of <= 4 096 lanes): one 16-lane tile per workgroup, split over its four waves (rollout_policy.hpp: policy_forward16q)
template <int T, int ENV, bool QUAD>
__global__ __launch_bounds__(kOffThreads) void rollout_offpolicy_kernel(OffPolicyArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int D = ENV == 0 ? 3 : 4;
    const MlpLayout L = mlp_layout(MLP_CRITIC, D, 0, T * 32);
    stage_image(lds, a.img, L.total / 4);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int N = a.n;
    const int m = (QUAD ? blockIdx.x : blockIdx.x * (kOffThreads / 64) + wave) * 16 + (lane & 15);
    const bool valid = m < N;
    const int i = valid ? m : N - 1;  // idle lanes shadow the last env (compute, never store)
    const bool writer = valid && (lane >> 4) == 0 && (!QUAD || wave == 0);   // the lane groups (QUAD: and the waves) carry copies of the same 16 envs
    const uint32_t gid = a.env_offset + (uint32_t)i;

    PhLane<float> E{};
    WtLane<float> W{};
    if constexpr (ENV == 0) ph_lane_load<float>(a.p, a.st, i, E);
    else wt_lane_load<float>(a.wp, a.wst, i, W);
    float obs[D];
#pragma unroll
    for (int j = 0; j < D; ++j) obs[j] = a.obs[(size_t)D * i + j];
    int slot = a.slot0;
    for (int t = 0; t < a.n_steps; ++t) {
        PIME_NO_HOIST();
        float mean;
        if constexpr (QUAD) mean = policy_forward16q<T, MLP_CRITIC, D, 0>(lds, lds + L.total, L, obs, lane, wave);
        else mean = policy_forward16<T, MLP_CRITIC, D, 0>(lds, L, obs, lane);
        double ua, ub;
        philox_pair(a.noise_seed, gid, a.noise_epoch, (uint32_t)t, STREAM_EXPLORE_OFFPOLICY, ua, ub);
        const float eps = (float)(sqrt(-2.0 * log(1.0 - ua)) * cos(6.283185307179586476925286766559 * ub));
        const float act = clip(tanhf(mean) + eps * a.explore_noise, -1.0f, 1.0f);   // agent.py:303-305
        double a_env = (double)act;
#pragma unroll
        for (int j = 0; j < D; ++j) a_env += (double)obs[j] * a.K.k[j];
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
            nxt[0] = W.h1; nxt[1] = W.h2; nxt[2] = W.r; nxt[3] = W.I;
        }
        if (writer) {   // replay.py:290-300: the state the action was taken in; (reward * scale, mask, action)
            float* s = a.ring_state + ((size_t)slot * N + i) * D;
#pragma unroll
            for (int j = 0; j < D; ++j) s[j] = obs[j];
            float* o = a.ring_other + ((size_t)slot * N + i) * 3;
            o[0] = rew * a.reward_scale; o[1] = d ? 0.0f : a.gamma; o[2] = act;
        }
        slot = slot + 1 == a.slots ? 0 : slot + 1;
#pragma unroll
        for (int j = 0; j < D; ++j) obs[j] = nxt[j];
    }
    if (writer) {
        if constexpr (ENV == 0) ph_lane_store<float>(a.p, a.st, i, E);
        else wt_lane_store<float>(a.wp, a.wst, i, W);
#pragma unroll
        for (int j = 0; j < D; ++j) a.obs[(size_t)D * i + j] = obs[j];
    }
}

int mlp_check(int kind, int D, int Di, int md);

template <int T, int ENV, bool QUAD>
static int launch_off_q(const OffPolicyArgs& a, hipStream_t s) {
    const size_t lds_bytes = ((size_t)mlp_layout(MLP_CRITIC, ENV == 0 ? 3 : 4, 0, T * 32).total + (QUAD ? quad_xchg_floats<T>() : 0)) * sizeof(float);
    static LdsLimit lds_limit;  // per instantiation
    PIME_RAISE_LDS(lds_limit, (rollout_offpolicy_kernel<T, ENV, QUAD>), 160 * 1024);
    const int per_wg = QUAD ? 16 : kOffThreads / 64 * 16;
    hipLaunchKernelGGL((rollout_offpolicy_kernel<T, ENV, QUAD>), dim3((a.n + per_wg - 1) / per_wg), dim3(kOffThreads), lds_bytes, s, a);
    PIME_HIP_TRY(hipGetLastError());
    return PIME_OK;
}
template <int T, int ENV>
static int launch_off_t(const OffPolicyArgs& a, hipStream_t s) {
    bool quad = a.n <= 4096;   // at most one tile per compute unit: split it over the workgroup's waves (csrc/rollout.hip: tiling)
    if (const char* e = std::getenv("PIME_ROLLOUT_NARROW")) quad = std::atoi(e) == 2;
    return quad ? launch_off_q<T, ENV, true>(a, s) : launch_off_q<T, ENV, false>(a, s);
}

int launch_rollout_offpolicy(int md, const OffPolicyArgs& a, hipStream_t s) {
    if (int rc = mlp_check(MLP_CRITIC, a.env == 0 ? 3 : 4, 0, md)) return rc;
    if (md == 128 && a.env == 0) return launch_off_t<4, 0>(a, s);
    if (md == 128 && a.env == 1) return launch_off_t<4, 1>(a, s);
    if (md == 64 && a.env == 0) return launch_off_t<2, 0>(a, s);
    if (md == 64 && a.env == 1) return launch_off_t<2, 1>(a, s);
    set_error("no fused off-policy rollout instantiation for env %d width %d", a.env, md);
    return PIME_ERR_ARG;
}

}  // namespace pime
