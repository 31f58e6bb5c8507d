// Launch arguments of the fused off-policy (TD3) exploration kernel, shared by rollout_offpolicy.hip and abi.hip.
#pragma once
#include "env_state.hpp"

namespace pime {
struct OffPolicyArgs {
    int env;                 // 0: pH; 1: water tank, Integrator observation
    int n;                   // lanes
    uint32_t env_offset;
    PhParams p;
    PhPtrs<float> st;
    WtParams wp;
    WtPtrs<float> wst;
    const float* img;        // packed deterministic actor: pime_mlp_pack image of kind PIME_MLP_CRITIC (same shape and ReLUs)
    PriorK K;                // prior-controller gain of the residual composition; zeros for plain TD3
    float explore_noise, gamma, reward_scale;
    int n_steps;
    uint64_t noise_seed;     // Philox key of the exploration noise (stream 2), counter (lane, noise_epoch, t)
    uint32_t noise_epoch;
    float* obs;              // [N, D]: in = the lanes' current observation, out = the observation after the last step
    float* ring_state;       // [slots, N, D]
    float* ring_other;       // [slots, N, 3] = (reward * scale, mask, action)
    int slot0, slots;        // first slot written; the ring wraps at `slots`
};
int launch_rollout_offpolicy(int md, const OffPolicyArgs& a, hipStream_t s);
}  // namespace pime
