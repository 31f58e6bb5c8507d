// Launch arguments of the fused rollout kernel, shared by rollout.hip and abi.hip.
#pragma once
#include "env_state.hpp"

namespace pime {
struct RolloutArgs {
    PhParams p;
    PhPtrs<float> st;
    const float* img;        // packed actor forward image (pime_mlp_pack)
    const float* a_std_log;  // [1]
    PriorK K;
    int n_steps;
    uint64_t noise_seed;     // Philox key of the exploration noise (stream 2), counter (lane, noise_epoch, t)
    uint32_t noise_epoch;
    float *state, *action, *noise, *reward;  // [n_steps+1, N, 3], [n_steps, N] x3
    uint8_t* done;                           // [n_steps, N]
};
}  // namespace pime
