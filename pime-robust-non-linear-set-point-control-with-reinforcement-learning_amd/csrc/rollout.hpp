// Launch arguments of the fused rollout kernel, shared by rollout.hip and abi.hip.
#pragma once
#include "env_state.hpp"

namespace pime {
struct RolloutArgs {
    int env;                 // 0: pH (obs [y, r, I]); 1: water tank, Integrator observation [h1, h2, r, I]; 2: water tank, Stacking
    int n;                   // lanes
    uint32_t env_offset;
    PhParams p;
    PhPtrs<float> st;
    WtParams wp;
    WtPtrs<float> wst;
    const float* img;        // packed actor forward image (pime_mlp_pack)
    const float* a_std_log;  // [1]
    PriorK K;
    int n_steps;
    uint64_t noise_seed;     // Philox key of the exploration noise (stream 2), counter (lane, noise_epoch, t)
    uint32_t noise_epoch;
    float *state, *action, *noise, *reward;  // [n_steps+1, N, D], [n_steps, N] x3
    uint8_t* done;                           // [n_steps, N]
    // PIME_STATE_MIXED16 (pime_rollout_h): the handle's binary16 integrated-error array (st.I / wst.I are NULL then) and binary16
    // observation / reward rows instead of `state` / `reward`; the policy sees the binary16 observation, as with the *_h steps
    half_t *I16, *state_h, *reward_h;
    // evaluation mode of the width-256 kernel (pime_rollout_eval at width 256): deterministic policy (no exploration noise), no
    // auto-reset, no trajectory writes; ret[lane] += the launch's sum of rewards (state / action / noise / reward / done unused)
    int eval_mode;
    double* ret;
    // ... with the evaluation kernel's extras (rollout_eval.hpp): a set-point schedule (every seg_len steps the set-point becomes
    // setpoint[t / seg_len], the integrated error and the episode clock restart, a new noise episode begins) and a float64 trace
    // [n_steps][6][N] (pH: y, r, I before the step | env action, reward, x after; tank: h1, h2, r, I after | reward, env action)
    int seg_len;
    double setpoint[16];
    double* trace;
};
}  // namespace pime
