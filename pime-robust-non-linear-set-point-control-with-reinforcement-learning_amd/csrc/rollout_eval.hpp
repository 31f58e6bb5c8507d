// Launch arguments of the fused evaluation kernel, shared by rollout_eval.hip and abi.hip.
#pragma once
#include "env_state.hpp"

namespace pime {
constexpr int kMaxSetpoints = 16;
template <typename S>
struct EvalArgs {
    int env;                 // 0: pH (obs [y, r, I]); 1: water tank, Integrator observation [h1, h2, r, I]
    int n;                   // lanes
    uint32_t env_offset;
    PhParams p;
    PhPtrs<S> st;
    WtParams wp;
    WtPtrs<S> wst;
    const float* img;        // packed actor forward image (pime_mlp_pack), unused for the prior controller alone
    PriorK K;
    int n_steps;
    int seg_len;             // 0: no set-point schedule; else a segment boundary every seg_len steps
    double setpoint[kMaxSetpoints];
    double* ret;             // [N] += sum of the launch's rewards, or NULL
    double* trace;           // [n_steps][6][N] or NULL
};
template <typename S>
int launch_rollout_eval(int kind, int md, const EvalArgs<S>& a, hipStream_t s);
}  // namespace pime
