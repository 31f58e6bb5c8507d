// Launch-argument blocks of the fused PPO gradient kernels, shared by ppo_train.hip and abi.hip.
#pragma once
#include <stdint.h>

namespace pime {

struct PpoArgs {
    const float *state, *action, *logprob, *adv, *r_sum;  // flat trajectory buffers, rows = transitions
    const int64_t* indices;                               // [B] minibatch rows
    int B, D, Di;
    const float* a_std_log;     // actor: [1] parameter
    double* moments;            // critic: [2] sum and sum of squares of the minibatch targets (float64 atomics)
    float ratio_clip, lambda_entropy;
    const float *img_fwd, *img_bwd;
    float *stash, *dout, *xg;  // xg: [tiles*32][D] gathered minibatch states
    float* loss_sums;  // [4]: sum(-surrogate), sum(entropy proxy), sum(smooth-l1), unused
    float* g_std;      // actor: gradient of a_std_log (accumulated)
    int stagger;       // start delay of waves 4-7 in units of s_sleep(127) (8128 cycles)
};

struct DwJob {
    const float* a_stash;  // dZ stash base (NULL => head job: the A "tile" is the dOut vector in feature 0)
    int a_nt, a_t0, a_tiles;
    const float* dout;
    int b_kind;            // 0: stash tiles, 1: first layer recomputed from the state, 2: raw state columns
    const float* b_stash;
    int b_nt, b_t0, b_tiles;
    const float *W, *bias; // b_kind 1: nn.Linear [32*b_tiles][Din] + bias
    int Din, col0, act;    // b_kind 1/2: state columns [col0, col0+Din); act of the recomputed layer
    float *dW, *db;        // dW row-major [a feats][ldw]
    int ldw, out_rows, out_cols;
    const float* xg;       // [tiles*32][D] gathered minibatch states (written by the net kernel)
    int D, B;
};

constexpr int kMaxDwJobs = 16;

struct DwArgs {
    DwJob job[kMaxDwJobs];
    int njobs, tiles_per_wg, debug_skip;  // debug_skip: timing-only ablation bits (PIME_DW_DEBUG env), 0 in production
};

}  // namespace pime
