// Launch-argument blocks of the fused PPO gradient kernels, shared by ppo_train.hip and abi.hip.
#pragma once
#include <stdint.h>

namespace pime {

struct PackArgs {
    const float* p[12];   // nn.Linear (W, b) pairs in module order
    int kind, D, Di, md;
};

struct PpoArgs {
    const float *state, *action, *logprob, *adv, *r_sum;  // flat trajectory buffers, rows = transitions
    const int64_t* indices;                               // [B] minibatch rows (row index_row[0] of a [rows][B] table)
    const int64_t* index_row;                             // NULL, or the device-side row cursor of the index table
    int B, D, Di;
    const float* a_std_log;     // actor: [1] parameter
    double* moments;            // critic: [2] sum and sum of squares of the minibatch targets (float64 atomics)
    float ratio_clip, lambda_entropy;
    const float *img_fwd, *img_bwd;
    float *stash, *dout, *xg;  // xg: [tiles*32][D] gathered minibatch states
    float* loss_sums;  // [4]: sum(-surrogate), sum(entropy proxy), sum(smooth-l1), unused
    float* g_std;      // actor: gradient of a_std_log (accumulated)
    int stagger;       // start delay of waves 4-7 in units of s_sleep(127) (8128 cycles)
    int trace_wg;      // workgroup whose marks are recorded
    long long* trace_span;  // tuning aid: [workgroup][2] start / end wall clock of every workgroup
    long long* trace;  // tuning aid (PIME_FUSED_TRACE): wall-clock marks of workgroup 0 / wave 0, NULL in production
    float* grad[12];   // split pipeline: gradient tensors in nn.Linear (W, b) order, accumulated with atomics
    float* slab;       // fused kernel: per-workgroup partial gradients [gridDim.x][slab_stride] (slab_layout order)
    int slab_stride, poff[13];  // float offsets of the params inside a slab; poff[np] = scalar slot (g_std / moments)
};

// The optimizer step fused into ppo_grad_reduce_kernel (pime_ppo_minibatch_step): torch.optim.Adam on the element just reduced
// and, with an image map, the element's new value written straight into the packed images (no re-pack launch).
struct ReduceAdam {
    float *flat_grad, *flat_param, *exp_avg, *exp_avg_sq, *step;   // step[0] counter, step[1] arrival counter (scratch)
    long long n;
    float lr, b1, b2, eps;
    const int32_t* image_map;   // [2 n]: position in img[net][0] / img[net][1] of every flat element, -1 = none; NULL = re-pack later
    float* img[2][2];           // [0 critic, 1 actor][0 forward, 1 transposed]
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

// Slab layout of one net: every parameter padded to a multiple of 4 floats, then a 4-float scalar slot
// (actor: d loss / d a_std_log; critic: the two float64 target moments).  Returns the stride (floats).
inline int slab_layout(int kind, int D, int Di, int md, int* poff, int* psize) {
    int np;
    if (kind == 2) {
        const int Do = D - Di, s[12] = {md * Do, md, (md / 2) * md, md / 2, md * Di, md, (md / 2) * md, md / 2, md * md, md, md, 1};
        np = 12;
        for (int i = 0; i < np; ++i) psize[i] = s[i];
    } else {
        const int s[8] = {md * D, md, md * md, md, md * md, md, md, 1};
        np = 8;
        for (int i = 0; i < np; ++i) psize[i] = s[i];
    }
    int o = 0;
    for (int i = 0; i < np; ++i) { poff[i] = o; o += (psize[i] + 3) & ~3; }
    poff[np] = o;
    return o + 4;
}

// Slab layout of a net of the 16-tile family (csrc/mlp16.hip): the three weight matrices are stored BLOCK-major in
// accumulator order -- block (a, b) of 16 x 16, then lane, then register: element ((a TB + b) 64 + lane) 4 + r is
// dW[16a + 4 (lane >> 4) + r][16b + (lane & 15)] -- so that a lane stores its four accumulator registers as ONE 16-byte
// word (a wave: 1 KB contiguous; in tensor order the same block is 64 stores of 64-byte segments).  The slab reduction
// un-permutes while it writes the 270 KB result.  First-layer columns are padded to TB0 = 1 or 2 tiles.
inline int slab_layout16(int D, int md, int* poff, int* psize) {
    const int tb0 = ((D + 3) & ~3) <= 16 ? 1 : 2;
    const int s[8] = {md * tb0 * 16, md, md * md, md, md * md, md, md, 1};
    int o = 0;
    for (int i = 0; i < 8; ++i) { psize[i] = s[i]; poff[i] = o; o += (s[i] + 3) & ~3; }
    poff[8] = o;
    return o + 4;
}

// The modular actor in the 16-tile family (mlp16.hip: ppo16m_kernel), parameter order of pime_ppo_net (other_net.0, other_net.2,
// integrator_net.0, integrator_net.2, net.0, net.2; W, b each): the five weight matrices block-major in accumulator order (first
// layers padded to one 16-column tile), biases and the head in tensor order.
inline int slab_layout16m(int md, int* poff, int* psize) {
    const int s[12] = {md * 16, md, (md / 2) * md, md / 2, md * 16, md, (md / 2) * md, md / 2, md * md, md, md, 1};
    int o = 0;
    for (int i = 0; i < 12; ++i) { psize[i] = s[i]; poff[i] = o; o += (s[i] + 3) & ~3; }
    poff[12] = o;
    return o + 4;
}

constexpr int kMaxDwJobs = 16;

struct DwArgs {
    DwJob job[kMaxDwJobs];
    int njobs, tiles_per_wg, debug_skip;  // debug_skip: timing-only ablation bits (PIME_DW_DEBUG env), 0 in production
};

}  // namespace pime
