// Launch-argument blocks and parameter / slab layouts of the fused TD3 optimizer step (td3_fused.hip), shared with abi.hip.
#pragma once
#include <stdint.h>

namespace pime {

constexpr int kTd3MaxD = 7;        // state columns: the critic's first layer sees D + 1 <= 8 inputs (two 16x16x4 k-steps)
constexpr int kTd3MaxSlabs = 512;  // workgroups (= partial-gradient slabs) per gradient launch

__host__ __device__ constexpr int td3_align4(int v) { return (v + 3) & ~3; }

// Flat parameter layout of the two nets: nn.Module parameter order, every tensor starting on a multiple of 4 floats (16-byte
// vector loads of rows, biases and heads); the padding words are zero and stay zero (their gradient is never written).
//   Actor      (net.py:96-110):   net.0.w [md][D], net.0.b, net.2.w [md][md], net.2.b, net.4.w, net.4.b, net.6.w [1][md], net.6.b
//   CriticTwin (net.py:305-332):  net_sa.0.w [md][D+1], net_sa.0.b, net_sa.2.w, net_sa.2.b, net_q1.w [1][md], net_q1.b, net_q2.w, net_q2.b
struct Td3ActorOff { int W1, b1, W2, b2, W3, b3, w4, b4, total; };
struct Td3CriticOff { int W1, b1, W2, b2, q1w, q1b, q2w, q2b, total; };

__host__ __device__ inline Td3ActorOff td3_actor_off(int D, int md) {
    Td3ActorOff o{};
    int p = 0;
    auto seg = [&](int& f, int n) { f = p; p = td3_align4(p + n); };
    seg(o.W1, md * D); seg(o.b1, md); seg(o.W2, md * md); seg(o.b2, md); seg(o.W3, md * md); seg(o.b3, md); seg(o.w4, md); seg(o.b4, 1);
    o.total = p;
    return o;
}
__host__ __device__ inline Td3CriticOff td3_critic_off(int D, int md) {
    Td3CriticOff o{};
    int p = 0;
    auto seg = [&](int& f, int n) { f = p; p = td3_align4(p + n); };
    seg(o.W1, md * (D + 1)); seg(o.b1, md); seg(o.W2, md * md); seg(o.b2, md); seg(o.q1w, md); seg(o.q1b, 1); seg(o.q2w, md); seg(o.q2b, 1);
    o.total = p;
    return o;
}

// One workgroup's partial gradient ("slab").  Weight matrices are BLOCK-major in accumulator order -- element ((a TB + b) 64 + lane) 4 + r
// is dW[16 a + 4 (lane >> 4) + r][16 b + (lane & 15)] (one 16-byte store per lane and block; the first layers are one column tile, TB = 1)
// -- vectors in tensor order; then a 4-float scalar slot ([0] = the loss sum of the group's samples).
struct Td3Seg {
    int slab_off, n4;      // position (floats) and length (16-byte words) inside a slab
    int flat_off;          // the tensor inside the flat parameter / gradient buffer
    int tb;                // 0: tensor order; > 0: block-major with tb column tiles
    int ldw, ncols;        // tb > 0: row length and valid columns of the tensor
    int n;                 // tb == 0: valid floats (the rest of the last word is padding)
};
struct Td3SlabLayout {
    Td3Seg seg[8];
    int nseg, scalar_off, stride;
};
__host__ __device__ inline Td3SlabLayout td3_actor_slab(int D, int md) {
    const Td3ActorOff P = td3_actor_off(D, md);
    const int NT = md / 16;
    Td3SlabLayout L{};
    int o = 0, k = 0;
    auto mat = [&](int flat, int tb, int ldw) { L.seg[k++] = Td3Seg{o, NT * tb * 64, flat, tb, ldw, ldw, 0}; o += NT * tb * 256; };
    auto vec = [&](int flat, int n) { L.seg[k++] = Td3Seg{o, td3_align4(n) / 4, flat, 0, 0, 0, n}; o += td3_align4(n); };
    mat(P.W1, 1, D); vec(P.b1, md); mat(P.W2, NT, md); vec(P.b2, md); mat(P.W3, NT, md); vec(P.b3, md); vec(P.w4, md); vec(P.b4, 1);
    L.nseg = k; L.scalar_off = o; L.stride = o + 4;
    return L;
}
__host__ __device__ inline Td3SlabLayout td3_critic_slab(int D, int md) {
    const Td3CriticOff P = td3_critic_off(D, md);
    const int NT = md / 16;
    Td3SlabLayout L{};
    int o = 0, k = 0;
    auto mat = [&](int flat, int tb, int ldw) { L.seg[k++] = Td3Seg{o, NT * tb * 64, flat, tb, ldw, ldw, 0}; o += NT * tb * 256; };
    auto vec = [&](int flat, int n) { L.seg[k++] = Td3Seg{o, td3_align4(n) / 4, flat, 0, 0, 0, n}; o += td3_align4(n); };
    mat(P.W1, 1, D + 1); vec(P.b1, md); mat(P.W2, NT, md); vec(P.b2, md); vec(P.q1w, md); vec(P.q1b, 1); vec(P.q2w, md); vec(P.q2b, 1);
    L.nseg = k; L.scalar_off = o; L.stride = o + 4;
    return L;
}

struct Td3Batch {
    const float* state;        // [rows][D] replay states
    const float* other;        // [rows][3]: reward * scale, mask (0 | gamma), action
    const int64_t* idx;        // [table rows][B] sampled rows; the successor state of row idx is row nxt
    const int64_t* nxt;
    const float* noise;        // [table rows][B] standard normal draws of the target policy smoothing, or NULL: Philox in the kernel
    long long row;             // table row of this optimizer step (a launch argument: every node of a captured update graph carries its own)
    const int64_t* epoch;      // NULL, or [dev] int64[1] added to noise_epoch (bumped by the host per update: fresh noise in every graph replay)
    int B;
    uint64_t noise_seed;       // noise == NULL: Philox key; counter (batch position, noise_epoch, table row, stream 3)
    uint32_t noise_epoch;
    float policy_noise, noise_clip;
};

struct Td3GradArgs {
    Td3Batch b;
    int D;
    const float *act, *cri;    // critic launch: TARGET actor, ONLINE critic;  actor launch: ONLINE actor, TARGET critic
    const float* cri_target;   // critic launch only: the target critic
    float* slab;               // [grid][stride]
    float* xg;                 // [B][8]: the minibatch's state rows (+ action), gathered by the critic launch, read back by the actor launch
    int stride, ngroups;
    long long* trace;          // tuning aid (PIME_TD3_TRACE): wall-clock marks of workgroup 0 / thread 0, NULL in production
};

struct Td3ApplyArgs {
    Td3SlabLayout L;
    const float* slab;
    int nslabs;
    float *param, *target, *grad, *exp_avg, *exp_avg_sq;   // flat tensors of this net (layout above)
    const float* step;         // [0] optimizer steps applied before table row 0 (the host adds an update's step count behind it)
    long long row;             // this launch is Adam step number step[0] + row + 1
    float lr, b1, b2, eps, tau;
    int soft;                  // 1: soft target update in this launch (delayed steps: row % update_freq == 0, agent.py:320-321,330-331)
    int mode;                  // 0: slab reduction + Adam (+ soft update); data-parallel callers split it around their all-reduce of `grad`:
                               // 1: slab reduction only (grad = the sum, loss words), 2: Adam (+ soft update) from `grad`, no slabs read
    float* loss;               // [4]: [slot] += value of this step, [2 + slot] = value of this step
    int loss_slot;             // 0: actor objective = -(q sum) / B, 1: critic objective = (loss sum) / B
    float inv_B;
};

}  // namespace pime
