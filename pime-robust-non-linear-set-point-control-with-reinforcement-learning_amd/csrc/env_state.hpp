// Device-side layout of the env state (SoA in HBM) and the per-launch parameter blocks.
#pragma once
#include "pime_common.hpp"

namespace pime {

// -------- pH (gym_control/envs/ph.py) -------------------------------------------------------------------
struct PhParams {
    int32_t n, max_steps, reward_type, integral_bound, resample_every, table_len, auto_reset, has_punish;
    uint32_t env_offset;
    uint64_t seed;
    double integral_max, integral_punish, action_punish, action_change_punish, thr;
    double sample_t, u_low, u_high, table_scale;
    double qww_lo, qww_hi, qc_lo, qc_hi, x0_lo, x0_hi, r_lo, r_hi;
};

// S = storage/arithmetic type of the "slow" state words (double: PIME_STATE_F64, float: PIME_STATE_MIXED).
// x and the ZOH plant stay float64 in both modes so that k = rint(C*x*1e5) is the reference's index.
// SI = storage type of the integrated error I: S, or IEEE binary16 under S = float (PIME_STATE_MIXED16: BASELINE.json config
// 5's "fp16 state" = SURVEY §8(d) cfg 5 "fp16 storage for obs / reward / I, f32 math, f64 x"; binary16 is a storage format
// here, never an arithmetic type).
using half_t = _Float16;
template <typename S, typename SI = S>
struct PhPtrs {
    double *x, *A, *B, *C, *qww, *qc;
    S *r, *last_a;
    SI* I;
    int32_t *t, *episode;
    const S* table;
};

// -------- water tank (gym_control/envs/nonlinear_watertank.py) ---------------------------------------------
struct WtParams {
    int32_t n, max_steps, reward_type, num_stack, resample_every, n_discrete, auto_reset, obs_dim;
    uint32_t env_offset;
    uint64_t seed;
    double integral_max, integral_punish, thr;
    double A1, A2, G, dt, noise_scale, z1, pmax;
    double a1_lo, a1_hi, a2_lo, a2_hi, kp_lo, kp_hi, h_lo, h_hi, r_lo, r_hi;
};

template <typename S, typename SI = S>
struct WtPtrs {
    S *h1, *h2, *r, *a1, *a2, *kp;
    SI* I;
    // Stacking variant only: the last num_stack frames [h1, h2, r] as a RING in SoA order, frames[(slot * 3 + c) * n + lane],
    // and the per-lane slot of the OLDEST frame (the deque's left end, nonlinear_watertank.py:1143-1144): a step overwrites that
    // slot and advances head, instead of shifting 3 (S - 1) words per lane; every access is a contiguous wave-wide segment.
    S* frames;
    int32_t* head;
    int32_t *t, *episode;
};

}  // namespace pime
