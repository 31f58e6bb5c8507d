/*
 * pime_oracle.c -- CPU restatement (fp64, scalar loops) of the reference's set-point-control hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Linked/loaded solely by tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg, as the checker.  Nothing under the product package imports it; the product path
 * (libpime_hip.so) fails loudly without a GPU instead of falling back to this file.
 *
 * Parity status: PINNED.  Every function below is checked in tests/test_oracle_golden.py against the
 * golden vectors in tests/golden (npz files), which were produced by running the unmodified reference
 * (tests/golden/make_golden.py).  Two third-party pieces the reference calls are restated from their
 * published behaviour and are pinned only through those vectors: control==0.9.1 tf2ss/c2d (closed-form
 * ZOH below) and gym==0.18.0 TimeLimit.  The counter RNG (Philox4x32-10) is NOT in the reference (it has
 * one env on MT19937); it is restated from Salmon et al., SC'11 and pinned by Random123's known answers.
 *
 * All citations are /root/reference paths.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORACLE_API __attribute__((visibility("default")))

/* Env lanes and MLP rows are independent, so the lane / row loops below carry `#pragma omp parallel for` (built with
 * -fopenmp; the pragmas vanish without it).  No reduction crosses lanes: results are identical for any thread count.
 * oracle_set_threads(1) gives the single-thread CPU baseline, oracle_set_threads(nproc) the all-cores one. */
ORACLE_API int oracle_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
    return omp_get_max_threads();
#else
    (void)n;
    return 1;
#endif
}

/* ------------------------------------------------------------------------------------------------
 * Titration table  (gym_control/envs/ph.py:72-84; constants :32-37; MHCl grid gym_control/__init__.py:12)
 * quartic in [H+], 5 Newton steps H <- |H - f/f'| warm-started from the previous entry.
 * ---------------------------------------------------------------------------------------------- */
ORACLE_API void oracle_ph_table(int n, double mhcl_step, double kw, double kchem, double ka, double MNaOH,
                                double MHA, double MNH3, double* pH) {
    double H = 1e-14 / MNaOH; /* ph.py:75 */
    for (int i = 0; i < n; ++i) {
        const double m = (double)i * mhcl_step; /* np.arange(0., 1, step) element i */
        const double ak = MNH3 - m + MNaOH + kchem + ka;                                                  /* :77 */
        const double bk = (kchem + ka) * MNaOH - (kchem + ka) * m - kw + MNH3 * ka + kchem * ka - ka * MHA; /* :78 */
        const double ck = MNaOH * kchem * ka - kw * (ka + kchem) - m * kchem * ka - ka * kchem * MHA;      /* :79 */
        const double dk = -kchem * ka * kw;                                                                /* :80 */
        for (int j = 0; j < 5; ++j) { /* :81-82, numpy float64 scalar ** int -> libm pow */
            const double f = pow(H, 4) + ak * pow(H, 3) + bk * pow(H, 2) + ck * H + dk;
            const double fp = 4 * pow(H, 3) + 3 * ak * pow(H, 2) + 2 * bk * H + ck;
            H = fabs(H - f / fp);
        }
        pH[i] = -1 * log10(H); /* :83 */
    }
}

/* ------------------------------------------------------------------------------------------------
 * ZOH discretisation of  G(s) = qc_V / (s + qww_V)  at sample_t  (ph.py:114-121).
 * control.tf2ss -> A_c = -qww_V, B_c = 1, C = qc_V; c2d 'zoh' -> A = e^{A_c T}, B = (A-1)/A_c.
 * ---------------------------------------------------------------------------------------------- */
ORACLE_API void oracle_ph_zoh(double qww_V, double qc_V, double sample_t, double* A, double* B, double* C) {
    *A = exp(-qww_V * sample_t);
    *B = -expm1(-qww_V * sample_t) / qww_V;
    *C = qc_V;
}

/* ------------------------------------------------------------------------------------------------
 * Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11).
 * ---------------------------------------------------------------------------------------------- */
ORACLE_API void oracle_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* 53-bit uniform in [0,1) from two words, the construction numpy's random_sample uses. */
static double u53(uint32_t a, uint32_t b) {
    return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) / 9007199254740992.0;
}

/* draw stream layout (DESIGN.md "Counter RNG"): counter = (env id, episode, slot, stream) */
enum { STREAM_RESET = 0, STREAM_NOISE = 1 };

static void philox_pair(uint64_t seed, uint32_t env, uint32_t episode, uint32_t slot, uint32_t stream,
                        double* ua, double* ub) {
    uint32_t ctr[4] = {env, episode, slot, stream}, key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)}, o[4];
    oracle_philox4x32_10(ctr, key, o);
    *ua = u53(o[0], o[1]);
    *ub = u53(o[2], o[3]);
}

ORACLE_API void oracle_philox_uniform_pair(uint64_t seed, uint32_t env, uint32_t episode, uint32_t slot,
                                           uint32_t stream, double* out2) {
    philox_pair(seed, env, episode, slot, stream, &out2[0], &out2[1]);
}

/* Exploration noise of the fused rollout (csrc/rollout.hip): eps ~ N(0,1) for lane i at step t of rollout `epoch`.
 * The reference draws torch.randn_like on the host (net_residual.py:178); the device path has no torch generator, so this
 * is new: Box-Muller (cosine branch) on the Philox pair of counter (global lane, epoch, t, stream 2), in float64, rounded
 * to float32 once. */
enum { STREAM_EXPLORE = 2 };
ORACLE_API void oracle_explore_noise(uint64_t seed, uint32_t env_offset, int n, uint32_t epoch, uint32_t t, float* eps) {
#pragma omp parallel for schedule(static) if (n >= 512)
    for (int i = 0; i < n; ++i) {
        double ua, ub;
        philox_pair(seed, env_offset + (uint32_t)i, epoch, t, STREAM_EXPLORE, &ua, &ub);
        eps[i] = (float)(sqrt(-2.0 * log(1.0 - ua)) * cos(6.283185307179586476925286766559 * ub));
    }
}

static double clipd(double v, double lo, double hi) { /* np.clip = minimum(maximum(v, lo), hi) */
    double m = v > lo ? v : lo;
    return m < hi ? m : hi;
}

static double reward_of(int reward_type, double achieved, double goal, double thr) {
    /* goal_distance + compute_reward: ph.py:8-12,202-225 / nonlinear_watertank.py:64-72,484-514 */
    const double d = fabs(achieved - goal);
    switch (reward_type) {
        case 0: return -d;                       /* 'distance' */
        case 1: return -(d * d);                 /* 'square_distance' */
        default: return -(double)(float)(d > thr); /* 'sparse': -(d > thr).astype(float32) */
    }
}

/* ================================================================================================
 * pH env, N independent instances  (PH1DChangingParamUniformGoalIntegrator[_NoBound] + gym TimeLimit)
 * ============================================================================================== */
typedef struct {
    int n, max_steps, reward_type, integral_bound, resample_every, table_len;
    double integral_max, integral_punish, action_punish, action_change_punish, thr;
    double sample_t, u_low, u_high, x0_lo, x0_hi, r_lo, r_hi, qww_lo, qww_hi, qc_lo, qc_hi, table_scale;
    uint64_t seed;
    uint32_t env_offset;
    const double* table;
    double *x, *I, *r, *y, *A, *B, *C, *qww, *qc, *last_a;
    int32_t *t, *episode;
} oracle_ph;

ORACLE_API oracle_ph* oracle_ph_create(int n, int max_steps, int reward_type, int integral_bound,
                                       int resample_every, const double* table, int table_len, uint64_t seed,
                                       uint32_t env_offset) {
    oracle_ph* e = (oracle_ph*)calloc(1, sizeof(oracle_ph));
    e->n = n; e->max_steps = max_steps; e->reward_type = reward_type; e->integral_bound = integral_bound;
    e->resample_every = resample_every; e->table = table; e->table_len = table_len; e->seed = seed;
    e->env_offset = env_offset;
    e->integral_max = 25.0;  /* ph.py:299 */
    e->thr = 0.05;           /* ph.py:44 */
    e->sample_t = 20.0;      /* ph.py:41 */
    e->u_low = 0.0; e->u_high = 1.5;      /* ph.py:146-147 */
    e->x0_lo = 0.0; e->x0_hi = 50.0;      /* ph.py:420 */
    e->r_lo = 3.0; e->r_hi = 11.0;        /* ph.py:424 */
    e->qww_lo = 0.005; e->qww_hi = 0.015; /* ph.py:357 */
    e->qc_lo = 0.0015; e->qc_hi = 0.0025; /* ph.py:358 */
    e->table_scale = 1e5;                 /* np.around(.., 5) on the 1e-5 MHCl grid, ph.py:188 */
    double** f[] = {&e->x, &e->I, &e->r, &e->y, &e->A, &e->B, &e->C, &e->qww, &e->qc, &e->last_a};
    for (unsigned i = 0; i < sizeof(f) / sizeof(f[0]); ++i) *f[i] = (double*)calloc((size_t)n, sizeof(double));
    e->t = (int32_t*)calloc((size_t)n, sizeof(int32_t));
    e->episode = (int32_t*)calloc((size_t)n, sizeof(int32_t));
    for (int i = 0; i < n; ++i) e->episode[i] = -1;
    return e;
}

/* ensemble ranges of sample_parameters (ph.py:409-411): gym.make kwargs qww_V / qc_V (gym_control/__init__.py) */
ORACLE_API void oracle_ph_set_ranges(oracle_ph* e, double qww_lo, double qww_hi, double qc_lo, double qc_hi) {
    e->qww_lo = qww_lo; e->qww_hi = qww_hi; e->qc_lo = qc_lo; e->qc_hi = qc_hi;
}

ORACLE_API void oracle_ph_set_punish(oracle_ph* e, double integral_punish, double action_punish,
                                     double action_change_punish) {
    e->integral_punish = integral_punish; e->action_punish = action_punish;
    e->action_change_punish = action_change_punish;
}

ORACLE_API void oracle_ph_destroy(oracle_ph* e) {
    if (!e) return;
    free(e->x); free(e->I); free(e->r); free(e->y); free(e->A); free(e->B); free(e->C); free(e->qww);
    free(e->qc); free(e->last_a); free(e->t); free(e->episode); free(e);
}

/* fields: 0 x, 1 I, 2 r, 3 y, 4 A, 5 B, 6 C, 7 qww_V, 8 qc_V, 9 t, 10 episode */
ORACLE_API void oracle_ph_get(const oracle_ph* e, int field, double* out) {
    const double* src[] = {e->x, e->I, e->r, e->y, e->A, e->B, e->C, e->qww, e->qc};
    for (int i = 0; i < e->n; ++i)
        out[i] = field < 9 ? src[field][i] : (field == 9 ? (double)e->t[i] : (double)e->episode[i]);
}

/* observe_state: ph.py:187-189.  first index with MHCl >= around(C*x, 5)  ==  rint(C*x*1e5)   */
static double ph_lookup(const oracle_ph* e, double C, double x) {
    long k = lrint(C * x * e->table_scale); /* round-half-even, as np.around */
    if (k < 0) k = 0;
    if (k >= e->table_len) k = e->table_len - 1; /* reference raises IndexError here; unreachable in range */
    return e->table[k];
}

ORACLE_API void oracle_ph_set(oracle_ph* e, int field, const double* in) {
    double* dst[] = {e->x, e->I, e->r, e->y, e->A, e->B, e->C, e->qww, e->qc};
    for (int i = 0; i < e->n; ++i) {
        if (field < 9) dst[field][i] = in[i];
        else if (field == 9) e->t[i] = (int32_t)in[i];
        else e->episode[i] = (int32_t)in[i];
        if (field == 0) e->y[i] = ph_lookup(e, e->C[i], e->x[i]); /* set_state: ph.py:233-236 */
        if (field == 7 || field == 8) /* the build's set_params DOES rebuild the plant (quirk C3 opt-out is host side) */
            oracle_ph_zoh(e->qww[i], e->qc[i], e->sample_t, &e->A[i], &e->B[i], &e->C[i]);
    }
}

/* reset_all / reset_r: ph.py:412-445.  draws (nullable) = [n][4] final values (qww_V, qc_V, x0, r) in the
 * reference's draw order; NULL -> Philox.  resample: params are redrawn when episode % resample_every == 0
 * (resample_every == 0: never, i.e. set_reset_all(False)). */
ORACLE_API void oracle_ph_reset(oracle_ph* e, const uint8_t* mask, const double* draws, float* obs) {
#pragma omp parallel for schedule(static) if (e->n >= 512)
    for (int i = 0; i < e->n; ++i) {
        if (mask && !mask[i]) continue;
        e->episode[i] += 1;
        const int resample = e->resample_every > 0 && (e->episode[i] % e->resample_every) == 0;
        double qww, qc, x0, r;
        if (draws) {
            qww = draws[4 * i + 0]; qc = draws[4 * i + 1]; x0 = draws[4 * i + 2]; r = draws[4 * i + 3];
        } else {
            double u0, u1, u2, u3;
            philox_pair(e->seed, e->env_offset + (uint32_t)i, (uint32_t)e->episode[i], 0, STREAM_RESET, &u0, &u1);
            philox_pair(e->seed, e->env_offset + (uint32_t)i, (uint32_t)e->episode[i], 1, STREAM_RESET, &u2, &u3);
            qww = e->qww_lo + (e->qww_hi - e->qww_lo) * u0; /* np.random.uniform(lo, hi), ph.py:410 */
            qc = e->qc_lo + (e->qc_hi - e->qc_lo) * u1;
            x0 = e->x0_lo + (e->x0_hi - e->x0_lo) * u2;     /* ph.py:420 */
            r = e->r_lo + (e->r_hi - e->r_lo) * u3;         /* ph.py:424 */
        }
        if (resample) {
            e->qww[i] = qww; e->qc[i] = qc;
            oracle_ph_zoh(qww, qc, e->sample_t, &e->A[i], &e->B[i], &e->C[i]); /* update_system, ph.py:414 */
        }
        e->x[i] = x0;
        e->y[i] = ph_lookup(e, e->C[i], x0); /* :422 */
        e->t[i] = 0;                         /* :423 (and TimeLimit.reset) */
        e->r[i] = r;
        e->I[i] = 0.0;                       /* :425 */
        if (obs) { obs[3 * i + 0] = (float)e->y[i]; obs[3 * i + 1] = (float)r; obs[3 * i + 2] = 0.0f; }
    }
}

/* step: ph.py:320-348 (bounded) / :448-478 (NoBound) + TimeLimit.  obs is written as float32, the cast
 * PreprocessEnv applies (elegantrl/env.py:72); obs64/x are optional float64 taps for the parity tests.
 * auto_reset: a lane that reports done is reset in the same call (draws as in oracle_ph_reset) and its
 * obs row is the new episode's first observation. */
ORACLE_API void oracle_ph_step(oracle_ph* e, const double* action, int auto_reset, const double* reset_draws,
                               float* obs, double* obs64, double* reward, uint8_t* done) {
    uint8_t* dmask = auto_reset ? (uint8_t*)calloc((size_t)e->n, 1) : NULL;
    int any_done = 0;
#pragma omp parallel for schedule(static) reduction(| : any_done) if (e->n >= 512)
    for (int i = 0; i < e->n; ++i) {
        const double a = clipd(action[i], -1.0, 1.0);                       /* :321 */
        const double delta_u = e->t[i] != 0 ? a - e->last_a[i] : 0.0;       /* :322 */
        e->last_a[i] = a;
        e->t[i] += 1;                                                       /* :325 */
        const double u = e->u_low + (e->u_high - e->u_low) * ((a - -1.0) / (1.0 - -1.0)); /* action(): :155-159 */
        e->x[i] = e->A[i] * e->x[i] + e->B[i] * u;                          /* :330 */
        const double y = ph_lookup(e, e->C[i], e->x[i]);                    /* :332 */
        e->y[i] = y;
        double rew = reward_of(e->reward_type, y, e->r[i], e->thr);         /* :334 */
        rew -= e->action_punish * fabs(u);                                  /* :336 */
        rew -= e->action_change_punish * fabs(delta_u);                     /* :337 (norm of a 1-vector) */
        const double I_raw = e->I[i] + (e->r[i] - y);                       /* :339-340 */
        e->I[i] = e->integral_bound ? clipd(I_raw, -e->integral_max, e->integral_max) : I_raw; /* :341 / :470 */
        rew += -e->integral_punish * fabs(e->integral_bound ? I_raw : e->I[i]); /* :343 / :473 */
        const int d = e->t[i] >= e->max_steps; /* TimeLimit (gym_control/__init__.py:6); env itself: False */
        if (obs) { obs[3 * i] = (float)y; obs[3 * i + 1] = (float)e->r[i]; obs[3 * i + 2] = (float)e->I[i]; }
        if (obs64) { obs64[3 * i] = y; obs64[3 * i + 1] = e->r[i]; obs64[3 * i + 2] = e->I[i]; }
        if (reward) reward[i] = rew;
        if (done) done[i] = (uint8_t)d;
        if (dmask && d) { dmask[i] = 1; any_done = 1; }
    }
    if (any_done) oracle_ph_reset(e, dmask, reset_draws, obs);
    free(dmask);
}

/* ================================================================================================
 * Water tank, N independent instances
 * (NonLinearWaterTankChangingParamUniformGoalIntegrator / ...GoalStacking, controller_type 'P')
 * ============================================================================================== */
typedef struct {
    int n, max_steps, reward_type, num_stack, resample_every, n_discrete;
    double integral_max, integral_punish, thr, A1, A2, G, dt, noise_scale, z1, pmax;
    double a1_lo, a1_hi, a2_lo, a2_hi, kp_lo, kp_hi, h_lo, h_hi, r_lo, r_hi;
    uint64_t seed;
    uint32_t env_offset;
    double *h1, *h2, *r, *I, *a1, *a2, *kp, *frames; /* frames: [n][num_stack][3], oldest first */
    int32_t *t, *episode;
} oracle_wt;

ORACLE_API oracle_wt* oracle_wt_create(int n, int max_steps, int reward_type, int num_stack, int resample_every,
                                       double noise_scale, uint64_t seed, uint32_t env_offset) {
    oracle_wt* e = (oracle_wt*)calloc(1, sizeof(oracle_wt));
    e->n = n; e->max_steps = max_steps; e->reward_type = reward_type; e->num_stack = num_stack;
    e->resample_every = resample_every; e->noise_scale = noise_scale; e->seed = seed; e->env_offset = env_offset;
    e->n_discrete = 20; e->dt = 2.0 / 20;              /* sample_t / n_discrete, gym_control/__init__.py:62-63 */
    e->A1 = 1; e->A2 = 1; e->G = 980;                  /* :58-61 */
    e->z1 = 1; e->pmax = 10.0; e->thr = 0.05;          /* nonlinear_watertank.py:92,110,105 */
    e->integral_max = 25.0;                            /* :732 */
    e->a1_lo = 0.0015; e->a1_hi = 0.0024; e->a2_lo = 0.0015; e->a2_hi = 0.0024; e->kp_lo = 0.07; e->kp_hi = 0.17;
    e->h_lo = 0.0; e->h_hi = 10.0; e->r_lo = 0.0; e->r_hi = 10.0; /* :912-913 */
    double** f[] = {&e->h1, &e->h2, &e->r, &e->I, &e->a1, &e->a2, &e->kp};
    for (unsigned i = 0; i < sizeof(f) / sizeof(f[0]); ++i) *f[i] = (double*)calloc((size_t)n, sizeof(double));
    if (num_stack > 0) e->frames = (double*)calloc((size_t)n * num_stack * 3, sizeof(double));
    e->t = (int32_t*)calloc((size_t)n, sizeof(int32_t));
    e->episode = (int32_t*)calloc((size_t)n, sizeof(int32_t));
    for (int i = 0; i < n; ++i) e->episode[i] = -1;
    return e;
}

/* ensemble ranges of sample_parameters (nonlinear_watertank.py:890-895): gym.make kwargs a1 / a2 / Kp */
ORACLE_API void oracle_wt_set_ranges(oracle_wt* e, double a1_lo, double a1_hi, double a2_lo, double a2_hi, double kp_lo,
                                     double kp_hi) {
    e->a1_lo = a1_lo; e->a1_hi = a1_hi; e->a2_lo = a2_lo; e->a2_hi = a2_hi; e->kp_lo = kp_lo; e->kp_hi = kp_hi;
}

ORACLE_API void oracle_wt_destroy(oracle_wt* e) {
    if (!e) return;
    free(e->h1); free(e->h2); free(e->r); free(e->I); free(e->a1); free(e->a2); free(e->kp); free(e->frames);
    free(e->t); free(e->episode); free(e);
}

ORACLE_API void oracle_wt_set_punish(oracle_wt* e, double integral_punish) { e->integral_punish = integral_punish; }
ORACLE_API void oracle_wt_set_max_steps(oracle_wt* e, int max_steps) { e->max_steps = max_steps; }

ORACLE_API int oracle_wt_obs_dim(const oracle_wt* e) { return e->num_stack > 0 ? 3 * e->num_stack : 4; }

/* fields: 0 h1, 1 h2, 2 r, 3 I, 4 a1, 5 a2, 6 Kp, 7 t, 8 episode */
ORACLE_API void oracle_wt_get(const oracle_wt* e, int field, double* out) {
    const double* src[] = {e->h1, e->h2, e->r, e->I, e->a1, e->a2, e->kp};
    for (int i = 0; i < e->n; ++i)
        out[i] = field < 7 ? src[field][i] : (field == 7 ? (double)e->t[i] : (double)e->episode[i]);
}
ORACLE_API void oracle_wt_set(oracle_wt* e, int field, const double* in) {
    double* dst[] = {e->h1, e->h2, e->r, e->I, e->a1, e->a2, e->kp};
    for (int i = 0; i < e->n; ++i) {
        if (field < 7) dst[field][i] = in[i];
        else if (field == 7) e->t[i] = (int32_t)in[i];
        else e->episode[i] = (int32_t)in[i];
    }
}

static void wt_write_obs(const oracle_wt* e, int i, float* obs, double* obs64) {
    const int D = oracle_wt_obs_dim(e);
    if (e->num_stack > 0) { /* _get_observe_P of the Stacking class: :1162-1164 */
        for (int j = 0; j < D; ++j) {
            if (obs) obs[(size_t)D * i + j] = (float)e->frames[(size_t)D * i + j];
            if (obs64) obs64[(size_t)D * i + j] = e->frames[(size_t)D * i + j];
        }
    } else { /* _get_observe_P of the Integrator class: :789-793 */
        const double v[4] = {e->h1[i], e->h2[i], e->r[i], e->I[i]};
        for (int j = 0; j < 4; ++j) {
            if (obs) obs[4 * (size_t)i + j] = (float)v[j];
            if (obs64) obs64[4 * (size_t)i + j] = v[j];
        }
    }
}

/* reset_all / reset_r: :902-939 (Integrator), :1166-1203 (Stacking).  draws (nullable) = [n][6] final
 * values (a1, a2, Kp, h1, h2, r) in the reference's global-stream order; NULL -> Philox. */
ORACLE_API void oracle_wt_reset(oracle_wt* e, const uint8_t* mask, const double* draws, float* obs) {
#pragma omp parallel for schedule(static) if (e->n >= 512)
    for (int i = 0; i < e->n; ++i) {
        if (mask && !mask[i]) continue;
        e->episode[i] += 1;
        const int resample = e->resample_every > 0 && (e->episode[i] % e->resample_every) == 0;
        double v[6];
        if (draws) {
            memcpy(v, draws + 6 * (size_t)i, sizeof(v));
        } else {
            double u[6];
            for (uint32_t s = 0; s < 3; ++s)
                philox_pair(e->seed, e->env_offset + (uint32_t)i, (uint32_t)e->episode[i], s, STREAM_RESET,
                            &u[2 * s], &u[2 * s + 1]);
            v[0] = e->a1_lo + (e->a1_hi - e->a1_lo) * u[0]; /* sample_parameters :890-894 */
            v[1] = e->a2_lo + (e->a2_hi - e->a2_lo) * u[1];
            v[2] = e->kp_lo + (e->kp_hi - e->kp_lo) * u[2];
            v[3] = e->h_lo + (e->h_hi - e->h_lo) * u[3];    /* :912 */
            v[4] = e->h_lo + (e->h_hi - e->h_lo) * u[4];
            v[5] = e->r_lo + (e->r_hi - e->r_lo) * u[5];    /* :913 */
        }
        if (resample) { e->a1[i] = v[0]; e->a2[i] = v[1]; e->kp[i] = v[2]; }
        e->h1[i] = v[3]; e->h2[i] = v[4]; e->r[i] = v[5];
        e->t[i] = 0; e->I[i] = 0.0;
        if (e->num_stack > 0) /* fill every frame with the first one: :1181-1183 */
            for (int s = 0; s < e->num_stack; ++s) {
                double* f = e->frames + ((size_t)i * e->num_stack + s) * 3;
                f[0] = v[3]; f[1] = v[4]; f[2] = v[5];
            }
        if (obs) wt_write_obs(e, i, obs, NULL);
    }
}

/* step: :800-826 (Integrator) / :1118-1147 (Stacking).  noise (nullable) = [n][2] the two already-scaled
 * normals added to h1 then h2 (:810-811); NULL -> Philox Box-Muller * noise_scale. */
ORACLE_API void oracle_wt_step(oracle_wt* e, const double* action, const double* noise, int auto_reset,
                               const double* reset_draws, float* obs, double* obs64, double* reward,
                               uint8_t* done) {
    uint8_t* dmask = auto_reset ? (uint8_t*)calloc((size_t)e->n, 1) : NULL;
    int any_done = 0;
    const double lo = -0.0, hi = INFINITY; /* Box(low=-ones*0, high=inf) cast to float32: :252-257 */
#pragma omp parallel for schedule(static) reduction(| : any_done) if (e->n >= 512)
    for (int i = 0; i < e->n; ++i) {
        e->t[i] += 1;                                                  /* :801 */
        const double u = action[i] * e->pmax / 2. + e->pmax / 2.;      /* action_P :258-260; no clip of a */
        double h1 = e->h1[i], h2 = e->h2[i];
        const double a1 = e->a1[i], a2 = e->a2[i], kp = e->kp[i];
        for (int s = 0; s < e->n_discrete; ++s) {                      /* :805-809 */
            const double n1 = h1 + (-a1 / e->A1 * sqrt(2 * e->G * h1) + kp / e->A1 * u) * e->dt;
            const double n2 = h2 + (a1 / e->A2 * sqrt(2 * e->G * h1) - a2 / e->A2 * sqrt(2 * e->G * h2)) * e->dt;
            h1 = clipd(n1, lo, hi);
            h2 = clipd(n2, lo, hi);
        }
        double z1n, z2n;
        if (noise) { z1n = noise[2 * i]; z2n = noise[2 * i + 1]; }
        else {
            double ua, ub;
            philox_pair(e->seed, e->env_offset + (uint32_t)i, (uint32_t)e->episode[i], (uint32_t)e->t[i],
                        STREAM_NOISE, &ua, &ub);
            const double rad = sqrt(-2.0 * log(1.0 - ua)), ang = 6.283185307179586476925286766559 * ub;
            z1n = e->noise_scale * (rad * cos(ang));
            z2n = e->noise_scale * (rad * sin(ang));
        }
        h1 += z1n; h2 += z2n;                                          /* :810-811 */
        h1 = clipd(h1, lo, hi); h2 = clipd(h2, lo, hi);                /* :812-813 */
        e->h1[i] = h1; e->h2[i] = h2;
        double rew = reward_of(e->reward_type, h2, e->r[i], e->thr);
        if (e->reward_type != 2) rew = rew * e->z1;                    /* :506,508 */
        const int d = !(e->t[i] < e->max_steps);                       /* :816-821 */
        if (e->num_stack > 0) {                                        /* :1143-1144: deque append */
            double* f = e->frames + (size_t)i * e->num_stack * 3;
            memmove(f, f + 3, sizeof(double) * 3 * (size_t)(e->num_stack - 1));
            double* last = f + 3 * (size_t)(e->num_stack - 1);
            last[0] = h1; last[1] = h2; last[2] = e->r[i];
        } else {
            const double I_raw = e->I[i] + (e->r[i] - h2);             /* :822-823 */
            rew += -e->integral_punish * fabs(I_raw);                  /* :824 */
            e->I[i] = clipd(I_raw, -e->integral_max, e->integral_max); /* :825 */
        }
        wt_write_obs(e, i, obs, obs64);
        if (reward) reward[i] = rew;
        if (done) done[i] = (uint8_t)d;
        if (dmask && d) { dmask[i] = 1; any_done = 1; }
    }
    if (any_done) oracle_wt_reset(e, dmask, reset_draws, obs);
    free(dmask);
}

/* ------------------------------------------------------------------------------------------------
 * Residual action composition (elegantrl/agent_residual.py:61): np.tanh(a_pre) of a float32 = the float64 tanh
 * rounded once to float32 (implementation-independent up to ~1e-9 probability, unlike tanhf whose last ulp
 * differs between libm / ocml / numpy); the prior term state_f32 @ priorK_f64 in float64, summed in float64.
 * ---------------------------------------------------------------------------------------------- */
ORACLE_API void oracle_residual_action(int n, int D, const float* a_pre, const float* obs, const double* priorK,
                                       double* action) {
#pragma omp parallel for schedule(static) if (n >= 512)
    for (int i = 0; i < n; ++i) {
        double dot = 0.0;
        for (int j = 0; j < D; ++j) dot += (double)obs[(size_t)D * i + j] * priorK[j];
        action[i] = (double)(float)tanh((double)a_pre[i]) + dot;
    }
}

/* ------------------------------------------------------------------------------------------------
 * ElegantRL's reward-sum / advantage recursion (elegantrl/agent.py:685-708 GAE, :666-683 plain),
 * float32 like the reference tensors, per lane over [T][N] (time-major) storage.  Normalisation
 * (adv - mean)/(std + 1e-5) is left to the caller (it is buffer-global).
 * ---------------------------------------------------------------------------------------------- */
ORACLE_API void oracle_gae(int T, int N, const float* reward, const float* mask, const float* value, float lambda,
                           int use_gae, float* r_sum, float* adv) {
#pragma omp parallel for schedule(static) if (N >= 512)
    for (int n = 0; n < N; ++n) {
        float pre_r = 0.f, pre_a = 0.f;
        for (int t = T - 1; t >= 0; --t) {
            const size_t k = (size_t)t * N + n;
            r_sum[k] = reward[k] + mask[k] * pre_r;       /* :701 */
            pre_r = r_sum[k];
            if (use_gae) {
                adv[k] = reward[k] + mask[k] * (pre_a - value[k]); /* :704 */
                pre_a = value[k] + adv[k] * lambda;               /* :705 */
            } else {
                adv[k] = r_sum[k] - mask[k] * value[k];           /* :681 */
            }
        }
    }
}

/* ------------------------------------------------------------------------------------------------
 * MLP forwards on float32 tensors.  Each dot product is accumulated in double and rounded to float32
 * once, so the result is within half an ulp of the exact sum: a tighter checker than any particular
 * float32 summation order (torch CPU, rocBLAS or MFMA).  Layers are row-major [out][in] like nn.Linear.weight.
 * act: 0 none, 1 relu, 2 tanh.
 * ---------------------------------------------------------------------------------------------- */
static void dense(int M, int K, int Nout, const float* x, int ldx, const float* W, const float* b, int act,
                  float* y, int ldy) {
#pragma omp parallel for schedule(static) if (M >= 64)
    for (int m = 0; m < M; ++m)
        for (int o = 0; o < Nout; ++o) {
            double acc = b ? b[o] : 0.0;
            for (int k = 0; k < K; ++k) acc += (double)x[(size_t)m * ldx + k] * W[(size_t)o * K + k];
            float v = (float)acc;
            if (act == 1) v = v > 0 ? v : 0;
            else if (act == 2) v = tanhf(v);
            y[(size_t)m * ldy + o] = v;
        }
}

/* CriticAdv effective net (elegantrl/net.py:274-277): D -> md ReLU -> md ReLU -> md ReLU -> 1 */
ORACLE_API void oracle_critic_forward(int M, int D, int md, const float* x, const float* W0, const float* b0,
                                      const float* W1, const float* b1, const float* W2, const float* b2,
                                      const float* W3, const float* b3, float* v) {
    float* h0 = (float*)malloc(sizeof(float) * (size_t)M * md);
    float* h1 = (float*)malloc(sizeof(float) * (size_t)M * md);
    dense(M, D, md, x, D, W0, b0, 1, h0, md);
    dense(M, md, md, h0, md, W1, b1, 1, h1, md);
    dense(M, md, md, h1, md, W2, b2, 1, h0, md);
    dense(M, md, 1, h0, md, W3, b3, 0, v, 1);
    free(h0); free(h1);
}

/* ActorResidualIntegratorModularPPO mean (elegantrl/net_residual.py:153-160,172-176):
 * other_net: (D-Di) -> md tanh -> md/2 tanh ; integrator_net: Di -> md tanh -> md/2 tanh ;
 * net: md -> md tanh -> 1.  Output a_avg (pre-tanh, without the prior term). */
ORACLE_API void oracle_modular_actor_mean(int M, int D, int Di, int md, const float* x, const float* Wo0,
                                          const float* bo0, const float* Wo1, const float* bo1, const float* Wi0,
                                          const float* bi0, const float* Wi1, const float* bi1, const float* Wn0,
                                          const float* bn0, const float* Wn1, const float* bn1, float* a_avg) {
    const int Do = D - Di, half = md / 2;
    float* t0 = (float*)malloc(sizeof(float) * (size_t)M * md);
    float* cat = (float*)malloc(sizeof(float) * (size_t)M * 2 * half);
    dense(M, Do, md, x, D, Wo0, bo0, 2, t0, md);
    dense(M, md, half, t0, md, Wo1, bo1, 2, cat, 2 * half);
    dense(M, Di, md, x + Do, D, Wi0, bi0, 2, t0, md);
    dense(M, md, half, t0, md, Wi1, bi1, 2, cat + half, 2 * half);
    dense(M, 2 * half, md, cat, 2 * half, Wn0, bn0, 2, t0, md);
    dense(M, md, 1, t0, md, Wn1, bn1, 0, a_avg, 1);
    free(t0); free(cat);
}

/* Plain 4-layer tanh actor mean (ActorResidualPPO / ActorPPO, net_residual.py:19-22): D -> md -> md -> md -> 1 */
ORACLE_API void oracle_plain_actor_mean(int M, int D, int md, const float* x, const float* W0, const float* b0,
                                        const float* W1, const float* b1, const float* W2, const float* b2,
                                        const float* W3, const float* b3, float* a_avg) {
    float* h0 = (float*)malloc(sizeof(float) * (size_t)M * md);
    float* h1 = (float*)malloc(sizeof(float) * (size_t)M * md);
    dense(M, D, md, x, D, W0, b0, 2, h0, md);
    dense(M, md, md, h0, md, W1, b1, 2, h1, md);
    dense(M, md, md, h1, md, W2, b2, 2, h0, md);
    dense(M, md, 1, h0, md, W3, b3, 0, a_avg, 1);
    free(h0); free(h1);
}
