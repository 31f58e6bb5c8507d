/*
 * pime_hip.h -- C ABI of libpime_hip.so, the MI355X (gfx950) implementation of the vectorised
 * set-point-control hot path.
 *
 * The reference (ruoqizzz/PIME-..., 100 % Python) has no FFI: its boundary for this path is the Python duck
 * type consumed by elegantrl/env.py:PreprocessEnv and the agents.  Each entry point below names the reference
 * interface it replaces (paths relative to the reference root); INTEGRATION.md shows the ctypes stub a
 * reference maintainer would add to bind them.
 *
 * Conventions
 *  - plain C types only; every buffer is caller-owned.  Pointers marked [dev] are device pointers on the
 *    handle's device (e.g. torch tensor .data_ptr()), [host] are host pointers.
 *  - every launch goes to the caller-supplied HIP stream (pime_stream = hipStream_t; pass torch's current
 *    stream) and returns without synchronising, so calls are hipGraph-capturable.  Functions documented as
 *    "synchronous" (field I/O, create/destroy) synchronise that stream themselves.
 *  - return value: 0 = PIME_OK, negative = error class; pime_last_error() returns a thread-local message.
 *    No C++ exception crosses the ABI.
 *  - a handle is single-writer: one host thread, one stream at a time.  Different handles are independent.
 *  - there is NO CPU fallback: without a usable gfx950 device pime_env_create() fails with PIME_ERR_DEVICE.
 */
#ifndef PIME_HIP_H
#define PIME_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
#pragma GCC visibility push(default) /* the library is built with -fvisibility=hidden */

#define PIME_ABI_VERSION 19

typedef struct pime_env pime_env; /* opaque: SoA env state + titration LUT replica, resident in HBM */
typedef void* pime_stream;        /* hipStream_t */

enum pime_status {
    PIME_OK = 0,
    PIME_ERR_ARG = -1,    /* bad argument / shape / enum */
    PIME_ERR_DEVICE = -2, /* no usable gfx950 device, or HIP runtime error */
    PIME_ERR_ALLOC = -3,
    PIME_ERR_STATE = -4   /* call sequence error (e.g. step before reset) */
};

enum pime_kind { PIME_ENV_PH = 0, PIME_ENV_WT = 1 };
/* PIME_STATE_F64: every state word float64 (reference precision).  PIME_STATE_MIXED: float32 state; the pH
 * reaction-invariant x and the discretised plant (A,B,C) stay float64 so the LUT index is the reference's.
 * PIME_STATE_MIXED16 (BASELINE.json config 5 "fp16 state" = SURVEY.md §8(d) cfg 5): as MIXED with the integrated error I STORED
 * as IEEE binary16 and, through the *_h entry points, observations and rewards written as binary16 (48 instead of 68 bytes of
 * algorithmic traffic per pH env-step); all arithmetic stays float32 / float64.  What it costs in accuracy is stated with
 * pime_env_step_h. */
enum pime_state_mode { PIME_STATE_F64 = 0, PIME_STATE_MIXED = 1, PIME_STATE_MIXED16 = 2 };
enum pime_reward { PIME_REWARD_DISTANCE = 0, PIME_REWARD_SQUARE = 1, PIME_REWARD_SPARSE = 2 };
enum pime_dtype { PIME_F32 = 0, PIME_F64 = 1 };

/* state fields for pime_env_read_field / pime_env_write_field */
enum pime_field {
    /* pH  (gym_control/envs/ph.py attributes: state, integrator, r, y, dsys.A/B/C, qww_V, qc_V, _episode_steps) */
    PIME_PH_X = 0, PIME_PH_I = 1, PIME_PH_R = 2, PIME_PH_Y = 3, PIME_PH_A = 4, PIME_PH_B = 5, PIME_PH_C = 6,
    PIME_PH_QWW_V = 7, PIME_PH_QC_V = 8, PIME_PH_T = 9, PIME_PH_EPISODE = 10,
    /* water tank (nonlinear_watertank.py attributes: h1, h2, r, integrator, a1, a2, Kp, _episode_steps) */
    PIME_WT_H1 = 32, PIME_WT_H2 = 33, PIME_WT_R = 34, PIME_WT_I = 35, PIME_WT_A1 = 36, PIME_WT_A2 = 37,
    PIME_WT_KP = 38, PIME_WT_T = 39, PIME_WT_EPISODE = 40
};

/* Titration chemistry, gym_control/envs/ph.py:32-37 */
typedef struct pime_ph_chem {
    double kw, kchem, ka, MNaOH, MHA, MNH3;
} pime_ph_chem;

typedef struct pime_env_cfg {
    int32_t kind;            /* pime_kind */
    int32_t n_envs;          /* instances advanced per launch (this rank's slice) */
    int32_t device_id;       /* HIP device ordinal */
    int32_t state_mode;      /* pime_state_mode */
    int32_t max_steps;       /* pH: gym TimeLimit (gym_control/__init__.py:6) ; WT: max_step (:55) */
    int32_t reward_type;     /* pime_reward */
    int32_t integral_bound;  /* 1: clip I to +-integral_max (ph.py:341, nonlinear_watertank.py:825); 0: _NoBound (ph.py:470) */
    int32_t num_stack;       /* WT only. 0: Integrator obs [h1,h2,r,I]; S>=1: Stacking obs, 3*S floats (:1056-1208) */
    int32_t resample_every;  /* n: redraw the ensemble params on every n-th reset; 0 = never (if_reset_all False) */
    uint32_t env_offset;     /* global id of lane 0 (rank * n_envs under data-parallel sharding); Philox counter word */
    uint64_t seed;           /* Philox key */
    double integral_max, integral_punish, action_punish, action_change_punish, distance_threshold;
    double range_lo[3], range_hi[3]; /* ensemble ranges: pH (qww_V, qc_V, -) ; WT (a1, a2, Kp) */
    double init_lo[2], init_hi[2];   /* [0]: initial state range (pH x0 / WT h1,h2) ; [1]: goal r range */
    /* pH plant */
    double ph_sample_t, ph_u_low, ph_u_high, ph_table_scale; /* T=20 s; action map [0,1.5]; 1e5 = 1/MHCl step */
    const double* ph_table;  /* [host] ph_table_len float64 entries (pime_ph_table_build); copied at create */
    int32_t ph_table_len;
    /* water-tank plant */
    int32_t wt_n_discrete;
    double wt_A1, wt_A2, wt_G, wt_dt, wt_noise_scale, wt_z1, wt_pmax;
} pime_env_cfg;

/* -- library --------------------------------------------------------------------------------------------- */
int pime_abi_version(void);
const char* pime_last_error(void);
/* number of visible gfx950 devices (0 on a CPU-only host); never throws */
int pime_device_count(void);

/* -- titration LUT ----------------------------------------------------------------------------------------
 * replaces: PH1D.__init__ table loop, gym_control/envs/ph.py:72-84 (13 s of Python per gym.make).
 * Sequential warm-started Newton => host code, fp64, ~10 ms.  out: [host] n entries, entry i for
 * MHCl = i * mhcl_step.  chem == NULL selects ph.py:32-37. */
int pime_ph_table_build(const pime_ph_chem* chem, double mhcl_step, int32_t n, double* out);

/* -- env handle -------------------------------------------------------------------------------------------
 * pime_env_cfg_default fills the registered configuration of
 *   kind PH: 'PH1DChangingParamUniformGoalIntegrator-SqaureDistance-v35' (gym_control/__init__.py:3-14)
 *   kind WT: 'NonLinearWaterTankChangingParamUniformGoalIntegrator-SquareDistance-v2' (:50-69)
 * (ph_table left NULL: the caller supplies it). */
int pime_env_cfg_default(int32_t kind, pime_env_cfg* cfg);
/* replaces: gym.make(id) -> env constructor (ph.py:353-407, nonlinear_watertank.py:830-888). synchronous. */
pime_env* pime_env_create(const pime_env_cfg* cfg);
void pime_env_destroy(pime_env* env);
/* Stream capture and device frees.  hipFree under an open stream capture aborts the process, and Python finalises handles whenever
 * its garbage collector pleases.  Bracket every capture (torch.cuda.graph, hipStreamBeginCapture) with pime_capture_begin / _end:
 * while the depth is non-zero pime_env_destroy and
 * pime_oneshot_destroy park their device memory instead of freeing it; pime_capture_end (at depth 0), pime_env_create and the
 * destroy calls outside a capture drain the queue.  pime_deferred_releases: pointers parked right now. */
void pime_capture_begin(void);
void pime_capture_end(void);
void pime_capture_leave(void);   /* pime_capture_end without the drain (a finaliser inside a capture the library was not told about) */
int pime_deferred_releases(void);
int32_t pime_env_obs_dim(const pime_env* env);
int32_t pime_env_num_envs(const pime_env* env);
/* number of float64 draws one reset consumes per lane: pH 4 (qww_V, qc_V, x0, r), WT 6 (a1, a2, Kp, h1, h2, r) */
int32_t pime_env_reset_draw_width(const pime_env* env);

/* replaces: env.reset() -> reset_all()/reset_r(), ph.py:412-445 ; nonlinear_watertank.py:902-939,1166-1208.
 *   mask  [dev] uint8[N] or NULL (= all lanes)
 *   draws [dev] float64[N, width] final values in the reference's draw order (seed-for-seed replay of the
 *         MT19937 streams, generated host side) or NULL (= in-kernel Philox4x32-10)
 *   obs   [dev] float32[N, obs_dim] written for the reset lanes (float32 = PreprocessEnv cast, env.py:46) */
int pime_env_reset(pime_env* env, const uint8_t* mask, const double* draws, float* obs, pime_stream stream);

/* replaces: PreprocessEnv.step_type -> TimeLimit.step -> env.step(action), ph.py:320-348,448-478 ;
 * nonlinear_watertank.py:800-826,1118-1147.
 *   action       [dev] N env actions, float32 or float64 (action_dtype = pime_dtype)
 *   noise        [dev] WT: float64[N,2] already-scaled normals added to h1,h2 (:810-811) or NULL (= Philox)
 *   auto_reset   1: a lane that reports done is reset in the same launch and its obs row is the first
 *                observation of the next episode (vector-env convention; the reference calls reset() itself)
 *   reset_draws  [dev] as pime_env_reset's draws, consumed only by lanes that auto-reset; or NULL
 *   obs [dev] float32[N, obs_dim]; reward [dev] float32[N]; done [dev] uint8[N] */
int pime_env_step(pime_env* env, const void* action, int32_t action_dtype, const double* noise,
                  int32_t auto_reset, const double* reset_draws, float* obs, float* reward, uint8_t* done,
                  pime_stream stream);

/* The same step with the residual-policy action composition fused into the kernel prologue.
 * replaces: elegantrl/agent_residual.py:61  env.step(np.tanh(action) + state @ self.priorK)
 *   a_pre   [dev] float32[N] pre-tanh residual action (mean + sigma*eps)
 *   obs_in  [dev] float32[N, obs_dim] the observation the policy saw
 *   priorK  [host] float64[obs_dim] (= -env.K); copied into the launch arguments */
int pime_env_step_residual(pime_env* env, const float* a_pre, const float* obs_in, const double* priorK,
                           const double* noise, int32_t auto_reset, const double* reset_draws, float* obs,
                           float* reward, uint8_t* done, pime_stream stream);

/* The same three calls with binary16 observation / reward buffers ([dev] uint16_t = IEEE binary16 bits, torch.float16), for
 * handles in PIME_STATE_MIXED16 mode (PIME_ERR_ARG otherwise).  obs_in of the residual form is the binary16 observation the
 * policy saw.  Accuracy: every stored word is the float32 value rounded to nearest binary16 (relative 2^-11 = 4.9e-4: pH 11 ->
 * +-0.004, reward -64 -> +-0.03); the dynamics (x, the LUT index, h1, h2) are untouched, but the integrated error accumulates one
 * binary16 rounding per step (|I| <= 25: <= 0.008 per step, random-walk growth), and with it the prior PI controller's action. */
int pime_env_reset_h(pime_env* env, const uint8_t* mask, const double* draws, uint16_t* obs, pime_stream stream);
int pime_env_step_h(pime_env* env, const float* action, const double* noise, int32_t auto_reset, const double* reset_draws,
                    uint16_t* obs, uint16_t* reward, uint8_t* done, pime_stream stream);
int pime_env_step_residual_h(pime_env* env, const float* a_pre, const uint16_t* obs_in, const double* priorK,
                             const double* noise, int32_t auto_reset, const double* reset_draws, uint16_t* obs,
                             uint16_t* reward, uint8_t* done, pime_stream stream);

/* replaces: attribute reads/writes on the env object (set_state/set_r/set_params/get_changable_parameters/
 * reset_changable_parameters: ph.py:233-270, nonlinear_watertank.py:205-212,896-900).  synchronous.
 *   out/in [host] float64[N]; mask [host] uint8[N] or NULL.  Writing PIME_PH_QWW_V / PIME_PH_QC_V rebuilds the
 *   lane's ZOH plant (A,B,C) as update_system does (ph.py:114-121).  PIME_PH_Y is read-only. */
int pime_env_read_field(pime_env* env, int32_t field, double* out, pime_stream stream);
int pime_env_write_field(pime_env* env, int32_t field, const double* in, const uint8_t* mask, pime_stream stream);
/* run-time changes the evaluation harness makes on a live env (utils/test.py:1062-1064, utils/robust_test.py:17) */
int pime_env_set_punish(pime_env* env, double integral_punish, double action_punish, double action_change_punish);
int pime_env_set_max_steps(pime_env* env, int32_t max_steps);
int pime_env_set_resample_every(pime_env* env, int32_t n);
/* writes the current observation of every lane (what _get_observe() returns) */
int pime_env_observe(pime_env* env, float* obs, pime_stream stream);

/* -- trajectory post-processing ---------------------------------------------------------------------------
 * replaces: AgentPPO.compute_reward_gae / compute_reward_adv, elegantrl/agent.py:666-708 (a Python loop over
 * device scalars).  Time-major [T, N] float32 buffers, one thread per lane, reverse scan over T.
 * Outputs are un-normalised; the buffer-global (adv-mean)/(std+1e-5) stays with the caller (agent.py:707). */
int pime_gae_scan(const float* reward, const float* mask, const float* value, int32_t T, int32_t N, float lambda,
                  int32_t use_gae, float* r_sum, float* adv, pime_stream stream);

/* -- batched MLP forwards on the f32 matrix cores -----------------------------------------------------------
 * (v_mfma_f32_32x32x2_f32: exact float32, the reference's precision.)  Two steps so that the weight permutation
 * is paid once per weight version, not once per call: pime_mlp_pack re-lays the nn.Linear tensors into the image
 * the kernel keeps in LDS; pime_mlp_forward runs the whole net per 32-row tile with activations in registers.
 *
 *   kind PIME_MLP_CRITIC         replaces CriticAdv.forward over the buffer, elegantrl/agent.py:619-620 with
 *                                net.py:274-277 (D -> md ReLU -> md ReLU -> md ReLU -> 1)
 *                                params[8]  = net.0 W,b ; net.2 W,b ; net.4 W,b ; net.6 W,b
 *   kind PIME_MLP_PLAIN_ACTOR    replaces ActorResidualPPO / ActorPPO mean, net_residual.py:19-22,45-48
 *                                (same shape, Tanh); params[8] as above
 *   kind PIME_MLP_MODULAR_ACTOR  replaces ActorResidualIntegratorModularPPO mean, net_residual.py:153-160,172-176
 *                                params[12] = other_net.0 W,b ; other_net.2 W,b ; integrator_net.0 W,b ;
 *                                integrator_net.2 W,b ; net.0 W,b ; net.2 W,b ; Di = integrator_dim (trailing
 *                                columns of x)
 * All weights are [dev] float32 in nn.Linear layout ([out, in] row-major).  D <= 32.  Widths: 64 and 128 (every kind: the
 * whole net stays in the 160 KB LDS of a persistent workgroup, csrc/mlp_mfma.hip) and 256 (every kind -- the width
 * run_watertank_changing.sh trains: ResidualPPO on the 30-float Stacking10 observation :20-27, ResidualIntegratorModularPPO on the
 * Integrator observation :11-18 -- on 16-sample tiles of v_mfma_f32_16x16x4_f32 with the weight images streamed through LDS in
 * k-slices, csrc/mlp16.hip; the modular actor's md -> md/2 tower layers are rectangular chain layers there).
 * out [dev] float32[M] is the scalar head (value, or pre-tanh action mean without noise and prior term). */
enum pime_mlp_kind { PIME_MLP_CRITIC = 0, PIME_MLP_PLAIN_ACTOR = 1, PIME_MLP_MODULAR_ACTOR = 2 };
/* floats in the packed image (0 and an error message if the shape is unsupported) */
int64_t pime_mlp_packed_floats(int32_t kind, int32_t D, int32_t Di, int32_t md);
int pime_mlp_pack(int32_t kind, int32_t D, int32_t Di, int32_t md, const float* const* params, float* packed,
                  pime_stream stream);
int pime_mlp_forward(int32_t kind, const float* x, int32_t M, int32_t D, int32_t Di, int32_t md, const float* packed,
                     float* out, pime_stream stream);

/* -- fused PPO minibatch gradients ------------------------------------------------------------------------------
 * replaces, per optimizer step of AgentPPO.update_net (elegantrl/agent.py:629-657): the minibatch gather (:632-636),
 * compute_logprob of the residual actors (net_residual.py:48-54,182-190), the clipped surrogate + entropy proxy
 * (:637-645), CriticAdv forward + SmoothL1 (:648-649), the united loss (:652) and `obj_united.backward()` (:654-655)
 * -- i.e. everything between drawing the indices and `optimizer.step()` (pime_adam_step or torch.optim.Adam).
 * Three launches (csrc/ppo_fused.hip): one kernel per net that forms forward, loss gradient, backward chain and the
 * weight gradients of a 256-sample group per workgroup and stores them as a per-workgroup slab, and one reduction of the
 * slabs in a fixed order (bit-for-bit reproducible gradients) that also applies the critic scale.  Observations too wide
 * for that kernel's LDS map use the older net + dW + scale kernels (csrc/ppo_train.hip, float atomics).
 *
 * pime_ppo_net describes one net: the same (kind, D, Di, md, params) as pime_mlp_pack; `grads` are the .grad tensors in
 * the same order and are ACCUMULATED into (zero them first); img_fwd = pime_mlp_pack image,
 * img_bwd = pime_ppo_pack_bwd image (both must be re-packed after the weights change); workspace =
 * pime_ppo_workspace_floats(kind, B, md) floats.  action_dim must be 1. */
typedef struct pime_ppo_net {
    int32_t kind, D, Di, md;
    const float* const* params;   /* [host] array of [dev] pointers, W,b pairs */
    float* const* grads;          /* [host] array of [dev] pointers, same order */
    const float* a_std_log;       /* [dev] actor only: the [1,1] log-std parameter */
    float* g_a_std_log;           /* [dev] actor only: its gradient (accumulated) */
    const float* img_fwd;         /* [dev] */
    const float* img_bwd;         /* [dev] */
    float* workspace;             /* [dev] */
} pime_ppo_net;

typedef struct pime_ppo_batch {
    const float* state;           /* [dev] float32[L, D] trajectory observations */
    const float* action;          /* [dev] float32[L] pre-tanh actions that were taken */
    const float* logprob;         /* [dev] float32[L] their log-probabilities under the old policy */
    const float* adv;             /* [dev] float32[L] normalised advantages */
    const float* r_sum;           /* [dev] float32[L] reward sums (critic targets) */
    const int64_t* indices;       /* [dev] int64[B] minibatch rows (torch.randint) */
    int32_t B;
    int32_t flags;                /* PIME_PPO_* bits, 0 = accumulate into the gradient tensors */
    int64_t* index_row;           /* [dev] int64[1] or NULL.  Not NULL: `indices` is a table int64[rows, B], this call uses
                                   * row index_row[0] and then advances it by one -- the minibatches of a whole update can be
                                   * drawn by one torch.randint and a captured HIP graph replayed per optimizer step */
    float* dp_moments;            /* [dev] float32[4] or NULL.  Not NULL (data-parallel callers): the critic's gradient is left
                                   * UNSCALED and dp_moments[0..2] = sum r_sum[indices], sum of squares, B are written -- the caller
                                   * all-reduces them with the gradients (they sit behind the flat gradient buffer) and
                                   * pime_adam_step_dp applies 1 / (std over the UNION minibatch + 1e-5), which is what agent.py:652
                                   * means when the minibatch is spread over several ranks.  Slab nets only (PIME_ERR_ARG if the
                                   * critic takes the split pipeline) */
} pime_ppo_batch;
/* the call OVERWRITES the gradient tensors (and g_a_std_log) instead of adding to them: saves the caller's zeroing launch */
#define PIME_PPO_OVERWRITE_GRADS 1

/* floats of img_fwd / img_bwd of a pime_ppo_net (filled by pime_ppo_repack; img_bwd alone by pime_ppo_pack_bwd).  The forward
 * image of the gradient path is NOT always pime_mlp_pack's: nets of width 64 / 128 whose observation is too wide for the
 * LDS-resident gradient kernel (the stacked water tank, 30 floats) take the streamed 16-tile family here. */
int64_t pime_ppo_fwd_image_floats(int32_t kind, int32_t D, int32_t Di, int32_t md);
int64_t pime_ppo_bwd_image_floats(int32_t kind, int32_t D, int32_t Di, int32_t md);
/* The leading part of img_bwd that is a permutation of the parameters (what pime_ppo_image_map covers).  Equal to
 * pime_ppo_bwd_image_floats unless the process runs with PIME_GRAD_BF16X3=1 (opt-in; widths 128 / 256 on the 16-tile family --
 * PIME_MLP16=1 routes width 128 there): the streamed layers of the gradient kernels then run on bf16 matrix instructions with every
 * f32 operand split into hi + mid + lo bf16 pieces (six v_mfma_f32_16x16x32_bf16 per product block; error at the f32 rounding level,
 * not bit-equal to the f32 MFMA chain), and the three bf16 planes of those layers' weights follow the f32 image.  The library
 * re-splits them itself after every optimizer step it applies (pime_ppo_minibatch_step, pime_adam_step_images / _dp) and in
 * pime_ppo_repack / pime_ppo_pack_bwd. */
int64_t pime_ppo_bwd_image_f32_floats(int32_t kind, int32_t D, int32_t Di, int32_t md);
int64_t pime_ppo_workspace_floats(int32_t kind, int32_t B, int32_t md);
int pime_ppo_pack_bwd(int32_t kind, int32_t D, int32_t Di, int32_t md, const float* const* params, float* image,
                      pime_stream stream);
/* Re-packs img_fwd and img_bwd of both nets from their `params` in ONE launch (after every optimizer step; the four
 * separate pack launches cost ~20 us of a ~400 us step). */
int pime_ppo_repack(const pime_ppo_net* actor, const pime_ppo_net* critic, pime_stream stream);
/* critic_scale: [dev] float32[1], WRITTEN: 1 / (r_sum[indices].std() + 1e-5) with torch's unbiased std (agent.py:652);
 *               the critic's gradients are multiplied by it (in the slab reduction)
 * moments:      [dev] float64[2], WRITTEN: sum and sum of squares of the minibatch targets r_sum[indices]
 * loss_sums:    [dev] float32[6], ACCUMULATED over calls (zero them per update): [0] sum(-min(surr1,surr2)),
 *               [1] sum(exp(logp)*logp), [2] sum(smooth_l1), [3] sum of the calls' critic_scale, [4] sum over calls of
 *               (the call's smooth_l1 sum * its critic_scale) -- the critic part of the logged united loss, agent.py:652 --
 *               [5] scratch (value of [2] after the previous call) */
int pime_ppo_minibatch_grad(const pime_ppo_net* actor, const pime_ppo_net* critic, const pime_ppo_batch* batch,
                            float ratio_clip, float lambda_entropy, float* critic_scale, double* moments,
                            float* loss_sums, pime_stream stream);

/* The same call with the optimizer step fused into its last launch (the slab reduction): gradients of the minibatch, then
 * torch.optim.Adam's update of every parameter (pime_adam_step semantics) -- one launch and one launch boundary less per
 * optimizer step.  replaces agent.py:632-657 (gather ... backward ... optimizer.step()).  param / grad are the two flat tensors
 * every pime_ppo_net params / grads pointer is a view into, at equal offsets; gradients of parameters outside them (frozen ones
 * routed to a scratch tensor) are left without an update.  Not available when a net takes the split pipeline (PIME_ERR_ARG);
 * data-parallel callers, who all-reduce the gradients first, use pime_ppo_minibatch_grad + pime_adam_step. */
typedef struct pime_adam {
    float* param;       /* [dev] float32[n] */
    float* grad;        /* [dev] float32[n] */
    float* exp_avg;     /* [dev] float32[n] */
    float* exp_avg_sq;  /* [dev] float32[n] */
    float* step;        /* [dev] float32[2], as pime_adam_step */
    int64_t n;
    float lr, beta1, beta2, eps;
    const int32_t* image_map;   /* [dev] int32[2 n] from pime_ppo_image_map, or NULL.  Not NULL: the launch that updates a
                                 * parameter also writes its new value into the nets' img_fwd / img_bwd, so that NO
                                 * pime_ppo_repack is needed after the step (one launch less per optimizer step) */
    const float* dp_moments;    /* pime_adam_step_dp: [dev] float32[4], the rank-AVERAGED words pime_ppo_batch.dp_moments wrote */
    int64_t critic_offset;      /* pime_adam_step_dp: flat elements [critic_offset, n) are the critic's */
    int32_t dp_world;           /* pime_adam_step_dp: ranks the all-reduce averaged over */
} pime_adam;
/* For every element j of the flat parameter tensor: image_map[2j] = its position in its net's img_fwd, image_map[2j+1] = its
 * position in img_bwd, each with the net in bits 28..29 (0 critic, 1 actor; -1: not in that image, e.g. hidden-layer biases in the
 * transposed image).  The images are permutations of
 * the parameters (plus zero padding); the map is derived by running the library's own pack kernels on index-coded parameters, so it
 * follows whichever kernel family serves each net.  Allocates and frees scratch memory and synchronises the stream: call it once
 * per (nets, flat tensor), outside any graph capture.  PIME_ERR_ARG if a parameter straddles the flat tensor's ends. */
int pime_ppo_image_map(const pime_ppo_net* actor, const pime_ppo_net* critic, const float* flat_param, int64_t n,
                       int32_t* image_map, pime_stream stream);
int pime_ppo_minibatch_step(const pime_ppo_net* actor, const pime_ppo_net* critic, const pime_ppo_batch* batch,
                            float ratio_clip, float lambda_entropy, float* critic_scale, double* moments,
                            float* loss_sums, const pime_adam* opt, pime_stream stream);

/* -- fused rollout ---------------------------------------------------------------------------------------------
 * replaces: the whole per-step loop of AgentResidual*.explore_env for one episode chunk (elegantrl/agent_residual.py:
 * 52-69: select_action -> env.step(np.tanh(action) + state @ priorK) -> buffer.append_buffer), as ONE launch: policy
 * forward (packed actor image, pime_mlp_pack), exploration noise (in-kernel Philox stream 2 keyed by noise_seed,
 * counter (lane, noise_epoch, t)), residual composition, env step with in-kernel auto-reset, trajectory writes.
 * pH or water-tank (Integrator observation) env handle in PIME_STATE_MIXED mode with Philox draws;
 * kind = PIME_MLP_PLAIN_ACTOR | PIME_MLP_MODULAR_ACTOR.
 *   state  [dev] float32[n_steps+1, N, obs_dim]: slot 0 must hold the current observation on entry (pime_env_reset /
 *          pime_env_observe), slots 1..n_steps are written;  action (pre-tanh), noise, reward [dev] float32[n_steps, N];
 *          done [dev] uint8[n_steps, N];  priorK [host] float64[obs_dim]. */
/* 1 if pime_rollout serves this env handle with this actor (PIME_STATE_MIXED / MIXED16, pH or water-tank Integrator / Stacking
 * observation, width 64 / 128 with the actor image resident in LDS, width 256 with the images streamed -- float32 rows only),
 * else 0: the caller then steps the env launch by launch (pime_mlp_forward + pime_env_step_residual). */
int pime_rollout_supported(const pime_env* env, int32_t kind, int32_t md);
int pime_rollout(pime_env* env, int32_t kind, int32_t md, const float* packed_actor, const float* a_std_log,
                 const double* priorK, int32_t n_steps, uint64_t noise_seed, uint32_t noise_epoch, float* state,
                 float* action, float* noise, float* reward, uint8_t* done, pime_stream stream);
/* The same launch for a handle in PIME_STATE_MIXED16 mode (BASELINE.json config 5, "fp16 state"; pH or Integrator water tank):
 * state [dev] binary16[n_steps+1, N, obs_dim] and reward [dev] binary16[n_steps, N] (torch.float16); action and noise stay
 * float32 (they feed the log-probabilities of the update).  Semantics of a chain of pime_env_step_residual_h calls: the policy
 * and the prior term see the binary16 observation, the integrated error is rounded to binary16 after every step (its storage
 * format in this mode), x / h1 / h2 / the plant never are.  replaces agent_residual.py:52-69 as pime_rollout does. */
int pime_rollout_h(pime_env* env, int32_t kind, int32_t md, const float* packed_actor, const float* a_std_log,
                   const double* priorK, int32_t n_steps, uint64_t noise_seed, uint32_t noise_epoch, uint16_t* state,
                   float* action, float* noise, uint16_t* reward, uint8_t* done, pime_stream stream);

/* -- fused evaluation ---------------------------------------------------------------------------------------------
 * replaces: get_episode_return (elegantrl/run.py:600-619: max_step x [act(s) -> env.step], summed rewards) on every lane, and the
 * set-point step-response protocols (utils/test.py:1369-1407 pH, :209-349 water tank; utils/robust_test.py:4-46), as ONE launch:
 * n_steps steps of every lane under the DETERMINISTIC residual policy a_env = tanh(mean(s)) + s @ priorK -- no exploration noise,
 * no auto-reset, no per-step host round trip -- with the env state in registers.  Handles in PIME_STATE_MIXED or PIME_STATE_F64
 * mode (the latter reproduces the reference's float64 protocol records to 1e-11), pH or Integrator water tank, Philox draws.
 *   kind         PIME_MLP_PLAIN_ACTOR | PIME_MLP_MODULAR_ACTOR with packed_actor = its pime_mlp_pack image (width 64 / 128), or -1:
 *                the prior controller alone (get_linear_action, ph.py:227-231), packed_actor ignored
 *   seg_len      0: plain episode.  > 0: every seg_len steps (from step 0) a protocol segment starts: set-point r =
 *                setpoints[segment], integrated error 0, step counter 0, plant state kept -- what the protocols' `env.reset();
 *                set_state(last); set_r(r)` leaves (:1377-1381); setpoints [host] float64[n_setpoints <= 16]
 *   ret          [dev] float64[N] or NULL: += the launch's per-lane sum of (float32) rewards, as run.py:613 accumulates them
 *   trace        [dev] float64[n_steps, 6, N] or NULL, per step and lane: pH (y, r, I BEFORE the step; action, reward, x after it: the
 *                protocol's per-step records, utils/test.py:1388-1396), water tank (h1, h2, r, I after the step; reward; action)
 * Leaves every lane n_steps further; the caller resets the env before it rolls out again.
 * pime_rollout_eval_supported: 0 = not served; 1 = served (widths 64 / 128 in either state mode; width 256 -- the streamed rollout
 * kernel's evaluation mode, csrc/mlp16.hip -- in PIME_STATE_MIXED on the pH / Integrator observation, trace and schedule included);
 * 2 = returns and trace, but no set-point schedule (a Stacking observation at width 256: seg_len must be 0). */
int pime_rollout_eval_supported(const pime_env* env, int32_t kind, int32_t md);
int pime_rollout_eval(pime_env* env, int32_t kind, int32_t md, const float* packed_actor, const double* priorK, int32_t n_steps,
                      int32_t seg_len, const double* setpoints, int32_t n_setpoints, double* ret, double* trace,
                      pime_stream stream);

/* -- fused off-policy exploration -----------------------------------------------------------------------------------
 * replaces, per lock-step of the vectorised off-policy agents: AgentBase.explore_env's body (elegantrl/agent.py:54-70) with
 * AgentTD3.select_action (:300-306: a = (act(s) + N(0, explore_noise)).clamp(-1, 1)), env.step, and ReplayBuffer.append_buffer
 * (replay.py:290-300) -- as ONE launch for n_steps lock-steps of every lane: deterministic actor forward on the matrix cores,
 * clipped exploration noise (Philox stream 2, counter (lane, noise_epoch, t)), env action a_env = a + s @ priorK (priorK zeros:
 * plain TD3), env step with in-kernel auto-reset, transition written into the device ring.  The running episodes CONTINUE (no
 * reset in front).  Handle in PIME_STATE_MIXED mode, pH or Integrator water tank, Philox draws.
 *   packed_actor  pime_mlp_pack image of kind PIME_MLP_CRITIC built from the TD3 Actor's tensors (net.py:96-110 has CriticAdv's
 *                 shape and ReLUs; the kernel applies the tanh), width md = 64 / 128
 *   obs           [dev] float32[N, obs_dim]: in = the lanes' current observation, out = the observation after the last step
 *   ring_state    [dev] float32[slots, N, obs_dim]; ring_other [dev] float32[slots, N, 3] = (reward * reward_scale, mask = 0 if done
 *                 else gamma, action); slots slot0 .. slot0 + n_steps - 1 (mod slots) are written: the successor of (slot, lane) is
 *                 (slot + 1, lane), where replay.py:344-351 uses row i + 1 */
int pime_rollout_offpolicy_supported(const pime_env* env, int32_t md);
int pime_rollout_offpolicy(pime_env* env, int32_t md, const float* packed_actor, const double* priorK, float explore_noise,
                           float gamma, float reward_scale, int32_t n_steps, uint64_t noise_seed, uint32_t noise_epoch,
                           float* obs, float* ring_state, float* ring_other, int32_t slot0, int32_t slots, pime_stream stream);

/* -- fused TD3 optimizer step ---------------------------------------------------------------------------------------
 * replaces, per iteration of AgentTD3.update_net's loop (elegantrl/agent.py:314-331): get_obj_critic_raw (:361-370 -- the gather of
 * buffer.sample_batch's rows (replay.py:344-351), act_target.get_action with clamped smoothing noise (net.py:107-110),
 * min(cri_target.get_q1_q2), q_label, cri.get_q1_q2, SmoothL1 x 2), obj_critic.backward(), cri_optimizer.step(), the delayed
 * soft_update(cri_target) (:116-124), obj_actor = -cri_target(state, act(state)).mean() (:323-324), obj_actor.backward(),
 * act_optimizer.step() and the delayed soft_update(act_target): FOUR launches (csrc/td3_fused.hip) -- critic gradients
 * (one 16-sample tile per workgroup, its four waves splitting every layer's output features), slab reduction + Adam + soft
 * update of the critic, actor gradients (through the TARGET critic's first head, as the reference has it), the same for the actor.
 * Nets: Actor (net.py:96-110: D -> md ReLU -> md ReLU -> md ReLU -> 1, tanh) and CriticTwin (net.py:305-332: D+1 -> md ReLU -> md
 * ReLU, two linear heads), action_dim 1, D <= 7, md 64 | 128.  The weights are read where they live: every net is ONE flat
 * float32 tensor in nn.Module parameter order with each tensor starting on a multiple of 4 floats (pime_td3_param_offsets;
 * padding words zero) -- no packed images, nothing to re-pack after a step. */
typedef struct pime_td3_net {
    float* param;        /* [dev] float32[pime_td3_param_floats]: the online net */
    float* target;       /* [dev] same layout: the target net (soft-updated in place) */
    float* grad;         /* [dev] same layout, WRITTEN: this step's gradient (the reference's .grad after backward()) */
    float* exp_avg;      /* [dev] Adam state, same layout; zero at construction */
    float* exp_avg_sq;
    const float* step;   /* [dev] float32[1]: optimizer steps applied BEFORE table row 0; a call is Adam step number step[0] + row + 1.
                          * The CALLER adds the number of steps behind an update (one stream-ordered add per update_net) */
    float lr, beta1, beta2, eps;
} pime_td3_net;
typedef struct pime_td3_batch {
    const float* state;      /* [dev] float32[rows, D]: replay states (ReplayBuffer.buf_state / VecReplayBuffer.buf_state) */
    const float* other;      /* [dev] float32[rows, 3]: reward * scale, mask, action (buf_other) */
    const int64_t* idx;      /* [dev] int64[table_rows, B]: the sampled rows of every optimizer step of an update */
    const int64_t* nxt;      /* [dev] int64[table_rows, B]: their successors (idx + 1 in the flat ring, idx + N lanes in the vector ring) */
    const float* noise;      /* [dev] float32[table_rows, B] standard normal draws of the smoothing noise (torch.randn_like), or NULL:
                              * drawn in the kernel -- Philox4x32-10 keyed by noise_seed, counter (batch position, noise_epoch,
                              * table row, stream 3), Box-Muller cosine branch in float32 */
    int64_t row;             /* the table row of THIS optimizer step.  A launch argument, not device state: every node of a captured
                              * update graph carries its own row, so no launch has to advance a shared cursor */
    const int64_t* epoch;    /* [dev] int64[1] or NULL: added to noise_epoch (the host bumps it once per update, so that a captured
                              * HIP graph draws fresh noise in every replay) */
    int32_t B;
    uint64_t noise_seed;
    uint32_t noise_epoch;
    float policy_noise, noise_clip;   /* 0.2, 0.5 (agent.py:281, net.py:109) */
} pime_td3_batch;
int pime_td3_supported(int32_t D, int32_t action_dim, int32_t md);
/* which: 0 actor, 1 critic.  offsets [8]: float offset of every parameter tensor (W, b pairs in module order) */
int64_t pime_td3_param_floats(int32_t which, int32_t D, int32_t md);
int pime_td3_param_offsets(int32_t which, int32_t D, int32_t md, int32_t* offsets);
int64_t pime_td3_workspace_floats(int32_t D, int32_t md, int32_t B);
/* soft_mode: 0 no soft target update, 1 soft update, 2 soft update when row % update_freq == 0 (the reference's delayed update).
 * phases: bit 0 = critic gradients, bit 1 = critic apply (slab reduction, Adam, soft update), bit 2 = actor gradients, bit 3 = actor
 *         apply; 15 = the whole step on one stream.  The four launches may also be issued on TWO streams, the data dependencies
 *         being: actor gradients(row) read the state rows the critic gradients(row) gathered (kept per row parity in the workspace)
 *         and, on a soft row, the critic target the critic apply(row) wrote; critic gradients(row + 1) read the critic the critic
 *         apply(row) wrote and, behind a soft row, the actor target the actor apply(row) wrote.  So on rows without a soft update
 *         the critic apply may run beside the actor gradients, and the actor apply beside the next row's critic gradients
 *         (AgentTD3._update_fused does that: parallel branches of the update's HIP graph).
 *         Data-parallel callers split the two apply launches around their all-reduce (mean) of the net's `grad` tensor: 16 = critic slab
 *         reduction ONLY (grad = this rank's gradient, loss words), 32 = critic Adam (+ soft update) FROM `grad` (no slabs read); 64 / 128
 *         the same for the actor.  One optimizer step on G ranks: phases 1|16, all-reduce critic grad, phases 32|4|64, all-reduce actor
 *         grad, phases 128 (order inside a call: 1, 2|16, 32, 4, 8|64, 128; 2 with 16 or 8 with 64 is PIME_ERR_ARG).
 * loss [dev] float32[4] or NULL: [0] += obj_actor, [1] += obj_critic of this step (zero them per update), [2], [3] = this step's values.
 * workspace [dev] float32[pime_td3_workspace_floats]. */
int pime_td3_step(int32_t D, int32_t md, const pime_td3_net* actor, const pime_td3_net* critic, const pime_td3_batch* batch,
                  float tau, int32_t update_freq, int32_t soft_mode, int32_t phases, float* workspace, float* loss,
                  pime_stream stream);

/* replaces: self.optimizer.step() of the single Adam over both nets (elegantrl/agent.py:565-566,656-657; no weight
 * decay, no amsgrad) when every parameter lives in ONE flat tensor.  All [dev] float32[n]; step [dev] float32[2], zeroed
 * by the caller at construction: step[0] is the step counter, incremented by the call on the device (so the launch can be
 * replayed from a HIP graph), step[1] is this optimizer's arrival counter (scratch).  One optimizer = one stream at a time;
 * different optimizers are independent. */
int pime_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1,
                   float beta2, float eps, float* step, pime_stream stream);
/* pime_adam_step for the nets of a PPO agent, with opt->image_map set: every new parameter value is also written into the nets'
 * packed images, so no pime_ppo_repack follows.  For data-parallel callers, whose all-reduce sits between pime_ppo_minibatch_grad
 * and the optimizer step (single-GPU callers get the same from pime_ppo_minibatch_step in one launch less). */
int pime_adam_step_images(const pime_adam* opt, const pime_ppo_net* actor, const pime_ppo_net* critic, pime_stream stream);

/* The optimizer step of a data-parallel rank, behind the all-reduce (AVG) of [flat gradients | dp_moments]: the critic's elements
 * grad[critic_offset .. n) are first multiplied by 1 / (std + 1e-5), std = torch's unbiased std of the dp_world * B targets of the
 * UNION minibatch, recovered from the averaged moments (agent.py:652: `obj_critic / (r_sum.std() + 1e-5)`; actor and critic
 * parameters are disjoint and the united loss is linear in that factor, so scaling the averaged critic gradient is exact), written
 * back into grad, then torch.optim.Adam as pime_adam_step.  image_map may be NULL (then pime_ppo_repack follows). */
int pime_adam_step_dp(const pime_adam* opt, const pime_ppo_net* actor, const pime_ppo_net* critic, pime_stream stream);

/* -- one-shot all-reduce of the flat gradient buffer over peer-mapped memory ---------------------------------------------
 * replaces: nothing in the reference (one process; elegantrl/run.py:232-247 is an unused mp.Pipe) -- it is the hand-written
 * alternative to the RCCL all-reduce of SURVEY.md section 8(e) for the 270 KB gradient message: every rank writes its vector into
 * every peer's inbox (hipIpc-mapped, one write per xGMI link, all links at once), raises a flag, waits for its peers' flags and
 * sums the rows in rank order: one latency step, bit-identical results on every rank.  csrc/allreduce.hip has the protocol.
 * Life cycle, per rank (one process per GPU): create -> export the 64-byte handle -> exchange the handles of all ranks (any
 * transport: torch.distributed all_gather) -> connect -> allreduce_mean per optimizer step (one launch on the caller's stream,
 * HIP-graph capturable) -> destroy.  world <= 8.  A peer that never arrives does not hang the device: the kernel gives up after
 * ~2 s and pime_oneshot_status() returns non-zero -- which the caller must treat as fatal. */
typedef struct pime_oneshot pime_oneshot;
pime_oneshot* pime_oneshot_create(int32_t rank, int32_t world, int64_t n_floats, int32_t device);
int pime_oneshot_export(pime_oneshot* h, void* handle_out /* [host] 64 bytes */);
int pime_oneshot_connect(pime_oneshot* h, const void* handles /* [host] world x 64 bytes, rank order */);
/* data [dev] float32[n_floats]: replaced by the mean over the ranks */
int pime_oneshot_allreduce_mean(pime_oneshot* h, float* data, pime_stream stream);
/* 0: every call so far completed.  Non-zero: a peer's rows did not arrive within ~2 s (1) or the local grid barrier timed out (2) --
 * the launch then left `data` partly or wholly UN-averaged: treat it as fatal (the replicas have diverged); the Python side checks it
 * at the end of every update and raises on every rank.  Synchronises the device. */
int pime_oneshot_status(pime_oneshot* h);
/* info [host] int32[5]: [0] 1 = the region is fine-grained memory (required between DIFFERENT devices), 0 = the runtime fell back to
 * coarse-grained memory (valid only between processes sharing one device); [1] device ordinal; [2..4] PCI domain / bus / device. */
int pime_oneshot_info(pime_oneshot* h, int32_t* info);
void pime_oneshot_destroy(pime_oneshot* h);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif /* PIME_HIP_H */
