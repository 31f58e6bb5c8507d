/*
 * pime_cpu.h -- C ABI of libpime_cpu.so, the CPU twins of the env step / reset entry points of pime_hip.h (SURVEY.md section 8(b):
 * "`*_cpu` twins of step / reset operating on host pointers (the CPU baseline)").
 *
 * NOT the product path and never a fallback: libpime_hip.so refuses to run without a gfx950 device, and the pime_amd package does
 * not load this library.  It is the product's OWN lane arithmetic (csrc/env_device.hpp, the functions the HIP kernels call)
 * compiled for the host, so that bench.py's `cpu_baseline` and the parity tests can run exactly that arithmetic on the box's cores.
 * float64 state only (the reference's precision: PIME_STATE_F64 semantics); pH, and the water tank with the Integrator observation.
 * Each entry point mirrors its pime_hip.h namesake -- same configuration struct, same draws / noise injection, same auto-reset --
 * with host pointers instead of device pointers and a `threads` count instead of a stream (lanes are independent: results do not
 * depend on it).  replaces: gym_control/envs/ph.py:320-348,409-445 and nonlinear_watertank.py:800-826,890-939, per lane.
 */
#ifndef PIME_CPU_H
#define PIME_CPU_H

#include "pime_hip.h"

#ifdef __cplusplus
extern "C" {
#endif
#pragma GCC visibility push(default)

typedef struct pime_env_cpu pime_env_cpu;

const char* pime_cpu_last_error(void);
/* cfg as pime_env_create (device_id ignored); state_mode must be PIME_STATE_F64, num_stack 0 */
pime_env_cpu* pime_env_create_cpu(const pime_env_cfg* cfg);
void pime_env_destroy_cpu(pime_env_cpu* env);
/* mask [host] uint8[N] or NULL; draws [host] float64[N, 4 | 6] or NULL (Philox); obs [host] float32[N, obs_dim] */
int pime_env_reset_cpu(pime_env_cpu* env, const uint8_t* mask, const double* draws, float* obs, int32_t threads);
/* action [host] float64[N] env actions; noise [host] float64[N, 2] or NULL (water tank: injected process noise) */
int pime_env_step_cpu(pime_env_cpu* env, const double* action, const double* noise, int32_t auto_reset, const double* reset_draws,
                      float* obs, float* reward, uint8_t* done, int32_t threads);
/* a_pre [host] float32[N] pre-tanh residual actions, obs_in [host] float32[N, obs_dim], priorK [host] float64[obs_dim]:
 * env action = tanh(a_pre) + obs_in @ priorK (agent_residual.py:61), as pime_env_step_residual */
int pime_env_step_residual_cpu(pime_env_cpu* env, const float* a_pre, const float* obs_in, const double* priorK, const double* noise,
                               int32_t auto_reset, const double* reset_draws, float* obs, float* reward, uint8_t* done,
                               int32_t threads);
/* field: enum pime_field; out [host] float64[N] */
int pime_env_read_field_cpu(pime_env_cpu* env, int32_t field, double* out);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif /* PIME_CPU_H */
