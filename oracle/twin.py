"""ctypes binding of libpime_cpu.so (include/pime_cpu.h): the CPU twin of the env entry points -- the product's own lane functions
(csrc/env_device.hpp) compiled for the host.  TEST INFRASTRUCTURE: imported by tests/ and bench.py's cpu_baseline only; the pime_amd
package never loads the library (it has no CPU path)."""
import ctypes as C
import os

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_PKG = os.path.join(_ROOT, "pime-robust-non-linear-set-point-control-with-reinforcement-learning_amd")
LIB_PATH = os.path.join(_PKG, "libpime_cpu.so")
_lib = None
_vp, _i32 = C.c_void_p, C.c_int32
EXPORTS = {
    "pime_cpu_last_error": (C.c_char_p, []),
    "pime_env_create_cpu": (_vp, [_vp]),
    "pime_env_destroy_cpu": (None, [_vp]),
    "pime_env_reset_cpu": (C.c_int, [_vp, _vp, _vp, _vp, _i32]),
    "pime_env_step_cpu": (C.c_int, [_vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _i32]),
    "pime_env_step_residual_cpu": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _i32]),
    "pime_env_read_field_cpu": (C.c_int, [_vp, _i32, _vp]),
}


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: build it with `make -C {_PKG}/csrc` (or __graft_entry__.build())")
        h = C.CDLL(LIB_PATH)
        for name, (res, args) in EXPORTS.items():
            fn = getattr(h, name)
            fn.restype, fn.argtypes = res, args
        _lib = h
    return _lib


def _p(a):
    return None if a is None else C.c_void_p(a.ctypes.data)


class TwinEnv:
    """N lanes of the pH ("ph") or Integrator water-tank ("wt") env on the host, float64 state, Philox or injected draws: the
    configuration is pime_env_cfg_default's (libpime_hip.so fills the struct; no GPU call), overridden by keyword."""

    def __init__(self, kind, n, table=None, seed=0, env_offset=0, threads=1, **cfg_over):
        import pime_amd.native as nt
        self.kind, self.n, self.threads = kind, int(n), int(threads)
        cfg = nt.EnvCfg()
        nt.check(nt.lib().pime_env_cfg_default(nt.ENV_PH if kind == "ph" else nt.ENV_WT, C.byref(cfg)))
        cfg.n_envs, cfg.state_mode, cfg.seed, cfg.env_offset = self.n, nt.STATE_F64, seed, env_offset
        for k, v in cfg_over.items():
            setattr(cfg, k, v)
        self._table = None
        if kind == "ph":
            self._table = np.ascontiguousarray(table if table is not None else nt.ph_table_build(), dtype=np.float64)
            cfg.ph_table = self._table.ctypes.data_as(C.POINTER(C.c_double))
            cfg.ph_table_len = len(self._table)
        self.obs_dim = 3 if kind == "ph" else 4
        self._prefix = "ph_" if kind == "ph" else "wt_"
        self._nt = nt
        self._h = C.c_void_p(lib().pime_env_create_cpu(C.byref(cfg)))
        if not self._h:
            raise RuntimeError("pime_env_create_cpu: " + lib().pime_cpu_last_error().decode())
        self.max_steps = int(cfg.max_steps)

    def _check(self, rc, what):
        if rc != 0:
            raise RuntimeError(f"{what} failed ({rc}): " + lib().pime_cpu_last_error().decode())

    def reset(self, mask=None, draws=None):
        obs = np.empty((self.n, self.obs_dim), dtype=np.float32)
        m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        d = None if draws is None else np.ascontiguousarray(draws, dtype=np.float64)
        self._check(lib().pime_env_reset_cpu(self._h, _p(m), _p(d), _p(obs), self.threads), "pime_env_reset_cpu")
        return obs

    def step(self, action, auto_reset=True, noise=None):
        a = np.ascontiguousarray(action, dtype=np.float64).reshape(-1)
        obs = np.empty((self.n, self.obs_dim), dtype=np.float32)
        rew, done = np.empty(self.n, dtype=np.float32), np.empty(self.n, dtype=np.uint8)
        nz = None if noise is None else np.ascontiguousarray(noise, dtype=np.float64)
        self._check(lib().pime_env_step_cpu(self._h, _p(a), _p(nz), int(auto_reset), None, _p(obs), _p(rew), _p(done), self.threads),
                    "pime_env_step_cpu")
        return obs, rew, done.astype(bool)

    def step_residual(self, a_pre, obs_in, priorK, auto_reset=True):
        a = np.ascontiguousarray(a_pre, dtype=np.float32).reshape(-1)
        oi = np.ascontiguousarray(obs_in, dtype=np.float32)
        k = np.ascontiguousarray(priorK, dtype=np.float64).reshape(-1)
        obs = np.empty((self.n, self.obs_dim), dtype=np.float32)
        rew, done = np.empty(self.n, dtype=np.float32), np.empty(self.n, dtype=np.uint8)
        self._check(lib().pime_env_step_residual_cpu(self._h, _p(a), _p(oi), _p(k), None, int(auto_reset), None, _p(obs), _p(rew),
                                                     _p(done), self.threads), "pime_env_step_residual_cpu")
        return obs, rew, done.astype(bool)

    def get(self, name):
        out = np.empty(self.n, dtype=np.float64)
        self._check(lib().pime_env_read_field_cpu(self._h, self._nt.FIELD[self._prefix + name], _p(out)), f"read_field({name})")
        return out

    def close(self):
        if getattr(self, "_h", None):
            lib().pime_env_destroy_cpu(self._h)
            self._h = None

    def __del__(self):
        self.close()
