"""ctypes binding of oracle/libpime_oracle.so (built on demand with gcc).  Test infrastructure only."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libpime_oracle.so")
_lib = None

_dp = C.POINTER(C.c_double)
_fp = C.POINTER(C.c_float)
_bp = C.POINTER(C.c_uint8)


def build(force=False):
    src = os.path.join(_HERE, "pime_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "libpime_oracle.so"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.oracle_ph_create.restype = C.c_void_p
        _lib.oracle_ph_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _dp, C.c_int, C.c_uint64,
                                          C.c_uint32]
        _lib.oracle_wt_create.restype = C.c_void_p
        _lib.oracle_wt_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_uint64,
                                          C.c_uint32]
        _lib.oracle_wt_obs_dim.argtypes = [C.c_void_p]
    return _lib


def _d(a):
    return None if a is None else a.ctypes.data_as(_dp)


def _f(a):
    return None if a is None else a.ctypes.data_as(_fp)


def _b(a):
    return None if a is None else a.ctypes.data_as(_bp)


def _c64(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float64)


def _c32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


PH_CHEM = dict(kw=1e-14, kchem=5.6e-10, ka=0.5e-5, MNaOH=0.01, MHA=0.005, MNH3=0.01)  # ph.py:32-37


def ph_table(n=100000, step=1e-5, **chem):
    c = dict(PH_CHEM, **chem)
    out = np.empty(n, dtype=np.float64)
    lib().oracle_ph_table(C.c_int(n), C.c_double(step), C.c_double(c["kw"]), C.c_double(c["kchem"]),
                          C.c_double(c["ka"]), C.c_double(c["MNaOH"]), C.c_double(c["MHA"]), C.c_double(c["MNH3"]),
                          _d(out))
    return out


def ph_zoh(qww_V, qc_V, sample_t=20.0):
    a, b, c = C.c_double(), C.c_double(), C.c_double()
    lib().oracle_ph_zoh(C.c_double(qww_V), C.c_double(qc_V), C.c_double(sample_t), C.byref(a), C.byref(b),
                        C.byref(c))
    return a.value, b.value, c.value


def philox4x32_10(ctr, key):
    ctr = np.asarray(ctr, dtype=np.uint32)
    key = np.asarray(key, dtype=np.uint32)
    out = np.empty(4, dtype=np.uint32)
    lib().oracle_philox4x32_10(ctr.ctypes.data_as(C.c_void_p), key.ctypes.data_as(C.c_void_p),
                               out.ctypes.data_as(C.c_void_p))
    return out


def philox_uniform_pair(seed, env, episode, slot, stream):
    out = np.empty(2, dtype=np.float64)
    lib().oracle_philox_uniform_pair(C.c_uint64(seed), C.c_uint32(env), C.c_uint32(episode), C.c_uint32(slot),
                                     C.c_uint32(stream), _d(out))
    return out


def set_threads(n=0):
    """OpenMP threads of the lane / row loops (0 = leave unchanged); returns the current maximum."""
    return int(lib().oracle_set_threads(int(n)))


def explore_noise(seed, env_offset, n, epoch, t):
    """float32[n]: the exploration noise the fused rollout kernel draws for lanes env_offset..+n at step t of rollout
    `epoch` (Philox stream 2 + Box-Muller, csrc/rollout.hip)."""
    eps = np.empty(n, dtype=np.float32)
    lib().oracle_explore_noise(C.c_uint64(seed), C.c_uint32(env_offset), C.c_int(n), C.c_uint32(epoch), C.c_uint32(t),
                               _f(eps))
    return eps


REWARD = {"distance": 0, "square_distance": 1, "sparse": 2}


class OraclePH:
    """N independent reference-semantics pH envs (fp64)."""
    FIELDS = dict(x=0, I=1, r=2, y=3, A=4, B=5, C=6, qww_V=7, qc_V=8, t=9, episode=10)
    obs_dim = 3

    def __init__(self, n, table, max_steps=50, reward_type="square_distance", integral_bound=True,
                 resample_every=1, seed=0, env_offset=0):
        self.n = n
        self.table = np.ascontiguousarray(table, dtype=np.float64)  # keep alive: the C side borrows it
        self._h = C.c_void_p(lib().oracle_ph_create(n, max_steps, REWARD[reward_type], int(integral_bound),
                                                    resample_every, _d(self.table), len(self.table), seed,
                                                    env_offset))

    def __del__(self):
        if getattr(self, "_h", None) and lib is not None and _lib is not None:
            _lib.oracle_ph_destroy(self._h)
            self._h = None

    def set_ranges(self, qww_V, qc_V):
        lib().oracle_ph_set_ranges(self._h, *(C.c_double(v) for v in (*qww_V, *qc_V)))

    def set_punish(self, integral=0.0, action=0.0, action_change=0.0):
        lib().oracle_ph_set_punish(self._h, C.c_double(integral), C.c_double(action), C.c_double(action_change))

    def get(self, field):
        out = np.empty(self.n)
        lib().oracle_ph_get(self._h, self.FIELDS[field], _d(out))
        return out

    def set(self, field, values):
        v = np.ascontiguousarray(np.broadcast_to(np.asarray(values, dtype=np.float64), (self.n,)))
        lib().oracle_ph_set(self._h, self.FIELDS[field], _d(v))

    def reset(self, mask=None, draws=None):
        obs = np.zeros((self.n, 3), dtype=np.float32)
        m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        dr = _c64(draws)
        lib().oracle_ph_reset(self._h, _b(m), _d(dr), _f(obs))
        return obs

    def step(self, action, auto_reset=False, reset_draws=None):
        a = _c64(np.broadcast_to(np.asarray(action, dtype=np.float64).reshape(-1), (self.n,)))
        obs = np.empty((self.n, 3), dtype=np.float32)
        obs64 = np.empty((self.n, 3))
        rew = np.empty(self.n)
        done = np.empty(self.n, dtype=np.uint8)
        dr = _c64(reset_draws)
        lib().oracle_ph_step(self._h, _d(a), int(auto_reset), _d(dr), _f(obs), _d(obs64), _d(rew), _b(done))
        return obs, obs64, rew, done.astype(bool)


class OracleWT:
    """N independent reference-semantics water-tank envs (fp64). num_stack=0: Integrator obs [h1,h2,r,I]."""
    FIELDS = dict(h1=0, h2=1, r=2, I=3, a1=4, a2=5, Kp=6, t=7, episode=8)

    def __init__(self, n, max_steps=200, reward_type="distance", num_stack=0, resample_every=1, noise_scale=0.01,
                 seed=0, env_offset=0):
        self.n = n
        self._h = C.c_void_p(lib().oracle_wt_create(n, max_steps, REWARD[reward_type], num_stack, resample_every,
                                                    noise_scale, seed, env_offset))
        self.obs_dim = lib().oracle_wt_obs_dim(self._h)

    def __del__(self):
        if getattr(self, "_h", None) and lib is not None and _lib is not None:
            _lib.oracle_wt_destroy(self._h)
            self._h = None

    def set_ranges(self, a1, a2, Kp):
        lib().oracle_wt_set_ranges(self._h, *(C.c_double(v) for v in (*a1, *a2, *Kp)))

    def set_punish(self, integral=0.0):
        lib().oracle_wt_set_punish(self._h, C.c_double(integral))

    def set_max_steps(self, n):
        lib().oracle_wt_set_max_steps(self._h, int(n))

    def get(self, field):
        out = np.empty(self.n)
        lib().oracle_wt_get(self._h, self.FIELDS[field], _d(out))
        return out

    def set(self, field, values):
        v = np.ascontiguousarray(np.broadcast_to(np.asarray(values, dtype=np.float64), (self.n,)))
        lib().oracle_wt_set(self._h, self.FIELDS[field], _d(v))

    def reset(self, mask=None, draws=None):
        obs = np.zeros((self.n, self.obs_dim), dtype=np.float32)
        m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        dr = _c64(draws)
        lib().oracle_wt_reset(self._h, _b(m), _d(dr), _f(obs))
        return obs

    def step(self, action, noise=None, auto_reset=False, reset_draws=None):
        a = _c64(np.broadcast_to(np.asarray(action, dtype=np.float64).reshape(-1), (self.n,)))
        obs = np.empty((self.n, self.obs_dim), dtype=np.float32)
        obs64 = np.empty((self.n, self.obs_dim))
        rew = np.empty(self.n)
        done = np.empty(self.n, dtype=np.uint8)
        nz, dr = _c64(noise), _c64(reset_draws)
        lib().oracle_wt_step(self._h, _d(a), _d(nz), int(auto_reset), _d(dr), _f(obs), _d(obs64), _d(rew),
                             _b(done))
        return obs, obs64, rew, done.astype(bool)


def residual_action(a_pre, obs, priorK):
    a_pre = _c32(np.asarray(a_pre).reshape(-1))
    obs = _c32(obs)
    n, D = obs.shape
    k = _c64(np.asarray(priorK).reshape(-1))
    out = np.empty(n)
    lib().oracle_residual_action(n, D, _f(a_pre), _f(obs), _d(k), _d(out))
    return out


def gae(reward, mask, value, lam, use_gae=True):
    """[T, N] float32 time-major arrays -> (r_sum, adv) un-normalised."""
    reward, mask, value = _c32(reward), _c32(mask), _c32(value)
    T, N = reward.shape
    r_sum = np.empty((T, N), dtype=np.float32)
    adv = np.empty((T, N), dtype=np.float32)
    lib().oracle_gae(T, N, _f(reward), _f(mask), _f(value), C.c_float(lam), int(use_gae), _f(r_sum), _f(adv))
    return r_sum, adv


def critic_forward(x, sd, prefix=""):
    x = _c32(x)
    M, D = x.shape
    w = [_c32(sd[f"{prefix}net.{i}.{p}"]) for i in (0, 2, 4, 6) for p in ("weight", "bias")]
    md = w[0].shape[0]
    out = np.empty((M, 1), dtype=np.float32)
    lib().oracle_critic_forward(M, D, md, _f(x), *[_f(a) for a in w], _f(out))
    return out


def plain_actor_mean(x, sd, prefix=""):
    x = _c32(x)
    M, D = x.shape
    w = [_c32(sd[f"{prefix}net.{i}.{p}"]) for i in (0, 2, 4, 6) for p in ("weight", "bias")]
    md = w[0].shape[0]
    out = np.empty((M, 1), dtype=np.float32)
    lib().oracle_plain_actor_mean(M, D, md, _f(x), *[_f(a) for a in w], _f(out))
    return out


def modular_actor_mean(x, sd, integrator_dim=1, prefix=""):
    x = _c32(x)
    M, D = x.shape
    names = [("other_net", 0), ("other_net", 2), ("integrator_net", 0), ("integrator_net", 2), ("net", 0), ("net", 2)]
    w = [_c32(sd[f"{prefix}{n}.{i}.{p}"]) for n, i in names for p in ("weight", "bias")]
    md = w[0].shape[0]
    out = np.empty((M, 1), dtype=np.float32)
    lib().oracle_modular_actor_mean(M, D, integrator_dim, md, _f(x), *[_f(a) for a in w], _f(out))
    return out
