"""ctypes binding of libpime_hip.so (include/pime_hip.h) -- the only door to the HIP kernels.

There is no CPU fallback anywhere behind this module: if the library is missing it raises at load, and every
entry point raises `PimeError` with the library's message on a non-zero status (e.g. no gfx950 device).
"""
import ctypes as C
import os
import subprocess

from ._pkg import PACKAGE_DIR

# The product loads the in-tree library and nothing else.  PIME_LIB_PATH (another build of the same library, tools/variant.sh) is honoured
# only when PIME_ALLOW_LIB_OVERRIDE=1 is set beside it -- the same-box kernel A/B scripts (tools/ab_*.sh) do that; on its own it is ignored.
_override = os.environ.get("PIME_LIB_PATH") if os.environ.get("PIME_ALLOW_LIB_OVERRIDE") == "1" else None
LIB_PATH = _override or os.path.join(PACKAGE_DIR, "libpime_hip.so")
CSRC = os.path.join(PACKAGE_DIR, "csrc")

OK = 0
ABI_VERSION = 19
ENV_PH, ENV_WT = 0, 1
STATE_F64, STATE_MIXED, STATE_MIXED16 = 0, 1, 2
REWARD = {"distance": 0, "square_distance": 1, "sparse": 2}
F32, F64 = 0, 1
MLP_CRITIC, MLP_PLAIN_ACTOR, MLP_MODULAR_ACTOR = 0, 1, 2

FIELD = dict(
    ph_x=0, ph_I=1, ph_r=2, ph_y=3, ph_A=4, ph_B=5, ph_C=6, ph_qww_V=7, ph_qc_V=8, ph_t=9, ph_episode=10,
    wt_h1=32, wt_h2=33, wt_r=34, wt_I=35, wt_a1=36, wt_a2=37, wt_Kp=38, wt_t=39, wt_episode=40,
)


class PimeError(RuntimeError):
    pass


class PhChem(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("kw", "kchem", "ka", "MNaOH", "MHA", "MNH3")]


class PpoNet(C.Structure):
    _fields_ = [("kind", C.c_int32), ("D", C.c_int32), ("Di", C.c_int32), ("md", C.c_int32),
                ("params", C.c_void_p), ("grads", C.c_void_p), ("a_std_log", C.c_void_p), ("g_a_std_log", C.c_void_p),
                ("img_fwd", C.c_void_p), ("img_bwd", C.c_void_p), ("workspace", C.c_void_p)]


class PpoBatch(C.Structure):
    _fields_ = [("state", C.c_void_p), ("action", C.c_void_p), ("logprob", C.c_void_p), ("adv", C.c_void_p),
                ("r_sum", C.c_void_p), ("indices", C.c_void_p), ("B", C.c_int32), ("flags", C.c_int32), ("index_row", C.c_void_p),
                ("dp_moments", C.c_void_p)]


PPO_OVERWRITE_GRADS = 1


class Adam(C.Structure):
    _fields_ = [("param", C.c_void_p), ("grad", C.c_void_p), ("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p),
                ("step", C.c_void_p), ("n", C.c_int64), ("lr", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float),
                ("eps", C.c_float), ("image_map", C.c_void_p), ("dp_moments", C.c_void_p), ("critic_offset", C.c_int64),
                ("dp_world", C.c_int32)]


class Td3Net(C.Structure):
    _fields_ = [("param", C.c_void_p), ("target", C.c_void_p), ("grad", C.c_void_p), ("exp_avg", C.c_void_p),
                ("exp_avg_sq", C.c_void_p), ("step", C.c_void_p), ("lr", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float),
                ("eps", C.c_float)]


class Td3Batch(C.Structure):
    _fields_ = [("state", C.c_void_p), ("other", C.c_void_p), ("idx", C.c_void_p), ("nxt", C.c_void_p), ("noise", C.c_void_p),
                ("row", C.c_int64), ("epoch", C.c_void_p), ("B", C.c_int32), ("noise_seed", C.c_uint64), ("noise_epoch", C.c_uint32),
                ("policy_noise", C.c_float), ("noise_clip", C.c_float)]


class EnvCfg(C.Structure):
    _fields_ = [
        ("kind", C.c_int32), ("n_envs", C.c_int32), ("device_id", C.c_int32), ("state_mode", C.c_int32),
        ("max_steps", C.c_int32), ("reward_type", C.c_int32), ("integral_bound", C.c_int32),
        ("num_stack", C.c_int32), ("resample_every", C.c_int32), ("env_offset", C.c_uint32), ("seed", C.c_uint64),
        ("integral_max", C.c_double), ("integral_punish", C.c_double), ("action_punish", C.c_double),
        ("action_change_punish", C.c_double), ("distance_threshold", C.c_double),
        ("range_lo", C.c_double * 3), ("range_hi", C.c_double * 3),
        ("init_lo", C.c_double * 2), ("init_hi", C.c_double * 2),
        ("ph_sample_t", C.c_double), ("ph_u_low", C.c_double), ("ph_u_high", C.c_double),
        ("ph_table_scale", C.c_double), ("ph_table", C.POINTER(C.c_double)), ("ph_table_len", C.c_int32),
        ("wt_n_discrete", C.c_int32),
        ("wt_A1", C.c_double), ("wt_A2", C.c_double), ("wt_G", C.c_double), ("wt_dt", C.c_double),
        ("wt_noise_scale", C.c_double), ("wt_z1", C.c_double), ("wt_pmax", C.c_double),
    ]


_vp, _i32, _u8p = C.c_void_p, C.c_int32, C.c_void_p
_SIGNATURES = {
    # name: (restype, argtypes)   -- one entry per function declared in include/pime_hip.h
    "pime_abi_version": (C.c_int, []),
    "pime_last_error": (C.c_char_p, []),
    "pime_device_count": (C.c_int, []),
    "pime_ph_table_build": (C.c_int, [C.POINTER(PhChem), C.c_double, _i32, _vp]),
    "pime_env_cfg_default": (C.c_int, [_i32, C.POINTER(EnvCfg)]),
    "pime_env_create": (_vp, [C.POINTER(EnvCfg)]),
    "pime_env_destroy": (None, [_vp]),
    "pime_capture_begin": (None, []),
    "pime_capture_end": (None, []),
    "pime_capture_leave": (None, []),
    "pime_deferred_releases": (C.c_int, []),
    "pime_env_obs_dim": (_i32, [_vp]),
    "pime_env_num_envs": (_i32, [_vp]),
    "pime_env_reset_draw_width": (_i32, [_vp]),
    "pime_env_reset": (C.c_int, [_vp, _u8p, _vp, _vp, _vp]),
    "pime_env_step": (C.c_int, [_vp, _vp, _i32, _vp, _i32, _vp, _vp, _vp, _vp, _vp]),
    "pime_env_step_residual": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp]),
    "pime_env_reset_h": (C.c_int, [_vp, _u8p, _vp, _vp, _vp]),
    "pime_env_step_h": (C.c_int, [_vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp]),
    "pime_env_step_residual_h": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp]),
    "pime_env_read_field": (C.c_int, [_vp, _i32, _vp, _vp]),
    "pime_env_write_field": (C.c_int, [_vp, _i32, _vp, _vp, _vp]),
    "pime_env_set_punish": (C.c_int, [_vp, C.c_double, C.c_double, C.c_double]),
    "pime_env_set_max_steps": (C.c_int, [_vp, _i32]),
    "pime_env_set_resample_every": (C.c_int, [_vp, _i32]),
    "pime_env_observe": (C.c_int, [_vp, _vp, _vp]),
    "pime_gae_scan": (C.c_int, [_vp, _vp, _vp, _i32, _i32, C.c_float, _i32, _vp, _vp, _vp]),
    "pime_mlp_packed_floats": (C.c_int64, [_i32, _i32, _i32, _i32]),
    "pime_mlp_pack": (C.c_int, [_i32, _i32, _i32, _i32, _vp, _vp, _vp]),
    "pime_mlp_forward": (C.c_int, [_i32, _vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp]),

    "pime_ppo_fwd_image_floats": (C.c_int64, [_i32, _i32, _i32, _i32]),
    "pime_ppo_bwd_image_floats": (C.c_int64, [_i32, _i32, _i32, _i32]),
    "pime_ppo_bwd_image_f32_floats": (C.c_int64, [_i32, _i32, _i32, _i32]),
    "pime_ppo_workspace_floats": (C.c_int64, [_i32, _i32, _i32]),
    "pime_ppo_pack_bwd": (C.c_int, [_i32, _i32, _i32, _i32, _vp, _vp, _vp]),
    "pime_ppo_repack": (C.c_int, [_vp, _vp, _vp]),
    "pime_rollout_supported": (C.c_int, [_vp, _i32, _i32]),
    "pime_rollout": (C.c_int, [_vp, _i32, _i32, _vp, _vp, _vp, _i32, C.c_uint64, C.c_uint32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "pime_rollout_eval_supported": (C.c_int, [_vp, _i32, _i32]),
    "pime_rollout_eval": (C.c_int, [_vp, _i32, _i32, _vp, _vp, _i32, _i32, _vp, _i32, _vp, _vp, _vp]),
    "pime_rollout_offpolicy_supported": (C.c_int, [_vp, _i32]),
    "pime_rollout_offpolicy": (C.c_int, [_vp, _i32, _vp, _vp, C.c_float, C.c_float, C.c_float, _i32, C.c_uint64, C.c_uint32, _vp, _vp,
                                         _vp, _i32, _i32, _vp]),
    "pime_oneshot_create": (_vp, [_i32, _i32, C.c_int64, _i32]),
    "pime_oneshot_export": (C.c_int, [_vp, _vp]),
    "pime_oneshot_connect": (C.c_int, [_vp, _vp]),
    "pime_oneshot_allreduce_mean": (C.c_int, [_vp, _vp, _vp]),
    "pime_oneshot_status": (C.c_int, [_vp]),
    "pime_oneshot_info": (C.c_int, [_vp, _vp]),
    "pime_oneshot_destroy": (None, [_vp]),
    "pime_rollout_h": (C.c_int, [_vp, _i32, _i32, _vp, _vp, _vp, _i32, C.c_uint64, C.c_uint32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "pime_adam_step": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int64, C.c_float, C.c_float, C.c_float, C.c_float, _vp, _vp]),
    "pime_ppo_minibatch_grad": (C.c_int, [_vp, _vp, _vp, C.c_float, C.c_float, _vp, _vp, _vp, _vp]),
    "pime_ppo_minibatch_step": (C.c_int, [_vp, _vp, _vp, C.c_float, C.c_float, _vp, _vp, _vp, _vp, _vp]),
    "pime_ppo_image_map": (C.c_int, [_vp, _vp, _vp, C.c_int64, _vp, _vp]),
    "pime_adam_step_images": (C.c_int, [_vp, _vp, _vp, _vp]),
    "pime_adam_step_dp": (C.c_int, [_vp, _vp, _vp, _vp]),
    "pime_td3_supported": (C.c_int, [_i32, _i32, _i32]),
    "pime_td3_param_floats": (C.c_int64, [_i32, _i32, _i32]),
    "pime_td3_param_offsets": (C.c_int, [_i32, _i32, _i32, _vp]),
    "pime_td3_workspace_floats": (C.c_int64, [_i32, _i32, _i32]),
    "pime_td3_step": (C.c_int, [_i32, _i32, _vp, _vp, _vp, C.c_float, _i32, _i32, _i32, _vp, _vp, _vp]),
}
EXPORTS = tuple(_SIGNATURES)

_lib = None


def build(verbose=False):
    """Compile libpime_hip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    cmd = ["make", "-C", CSRC, "-j8"] + ([] if verbose else ["-s"])
    subprocess.check_call(cmd)
    return LIB_PATH


def lib():
    """Load libpime_hip.so.  torch is imported first so that the library binds to the HIP runtime torch already
    loaded (same libamdhip64.so.7 soname) and streams/pointers are shared with it."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise PimeError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                            f"or `make -C {CSRC}`.  pime_amd has no CPU fallback.")
        import torch  # noqa: F401  (loads the HIP runtime this library must share)
        handle = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError here = header/library mismatch
            fn.restype = res
            fn.argtypes = args
        if handle.pime_abi_version() != ABI_VERSION:
            raise PimeError(f"libpime_hip.so ABI {handle.pime_abi_version()} != binding ABI {ABI_VERSION}; rebuild")
        _lib = handle
    return _lib


class capture_guard:
    """Context manager around EVERY stream capture (torch.cuda.graph) of this package: no cyclic garbage collection inside (a
    collection can finalise an unrelated object that owns device memory), and the library parks any device memory a handle releases
    meanwhile instead of calling hipFree under the capture, which aborts the process (include/pime_hip.h: pime_capture_begin)."""

    def __enter__(self):
        import gc
        self._gc = gc.isenabled()
        gc.disable()
        lib().pime_capture_begin()
        return self

    def __exit__(self, *exc):
        import gc
        lib().pime_capture_end()
        if self._gc:
            gc.enable()
        return False


def destroy_handle(destroy, handle):
    """The one finaliser path of every Python object that owns library-side device memory (env handles, one-shot all-reduce
    regions): outside a capture `destroy(handle)` frees at once; inside one -- announced by capture_guard, or seen by torch on the
    current stream -- the library parks the memory and frees it at its next entry point outside the capture."""
    import torch
    unannounced = False
    try:
        unannounced = torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()
    except Exception:   # interpreter shutdown
        pass
    if unannounced:
        lib().pime_capture_begin()
    try:
        destroy(handle)
    finally:
        if unannounced:
            lib().pime_capture_leave()


def last_error():
    return lib().pime_last_error().decode("utf-8", "replace")


def check(status, what=""):
    if status != OK:
        raise PimeError(f"{what or 'libpime_hip'} failed ({status}): {last_error()}")


def device_count():
    return int(lib().pime_device_count())


def ptr(t):
    """Device/host address of a torch tensor or numpy array; None -> NULL."""
    if t is None:
        return None
    if hasattr(t, "data_ptr"):
        return C.c_void_p(t.data_ptr())
    return C.c_void_p(t.ctypes.data)


def current_stream():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ph_table_build(n=100000, step=1e-5, chem=None):
    import numpy as np
    out = np.empty(n, dtype=np.float64)
    c = None if chem is None else C.byref(PhChem(**chem))
    check(lib().pime_ph_table_build(c, step, n, ptr(out)), "pime_ph_table_build")
    return out
