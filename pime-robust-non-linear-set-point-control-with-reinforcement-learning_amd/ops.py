"""Tensor-level wrappers over the trajectory / MLP kernels of libpime_hip.so.

All operands are CUDA (ROCm) tensors; launches go to torch's current stream.  Nothing here has a CPU path:
a CPU tensor raises.
"""
import ctypes as C

import torch

from . import native

_KINDS = {"critic": native.MLP_CRITIC, "plain_actor": native.MLP_PLAIN_ACTOR, "modular_actor": native.MLP_MODULAR_ACTOR}
_PARAM_ORDER = {
    "critic": ["net.0", "net.2", "net.4", "net.6"],
    "plain_actor": ["net.0", "net.2", "net.4", "net.6"],
    "modular_actor": ["other_net.0", "other_net.2", "integrator_net.0", "integrator_net.2", "net.0", "net.2"],
}


def _need_cuda(*ts):
    for t in ts:
        if not t.is_cuda:
            raise native.PimeError("pime_amd.ops kernels need GPU tensors (no CPU fallback)")


def _stream(t):
    return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def gae_scan(reward, mask, value, lam, use_gae=True, out_r_sum=None, out_adv=None):
    """ElegantRL's reward-sum / advantage recursion over a time-major [T, N] float32 buffer
    (replaces elegantrl/agent.py:666-708).  Returns un-normalised (r_sum, adv), both [T, N]."""
    _need_cuda(reward, mask, value)
    T, N = reward.shape
    reward, mask, value = (x.contiguous() for x in (reward, mask, value))
    assert reward.dtype == mask.dtype == value.dtype == torch.float32 and mask.shape == value.shape == (T, N)
    r_sum = torch.empty_like(reward) if out_r_sum is None else out_r_sum
    adv = torch.empty_like(reward) if out_adv is None else out_adv
    with torch.cuda.device(reward.device):
        native.check(native.lib().pime_gae_scan(native.ptr(reward), native.ptr(mask), native.ptr(value), T, N,
                                                C.c_float(lam), int(bool(use_gae)), native.ptr(r_sum), native.ptr(adv),
                                                _stream(reward)), "pime_gae_scan")
    return r_sum, adv


class PackedMLP:
    """A small MLP re-laid for the fused f32-MFMA forward kernel (csrc/mlp_mfma.hip).

    `repack()` must be called after the source weights change (it is one tiny launch); `__call__` runs the whole
    net per 32-row tile with activations in registers and returns the scalar head, shape [M]."""

    def __init__(self, kind, state_dim, integrator_dim, mid_dim, device):
        self.kind, self.D, self.Di, self.md = kind, int(state_dim), int(integrator_dim), int(mid_dim)
        self.device = torch.device(device)
        n = native.lib().pime_mlp_packed_floats(_KINDS[kind], self.D, self.Di, self.md)
        if n <= 0:
            raise native.PimeError(f"fused MLP forward unsupported for {kind} D={state_dim} width={mid_dim}: "
                                   f"{native.last_error()}")
        self.packed = torch.empty(n, dtype=torch.float32, device=self.device)
        self._src = None

    @staticmethod
    def supported(kind, state_dim, integrator_dim, mid_dim):
        return native.lib().pime_mlp_packed_floats(_KINDS[kind], int(state_dim), int(integrator_dim), int(mid_dim)) > 0

    @classmethod
    def from_state_dict(cls, kind, sd, state_dim, integrator_dim=0, prefix=""):
        names = _PARAM_ORDER[kind]
        md = sd[f"{prefix}{names[0]}.weight"].shape[0]
        dev = sd[f"{prefix}{names[0]}.weight"].device
        self = cls(kind, state_dim, integrator_dim, md, dev)
        self._src = [sd[f"{prefix}{n}.{p}"] for n in names for p in ("weight", "bias")]
        self.repack()
        return self

    @classmethod
    def from_module(cls, module):
        """module: CriticAdv / ActorResidualPPO / ActorPPO / ActorResidualIntegratorModularPPO of this package."""
        kind = module.packed_kind
        sd = dict(module.named_parameters())
        return cls.from_state_dict(kind, sd, module.state_dim, getattr(module, "integrator_dim", 0))

    def repack(self):
        srcs = []
        for t in self._src:
            t = t.detach()
            _need_cuda(t)
            assert t.dtype == torch.float32
            srcs.append(t.contiguous())
        arr = (C.c_void_p * len(srcs))(*[t.data_ptr() for t in srcs])
        with torch.cuda.device(self.device):
            native.check(native.lib().pime_mlp_pack(_KINDS[self.kind], self.D, self.Di, self.md, arr,
                                                    native.ptr(self.packed), _stream(self.packed)), "pime_mlp_pack")
        return self

    def __call__(self, x, out=None):
        _need_cuda(x)
        x = x.contiguous()
        assert x.dtype == torch.float32 and x.dim() == 2 and x.shape[1] == self.D, (x.shape, self.D)
        M = x.shape[0]
        out = torch.empty(M, dtype=torch.float32, device=x.device) if out is None else out
        with torch.cuda.device(x.device):
            native.check(native.lib().pime_mlp_forward(_KINDS[self.kind], native.ptr(x), M, self.D, self.Di, self.md,
                                                       native.ptr(self.packed), native.ptr(out), _stream(x)),
                         "pime_mlp_forward")
        return out
