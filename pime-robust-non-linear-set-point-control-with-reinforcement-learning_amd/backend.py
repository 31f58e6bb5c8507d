"""Compute backend of the agents' non-autograd hot ops (value pass, policy mean for rollouts, GAE scan).

The product ships exactly one backend, `HipBackend`, which drives libpime_hip.so and refuses CPU tensors.
The agents take the backend as a constructor argument only so that the *tests* can inject the CPU oracle
(oracle/cpu_stack.py:OracleBackend) to exercise the host-side update logic in a GPU-less container and under gloo;
nothing in this package constructs any other backend, and there is no automatic selection or fallback.
"""
import warnings

import torch

from . import native, ops

_warned = set()


def _warn_once(key, msg):
    """A net shape without a hand-written kernel runs as plain torch modules on the GPU (rocBLAS + autograd): several
    times slower (profiles/r01_a_*), never silently."""
    if key not in _warned:
        _warned.add(key)
        warnings.warn(msg, RuntimeWarning, stacklevel=3)


class HipBackend:
    name = "hip"

    def check_device(self, device):
        if torch.device(device).type != "cuda":
            raise native.PimeError("pime_amd agents run their rollout/GAE/value-pass on a gfx950 GPU; "
                                   f"device '{device}' has no implementation (no CPU fallback)")

    def gae(self, reward, mask, value, lam, use_gae):
        return ops.gae_scan(reward, mask, value, lam, use_gae)

    def packed(self, module):
        """PackedMLP for `module` if the fused kernel supports its shape, else None (-> plain torch forward on
        the GPU, i.e. rocBLAS; still not a CPU path)."""
        kind = getattr(module, "packed_kind", None)
        if kind is None or getattr(module, "action_dim", 1) != 1:
            _warn_once(("fwd", type(module).__name__), f"pime_amd: no fused forward kernel for {type(module).__name__} "
                       f"(action_dim {getattr(module, 'action_dim', '?')}): using the torch module on the GPU")
            return None
        md = module.net[0].out_features if kind != "modular_actor" else module.other_net[0].out_features
        if not ops.PackedMLP.supported(kind, module.state_dim, getattr(module, "integrator_dim", 0), md):
            _warn_once(("fwd", kind, md, module.state_dim),
                       f"pime_amd: no fused forward kernel for {kind} width {md} state_dim {module.state_dim} "
                       f"({native.last_error()}); using the torch module on the GPU.  Supported: INTEGRATION.md 'Supported shapes'")
            return None
        return ops.PackedMLP.from_module(module)

    def packed_twin_heads(self, cri):
        """(PackedMLP, PackedMLP) for the two Q heads of a CriticTwin (net.py: trunk D+A -> md ReLU -> md ReLU, heads md -> 1), or
        None.  The forward kernel's critic image is three hidden ReLU layers + a head; the twin is served by an IDENTITY third layer:
        relu(I h + 0) = h exactly for h = relu(..) >= 0 (every product is h_j * 1 or h_j * 0), so q_k = head_k(trunk(s, a)) comes out of
        one pime_mlp_forward launch per head with the trunk's own weights."""
        sa = getattr(cri, "net_sa", None)
        if sa is None or cri.net_q1.out_features != 1:
            return None
        md, din = sa[0].out_features, sa[0].in_features
        if not ops.PackedMLP.supported("critic", din, 0, md):
            return None
        dev = sa[0].weight.device
        eye, zero = torch.eye(md, device=dev), torch.zeros(md, device=dev)
        out = []
        for head in (cri.net_q1, cri.net_q2):
            pk = ops.PackedMLP("critic", din, 0, md, dev)
            pk._src = [sa[0].weight, sa[0].bias, sa[2].weight, sa[2].bias, eye, zero, head.weight, head.bias]
            out.append(pk.repack())
        return tuple(out)

    def fused_ppo(self, act, cri, max_batch):
        """ops.FusedPPOGrad for these nets, or False when their shape has no fused kernel."""
        if not ops.FusedPPOGrad.supported(act, cri):
            _warn_once(("grad", type(act).__name__, type(cri).__name__, getattr(act, "state_dim", None)),
                       f"pime_amd: no fused PPO gradient kernel for ({type(act).__name__}, {type(cri).__name__}) at these "
                       f"shapes ({native.last_error()}); update_net runs through torch autograd on the GPU (several times "
                       "slower).  Supported: INTEGRATION.md 'Supported shapes'")
            return False
        return ops.FusedPPOGrad(act, cri, max_batch)

    def fused_td3(self, agent, max_batch):
        """ops.FusedTD3 for a TD3 agent's four nets, or False when their shape has no fused optimizer step."""
        if not ops.FusedTD3.supported(agent.act, agent.cri):
            _warn_once(("td3", type(agent.act).__name__, getattr(agent.act, "state_dim", None)),
                       f"pime_amd: no fused TD3 step for ({type(agent.act).__name__}, {type(agent.cri).__name__}) at these shapes; "
                       "update_net runs through torch autograd on the GPU (several times slower).  Supported: state_dim <= 7, "
                       "width 64 / 128, action_dim 1")
            return False
        return ops.FusedTD3(agent.act, agent.act_target, agent.cri, agent.cri_target, max_batch, agent.learning_rate)
