"""The streamed 16x16x4 kernel family (csrc/mlp16.hip): width-256 nets -- the reference's live water-tank configuration,
/root/reference/run_watertank_changing.sh:20-27 (ResidualPPO + CriticAdv, net_dim 256, 30-float Stacking10 observation) --
and, under PIME_MLP16=1 (a child process: the switch is read once per process), widths 64 / 128 through the same code.

  forward   : pime_mlp_forward against the oracle's double-accumulated forward of the same weights (3e-5 rel, as the
              32x32x2 family's test) on ragged row counts;
  gradients : pime_ppo_minibatch_grad against PyTorch fp32 autograd of the reference loss (agent.py:637-655) on the same
              minibatch, 3e-4 of each tensor's largest entry (f32 sums over B samples in another order), for batch sizes
              that are / are not multiples of the 64-sample group and that exceed one group per workgroup (slab accumulation);
  bitwise reproducibility of the gradients (slabs summed in slab order);
  the whole update_net against the REFERENCE's weights: tests/test_gpu_update_golden.py [wts10_256]."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def check_forward(kind, md, D, M, seed=0):
    import oracle
    from pime_amd import ops
    from pime_amd.elegantrl.net import CriticAdv
    from pime_amd.elegantrl.net_residual import ActorResidualIntegratorModularPPO, ActorResidualPPO
    torch.manual_seed(seed)
    net = (CriticAdv(D, md) if kind == "critic" else ActorResidualIntegratorModularPPO(md, D, 1, 1) if kind == "modular"
           else ActorResidualPPO(md, D, 1)).to(DEV)
    with torch.no_grad():
        net.net[-1].weight.mul_(5.0)
    x = (torch.randn(M, D) * torch.tensor(([3., 3., 8., 1.] * 8)[:D]) + torch.tensor(([7., 7., 0., 0.] * 8)[:D])).to(DEV)
    pk = ops.PackedMLP.from_module(net)
    got = pk(x).cpu().numpy()
    sd = {k: v.detach().cpu().numpy() for k, v in net.state_dict().items()}
    want = (oracle.critic_forward if kind == "critic" else oracle.modular_actor_mean if kind == "modular"
            else oracle.plain_actor_mean)(x.cpu().numpy(), sd)[:, 0]
    np.testing.assert_allclose(got, want, rtol=3e-5, atol=3e-5 * max(1.0, float(np.abs(want).max())))
    with torch.no_grad():   # and against torch fp32 on the device
        ref = (net(x)[:, 0] if kind == "critic" else net.mean(x)[:, 0]).cpu().numpy()
    np.testing.assert_allclose(got, ref, rtol=1e-4, atol=1e-4 * max(1.0, float(np.abs(ref).max())))


def check_grads(kind, md, D, B, seed=1):
    from pime_amd import ops
    from test_gpu_ppo_fused import _data, _make, _torch_grads
    act, cri = _make(kind, md, D, seed=B + md)
    L = max(3 * B, 5000)
    state, action, logprob, adv, r_sum = _data(L, D, act, seed=seed)
    idx = torch.randint(L, (B,), device=DEV, generator=torch.Generator(device=DEV).manual_seed(2))
    want, s_sur, s_ent, s_cri, scale = _torch_grads(act, cri, state, action, logprob, adv, r_sum, idx, 0.2, 0.02)
    fused = ops.FusedPPOGrad(act, cri, B)
    fused.zero_grad()
    fused.loss_sums.zero_()
    got_scale = torch.zeros(1, device=DEV)
    fused(state, action.reshape(-1).contiguous(), logprob, adv, r_sum, idx, 0.2, 0.02, got_scale)
    torch.cuda.synchronize()
    np.testing.assert_allclose(got_scale.item(), scale.item(), rtol=3e-6)
    got = {n: p.grad for n, p in list(act.named_parameters()) + [("cri." + k, v) for k, v in cri.named_parameters()]
           if p.requires_grad}
    assert set(got) == set(want)
    for name in want:
        w, g = want[name], got[name]
        tol = 3e-4 * float(w.abs().max()) + 1e-7
        err = float((w - g).abs().max())
        assert err <= tol, f"{name}: max |diff| {err:.3e} > {tol:.3e} (|grad|max {float(w.abs().max()):.3e})"
    sums = fused.loss_sums.tolist()
    np.testing.assert_allclose(sums[0], s_sur, rtol=2e-4, atol=1e-3 * B ** 0.5)
    np.testing.assert_allclose(sums[1], s_ent, rtol=2e-4, atol=1e-3 * B ** 0.5)
    np.testing.assert_allclose(sums[2], s_cri, rtol=2e-4)
    np.testing.assert_allclose(sums[4], s_cri * scale.item(), rtol=3e-4)   # the critic part of the logged united loss
    # bitwise reproducibility + the OVERWRITE flag
    g1 = fused.flat_grad.clone()
    fused.flat_grad.fill_(3.0)
    fused(state, action.reshape(-1).contiguous(), logprob, adv, r_sum, idx, 0.2, 0.02, got_scale, overwrite=True)
    torch.cuda.synchronize()
    assert torch.equal(fused.flat_grad, g1), "gradients differ between two identical calls"


@pytest.mark.parametrize("kind,D,M", [("critic", 30, 5000), ("critic", 3, 64), ("critic", 4, 70001), ("actor", 30, 4097),
                                      ("actor", 3, 1)])
def test_forward_width_256(kind, D, M):
    check_forward(kind, 256, D, M)


@pytest.mark.parametrize("D,B", [(30, 4096), (30, 1000), (3, 2048), (4, 777), (12, 40000)])
def test_gradients_width_256_match_autograd(D, B):
    check_grads("resid", 256, D, B)


def test_plain_ppo_actor_width_256():
    check_grads("ppo", 256, 3, 1024)


# ---- the modular actor at width 256 (round 3: mlp16m_forward_kernel / ppo16m_kernel; run_watertank_changing.sh:11-18) -------------
@pytest.mark.parametrize("D,M", [(4, 5000), (3, 64), (4, 70001), (3, 1)])
def test_modular_forward_width_256(D, M):
    check_forward("modular", 256, D, M)


@pytest.mark.parametrize("D,B", [(4, 4096), (3, 1000), (4, 777), (3, 40000)])
def test_modular_gradients_width_256_match_autograd(D, B):
    """ActorResidualIntegratorModularPPO(256) + CriticAdv(256) against PyTorch fp32 autograd of the reference loss, 3e-4 of each
    tensor's largest entry; bitwise repeatable; batch sizes that are / are not multiples of the 64-sample group and that give a
    workgroup more than one group (slab accumulation: 40 000 samples = 625 groups on 256 workgroups)."""
    check_grads("modular", 256, D, B)


def test_width_256_modular_agent_takes_the_hip_path():
    """AgentResidualIntegratorModularPPO at net_dim 256 on the Integrator water tank (the reference script's second block): packed
    forwards + fused gradients + the image map (no torch fallback, no warning), whole explore + update_net; three updates of
    the Adam-fused step keep the packed images equal to a re-pack."""
    import warnings
    from pime_amd import gym_control
    from pime_amd.elegantrl.agent_residual import AgentResidualIntegratorModularPPO
    from pime_amd.elegantrl.run import make_buffer
    env = gym_control.make_vec(gym_control.WT_INTEGRATOR, 512, device=DEV, seed=1, reward_type="distance", max_step=20)
    torch.manual_seed(0)
    with warnings.catch_warnings():
        warnings.simplefilter("error", RuntimeWarning)
        ag = AgentResidualIntegratorModularPPO(device=DEV)
        ag.init(256, env.state_dim, 1, env.n_integrator)
        ag.init_residual({"init_K": env.K.reshape(-1, 1)})
        with torch.no_grad():
            ag.act.net[-1].weight.normal_(0, 0.05)
        ag.weights_changed()
        assert ag._packed_for("act") is not None and ag._packed_for("cri") is not None
        buf = make_buffer(ag, env, 512 * 20)
        for _ in range(3):
            steps = ag.explore_env(env, buf, 512 * 20, 1.0, 0.99)
            oa, oc = ag.update_net(buf, steps, 2048, 2)
    fused = ag._packed.get("fused")
    assert fused, "update_net fell back to torch autograd for the width-256 modular actor"
    assert fused.images_follow_step, fused.image_map_error
    assert np.isfinite(oa) and np.isfinite(oc)
    torch.cuda.synchronize()
    got = [(n["img_fwd"].clone(), n["img_bwd"].clone()) for n in fused.nets]
    fused.repack()
    torch.cuda.synchronize()
    for (gf, gb), n in zip(got, fused.nets):
        assert torch.equal(gf, n["img_fwd"]) and torch.equal(gb, n["img_bwd"]), "packed images drifted from the parameters"
    env.close()


@pytest.mark.parametrize("md,D,B", [(128, 30, 2048), (64, 30, 1000), (128, 12, 4096)])
def test_wide_observation_gradients_are_deterministic(md, D, B):
    """Stacking10 / Stacking4 observations at width 64 / 128: too wide for the LDS-resident gradient kernel, served by the
    16-tile family's slabs (bit-for-bit reproducible; the split pipeline's float atomics were not) -- check_grads asserts both
    the autograd match and the bitwise repeat."""
    check_grads("resid", md, D, B)


def test_width_256_agent_takes_the_hip_path():
    """AgentResidualPPO at net_dim 256 on the Stacking10 env: packed forwards + fused gradients (no silent torch fallback),
    and since round 3 the fused rollout (rollout16_kernel; tests/test_gpu_rollout_oracle.py replays it through the oracle)."""
    from pime_amd import gym_control
    from pime_amd.elegantrl.agent_residual import AgentResidualPPO
    from pime_amd.elegantrl.run import make_buffer
    env = gym_control.make_vec(gym_control.WT_STACKING.format(10), 512, device=DEV, seed=1, reward_type="distance", max_step=20)
    torch.manual_seed(0)
    ag = AgentResidualPPO(device=DEV)
    ag.init(256, env.state_dim, 1)
    ag.init_residual({"init_K": env.K.reshape(-1, 1)})
    assert ag._packed_for("act") is not None and ag._packed_for("cri") is not None
    assert ag._fused_rollout_ok(env)
    buf = make_buffer(ag, env, 512 * 20)
    steps = ag.explore_env(env, buf, 512 * 20, 1.0, 0.99)
    oa, oc = ag.update_net(buf, steps, 2048, 2)
    assert ag._packed.get("fused"), "update_net fell back to torch autograd at width 256"
    assert np.isfinite(oa) and np.isfinite(oc)
    env.close()


_CHILD = r'''
import os, sys
sys.path.insert(0, os.environ["PIME_ROOT"]); sys.path.insert(0, os.path.join(os.environ["PIME_ROOT"], "tests"))
import test_gpu_mlp16 as t
for kind, md, D, M in (("critic", 128, 3, 5000), ("actor", 128, 4, 333), ("critic", 64, 30, 4097), ("actor", 64, 3, 64)):
    t.check_forward(kind, md, D, M)
for md, D, B in ((128, 3, 4096), (128, 3, 70000), (128, 30, 1000), (64, 4, 2048), (64, 12, 777)):
    t.check_grads("resid", md, D, B)
for D, B in ((3, 4096), (4, 1000), (4, 40000)):   # the modular actor at width 128 on ppo16m_kernel<8> (round 4)
    t.check_grads("modular", 128, D, B)
print("MLP16_FORCED_OK")
'''


def test_widths_64_and_128_through_the_16_tile_family():
    env = dict(os.environ, PIME_ROOT=ROOT, PIME_MLP16="1")
    r = subprocess.run([sys.executable, "-c", _CHILD], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "MLP16_FORCED_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
