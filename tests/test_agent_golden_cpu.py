"""Host-side agent logic against the reference-generated golden (tests/golden/ppo_update.npz), on CPU tensors with
the oracle injected as the backend (the product backend is HIP-only).  Pins, seed for seed:
  * same-seed network initialisation (construction order / RNG consumption of the reference),
  * AgentResidual*.explore_env on a one-instance env (actions, noise, rewards, masks exactly as the reference buffer),
  * AgentPPO.update_net (GAE + PPO losses + Adam) given the recorded minibatch indices,
  * the deterministic evaluation episode (run.py:600-619)."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle.cpu_stack import OracleBackend
from oracle_env import OracleSinglePH, OracleSingleWT

CASES = {
    "ph": dict(agent="AgentResidualIntegratorModularPPO", state_dim=3, integrator=1),
    "wt": dict(agent="AgentResidualPPO", state_dim=3, integrator=None),
}


def _sd(g, prefix):
    return {k[len(prefix) + 1:]: torch.from_numpy(g[k].copy()) for k in g.files if k.startswith(prefix + ".")}


def _agent(tag, g):
    from pime_amd.elegantrl import agent_residual
    hyper = g[f"{tag}:hyper"]
    net_dim, lam = int(hyper[0]), float(hyper[4])
    c = CASES[tag]
    ag = getattr(agent_residual, c["agent"])(backend=OracleBackend(), device="cpu")
    ag.lambda_gae_adv = lam
    torch.manual_seed(3)
    if c["integrator"] is not None:
        ag.init(net_dim, c["state_dim"], 1, c["integrator"])
    else:
        ag.init(net_dim, c["state_dim"], 1)
    K = np.array([-0.02, 0.02, 0.035]) if tag == "ph" else np.array([0., 0.4, -0.4])
    ag.init_residual({"init_K": K.reshape(-1, 1)})
    ag.init_actor_zero()
    ag.fix_K()
    with torch.no_grad():
        ag.act.net[-1].weight.normal_(0, 0.05)
    ag.weights_changed()
    return ag, hyper


@pytest.mark.parametrize("tag", ["ph", "wt"])
def test_same_seed_initialisation(tag):
    g = load_golden("ppo_update.npz")
    ag, _ = _agent(tag, g)
    for name, net in (("act0", ag.act), ("cri0", ag.cri)):
        want = _sd(g, f"{tag}:{name}")
        got = net.state_dict()
        assert set(got) == set(want)
        for k in want:
            np.testing.assert_array_equal(got[k].numpy(), want[k].numpy(), err_msg=f"{name}.{k}")


@pytest.mark.parametrize("tag", ["ph", "wt"])
def test_explore_env_single_instance(tag, ph_table_oracle):
    """The reference buffer (states, reward, mask, pre-tanh action, noise) is reproduced exactly: the torch RNG is in
    the same state as the reference's after the same-seed construction, the env draws are injected."""
    from pime_amd.elegantrl.replay import ReplayBuffer
    g = load_golden("ppo_update.npz")
    ag, hyper = _agent(tag, g)
    target_step = int(hyper[1])
    if tag == "ph":
        env = OracleSinglePH(ph_table_oracle, g["ph:draws"])
    else:
        env = OracleSingleWT(g["wt:draws"], g["wt:step_noise"], num_stack=1)
    buf = ReplayBuffer(target_step + env.max_step, env.state_dim, 1, if_on_policy=True, device="cpu")
    steps = ag.explore_env(env, buf, target_step, 1.0, 0.99)
    assert steps == int(g[f"{tag}:steps"])
    buf.update_now_len_before_sample()
    np.testing.assert_array_equal(buf.buf_state[:buf.now_len].numpy(), g[f"{tag}:buf_state"])
    got, want = buf.buf_other[:buf.now_len].numpy(), g[f"{tag}:buf_other"]
    np.testing.assert_array_equal(got[:, 1], want[:, 1])           # mask
    np.testing.assert_array_equal(got[:, 3], want[:, 3])           # noise (same torch stream)
    np.testing.assert_allclose(got[:, 2], want[:, 2], rtol=0, atol=0)  # pre-tanh action
    np.testing.assert_allclose(got[:, 0], want[:, 0], rtol=1e-6, atol=1e-6)  # reward*scale


@pytest.mark.parametrize("tag", ["ph", "wt"])
def test_update_net_matches_reference(tag):
    from pime_amd.elegantrl.replay import ReplayBuffer
    g = load_golden("ppo_update.npz")
    ag, hyper = _agent(tag, g)
    target_step, batch, repeat = int(hyper[1]), int(hyper[2]), int(hyper[3])
    ag.act.load_state_dict(_sd(g, f"{tag}:act0"))
    ag.cri.load_state_dict(_sd(g, f"{tag}:cri0"))
    ag.weights_changed()
    state, other = g[f"{tag}:buf_state"], g[f"{tag}:buf_other"]
    buf = ReplayBuffer(len(state) + 8, state.shape[1], 1, if_on_policy=True, device="cpu")
    buf.extend_buffer(state, other)
    idx = g[f"{tag}:indices"]
    ag.index_hook = lambda step, L, B: torch.from_numpy(idx[step])
    obj_a, obj_c = ag.update_net(buf, target_step, batch, repeat)
    assert idx.shape[0] == int(repeat * len(state) / batch)
    for name, net in (("act1", ag.act), ("cri1", ag.cri)):
        want = _sd(g, f"{tag}:{name}")
        for k, v in net.state_dict().items():
            np.testing.assert_allclose(v.numpy(), want[k].numpy(), rtol=0, atol=2e-6, err_msg=f"{name}.{k}")
    np.testing.assert_allclose([obj_a, obj_c], g[f"{tag}:obj"], rtol=2e-4, atol=1e-5)


WIDE = {
    "ph128": dict(agent="AgentResidualIntegratorModularPPO", integrator=1, K=[-0.02, 0.02, 0.035]),
    "wt64": dict(agent="AgentResidualIntegratorModularPPO", integrator=1, K=[0., 0.4, -0.4, 0.]),
    "wts10_256": dict(agent="AgentResidualPPO", integrator=None, K=[0.] * 27 + [0., 0.4, -0.4]),
}


@pytest.mark.parametrize("tag", list(WIDE))
def test_update_net_matches_reference_at_kernel_widths(tag):
    """The same host logic at the widths the HIP kernels serve (ppo_update_wide.npz: net_dim 128 / 64 / 256); the GPU
    twin of this test (test_gpu_update_golden.py) runs the fused kernels against the same reference weights."""
    from pime_amd.elegantrl import agent_residual
    from pime_amd.elegantrl.replay import ReplayBuffer
    g = load_golden("ppo_update_wide.npz")
    hyper = g[f"{tag}:hyper"]
    net_dim, target_step, batch, repeat, lam = int(hyper[0]), int(hyper[1]), int(hyper[2]), int(hyper[3]), float(hyper[4])
    c = WIDE[tag]
    state, other = g[f"{tag}:buf_state"], g[f"{tag}:buf_other"]
    ag = getattr(agent_residual, c["agent"])(backend=OracleBackend(), device="cpu")
    ag.lambda_gae_adv = lam
    if c["integrator"] is not None:
        ag.init(net_dim, state.shape[1], 1, c["integrator"])
    else:
        ag.init(net_dim, state.shape[1], 1)
    ag.init_residual({"init_K": np.array(c["K"]).reshape(-1, 1)})
    ag.fix_K()
    ag.act.load_state_dict(_sd(g, f"{tag}:act0"))
    ag.cri.load_state_dict(_sd(g, f"{tag}:cri0"))
    ag.weights_changed()
    buf = ReplayBuffer(len(state) + 8, state.shape[1], 1, if_on_policy=True, device="cpu")
    buf.extend_buffer(state, other)
    idx = g[f"{tag}:indices"]
    ag.index_hook = lambda step, L, B: torch.from_numpy(idx[step])
    obj_a, obj_c = ag.update_net(buf, target_step, batch, repeat)
    for name, net in (("act1", ag.act), ("cri1", ag.cri)):
        want = _sd(g, f"{tag}:{name}")
        for k, v in net.state_dict().items():
            np.testing.assert_allclose(v.numpy(), want[k].numpy(), rtol=0, atol=2e-6, err_msg=f"{name}.{k}")
    np.testing.assert_allclose([obj_a, obj_c], g[f"{tag}:obj"], rtol=2e-4, atol=1e-5)


def test_evaluation_episode(ph_table_oracle):
    """get_episode_return with the updated reference policy on the seeded env (run.py:600-619)."""
    from pime_amd.elegantrl.run import get_episode_return
    from pime_amd import gym_compat
    g = load_golden("ppo_update.npz")
    ag, _ = _agent("ph", g)
    ag.act.load_state_dict(_sd(g, "ph:act1"))
    # the reference seeded env.seed(17); np.random.seed(17) and then reset(): params from the global stream, x0/r from np_random
    glob = np.random.RandomState(17)
    envrng, _ = gym_compat.np_random(17)
    qww, qc = glob.uniform(0.005, 0.015), glob.uniform(0.0015, 0.0025)
    x0, r = envrng.uniform(low=0, high=50), envrng.uniform(3., 11.)
    env = OracleSinglePH(ph_table_oracle, [(qww, qc, x0, r)])
    ret, n = get_episode_return(env, ag.act, torch.device("cpu"))
    want_ret, want_n = g["ph:eval"]
    assert n == int(want_n)
    # Not bitwise: in the reference's EVALUATION path the action is a float32 array, and because the freshly reset
    # plant state is a python float, numpy's weak-scalar promotion turns `A*state + B*action` (ph.py:330) and the
    # np.around of the LUT lookup (ph.py:188) into float32 arithmetic from the first step on (measured: state becomes
    # a float32 ndarray; the LUT index is then k or k+1).  The rollout path (float64 actions, agent_residual.py:61)
    # stays float64 and is reproduced exactly above.  This build keeps float64 everywhere: returns agree to 2e-4.
    np.testing.assert_allclose(ret, want_ret, rtol=2e-4)


def test_replay_buffer_ring_semantics():
    from pime_amd.elegantrl.replay import ReplayBuffer
    b = ReplayBuffer(10, 2, 1, if_on_policy=False, device="cpu")
    for i in range(7):
        b.append_buffer(np.array([i, -i], dtype=np.float32), (float(i), 0.99, 0.5))
    b.extend_buffer(np.arange(12, dtype=np.float32).reshape(6, 2), np.ones((6, 3), dtype=np.float32) * 7)
    assert b.if_full and b.next_idx == 3
    b.update_now_len_before_sample()
    assert b.now_len == 10
    r, m, a, s, s2 = b.sample_batch(4, indices=torch.tensor([0, 1, 7, 8]))
    np.testing.assert_array_equal(s2.numpy(), b.buf_state[[1, 2, 8, 9]].numpy())  # successor = next row
    assert r.shape == (4, 1) and a.shape == (4, 1)


def _sd_nets(g, tag):
    return {k[len(tag) + 1:]: torch.from_numpy(g[k].copy()) for k in g.files if k.startswith(tag + ".")}


def test_reference_state_dicts_load_and_reproduce_outputs():
    """Checkpoint compatibility (agent.py:86-114 writes plain state_dicts): the reference's state_dict keys load
    strictly into this package's modules, and forward / get_action_noise (injected noise) / compute_logprob reproduce
    the reference's own torch outputs (tests/golden/nets.npz)."""
    from pime_amd.elegantrl.net import Actor, ActorPPO, CriticAdv, CriticTwin
    from pime_amd.elegantrl.net_residual import ActorResidualIntegratorModularPPO, ActorResidualPPO
    g = load_golden("nets.npz")
    x3, x4, a1, eps = (torch.from_numpy(g[k]) for k in ("x3", "x4", "a1", "eps"))
    cases = [("modular3", ActorResidualIntegratorModularPPO(128, 3, 1, 1), x3),
             ("modular4", ActorResidualIntegratorModularPPO(64, 4, 1, 1), x4),
             ("resid3", ActorResidualPPO(32, 3, 1), x3)]
    for tag, net, x in cases:
        net.load_state_dict(_sd_nets(g, tag), strict=True)
        with torch.no_grad():
            np.testing.assert_allclose(net(x).numpy(), g[f"{tag}:forward"], rtol=1e-6, atol=1e-6)
            action, noise = net.get_action_noise(x, noise=eps)
            np.testing.assert_allclose(action.numpy(), g[f"{tag}:action"], rtol=1e-6, atol=1e-6)
            np.testing.assert_allclose(net.compute_logprob(x, a1).numpy(), g[f"{tag}:logprob"], rtol=1e-5, atol=1e-5)
    cri = CriticAdv(3, 128)
    cri.load_state_dict(_sd_nets(g, "critic3"), strict=True)
    twin = CriticTwin(32, 4, 1)
    twin.load_state_dict(_sd_nets(g, "twin4"), strict=True)
    det = Actor(32, 4, 1)
    det.load_state_dict(_sd_nets(g, "actor4"), strict=True)
    ppo = ActorPPO(32, 3, 1)
    ppo.load_state_dict(_sd_nets(g, "ppo3"), strict=True)
    with torch.no_grad():
        np.testing.assert_allclose(cri(x3).numpy(), g["critic3:forward"], rtol=1e-6, atol=1e-6)
        q1, q2 = twin.get_q1_q2(x4, a1)
        np.testing.assert_allclose(q1.numpy(), g["twin4:q1"], rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(q2.numpy(), g["twin4:q2"], rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(det(x4).numpy(), g["actor4:forward"], rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(ppo(x3).numpy(), g["ppo3:forward"], rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(ppo.compute_logprob(x3, a1).numpy(), g["ppo3:logprob"], rtol=1e-5, atol=1e-5)


def test_save_load_model_round_trip(tmp_path):
    g = load_golden("ppo_update.npz")
    ag, _ = _agent("ph", g)
    ag.save_load_model(str(tmp_path), if_save=True)
    assert sorted(p.name for p in tmp_path.iterdir()) == ["actor.pth", "critic.pth"]
    want = {k: v.clone() for k, v in ag.act.state_dict().items()}
    with torch.no_grad():
        for p in ag.act.parameters():
            p.add_(1.0)
    ag.save_load_model(str(tmp_path), if_save=False)
    for k, v in ag.act.state_dict().items():
        assert torch.equal(v, want[k])
    assert set(want) == {"a_std_log", "priorK", "other_net.0.weight", "other_net.0.bias", "other_net.2.weight",
                         "other_net.2.bias", "integrator_net.0.weight", "integrator_net.0.bias", "integrator_net.2.weight",
                         "integrator_net.2.bias", "net.0.weight", "net.0.bias", "net.2.weight", "net.2.bias"}


@pytest.mark.parametrize("case", ["mw", "big"])
def test_update_net_multi_workgroup_golden_host_logic(case):
    """tests/golden/ppo_update_multi.npz (batch 4 096 / 70 000, the reference's first-step .grad tensors) through this package's
    update_net host logic on CPU tensors (torch autograd; oracle GAE): what the GPU test compares the HIP kernels with is first
    shown to be reproduced by the same host code path -- gradients 2e-5 of each tensor's largest entry (70 000-row
    sums in a different thread order than the single-threaded reference run), weights 2e-6 abs."""
    from pime_amd.elegantrl import agent_residual
    from pime_amd.elegantrl.replay import ReplayBuffer
    g = load_golden("ppo_update_multi.npz")
    tag = "ph128"
    hyper = g[f"{tag}:{case}:hyper"]
    net_dim, target_step, batch, repeat, lam = int(hyper[0]), int(hyper[1]), int(hyper[2]), int(hyper[3]), float(hyper[4])
    state, other = g[f"{tag}:buf_state"], g[f"{tag}:buf_other"]
    ag = agent_residual.AgentResidualIntegratorModularPPO(backend=OracleBackend(), device="cpu")
    ag.lambda_gae_adv = lam
    ag.init(net_dim, 3, 1, 1)
    ag.init_residual({"init_K": np.array([-0.02, 0.02, 0.035]).reshape(-1, 1)})
    ag.fix_K()
    ag.act.load_state_dict(_sd(g, f"{tag}:act0"))
    ag.cri.load_state_dict(_sd(g, f"{tag}:cri0"))
    buf = ReplayBuffer(len(state) + 8, 3, 1, if_on_policy=True, device="cpu")
    buf.extend_buffer(state, other)
    idx = torch.from_numpy(g[f"{tag}:{case}:indices"].astype(np.int64))
    ag.index_hook = lambda step, L, B: idx[step]
    grads = {}
    orig_step = ag.optimizer.step

    def rec_step(*a, **k):
        if not grads:
            for net_tag, net in (("act", ag.act), ("cri", ag.cri)):
                for name, p in net.named_parameters():
                    if p.grad is not None and p.requires_grad:
                        grads[f"{net_tag}.{name}"] = p.grad.detach().numpy().copy()
        return orig_step(*a, **k)
    ag.optimizer.step = rec_step
    obj_a, obj_c = ag.update_net(buf, target_step, batch, repeat)
    want_keys = {k[len(f"{tag}:{case}:grad1:"):] for k in g.files if k.startswith(f"{tag}:{case}:grad1:")}
    assert set(grads) == want_keys
    for k in want_keys:
        want = g[f"{tag}:{case}:grad1:{k}"]
        np.testing.assert_allclose(grads[k], want, rtol=0, atol=2e-5 * float(np.abs(want).max()) + 1e-12, err_msg=k)
    for name, net in (("act1", ag.act), ("cri1", ag.cri)):
        want = _sd(g, f"{tag}:{case}:{name}")
        for k, v in net.state_dict().items():
            np.testing.assert_allclose(v.numpy(), want[k].numpy(), rtol=0, atol=2e-6, err_msg=f"{name}.{k}")
    np.testing.assert_allclose([obj_a, obj_c], g[f"{tag}:{case}:obj"], rtol=1e-4, atol=1e-6)
