"""AgentPPO.update_net through the HIP path (value pass on the f32 matrix cores, GAE scan kernel, fused gradient
kernels, flat Adam kernel, re-pack; replayed from HIP graphs) against the REFERENCE's own update_net
(/root/reference/elegantrl/agent.py:611-664 run by tests/golden/make_golden.py on the unmodified reference):
`tests/golden/ppo_update_wide.npz` holds the reference's buffer, its minibatch indices and its weights before / after at the
widths the fused kernels serve (pH ModularPPO net_dim 128 = run_ph_changing.sh; water-tank Integrator ModularPPO 64;
water-tank Stacking10 ResidualPPO 256 = run_watertank_changing.sh).

Both index paths are run: `index_hook` (one tensor per step -> two captured graphs per optimizer step, [gradients] and
[Adam, re-pack]) and `index_table_hook` (all minibatches pre-loaded into the index table the kernels walk with a device-side
row cursor -> ONE captured graph per optimizer step, the path bench.py's timed region replays).

Tolerance: 6-7 Adam steps at lr 1e-4.  Adam normalises the gradient, so a parameter whose gradient is at rounding-noise level
can move by up to lr per step in EITHER direction: weights agree to 3e-5 abs (most to 1e-6) on >= 99 % of every tensor's
elements, and no element is further off than one sign-flipped step (2 lr = 2e-4, + 3e-5); losses to 1e-3 rel.  The gradients
themselves are pinned to the reference's first-step .grad tensors in test_multi_workgroup_update_against_reference_gradients."""
import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

CASES = {
    "wtmod256": dict(agent="AgentResidualIntegratorModularPPO", integrator=1, K=[0., 0.4, -0.4, 0.], file="ppo_update_mod256.npz"),
    "ph128": dict(agent="AgentResidualIntegratorModularPPO", integrator=1, K=[-0.02, 0.02, 0.035]),
    "wt64": dict(agent="AgentResidualIntegratorModularPPO", integrator=1, K=[0., 0.4, -0.4, 0.]),
    "wts10_256": dict(agent="AgentResidualPPO", integrator=None, K=[0.] * 27 + [0., 0.4, -0.4]),
}


def _sd(g, prefix):
    return {k[len(prefix) + 1:]: torch.from_numpy(g[k].copy()) for k in g.files if k.startswith(prefix + ".")}


def _run(tag, mode):
    from pime_amd.elegantrl import agent_residual
    from pime_amd.elegantrl.replay import ReplayBuffer
    g = load_golden(CASES[tag].get("file", "ppo_update_wide.npz"))
    hyper = g[f"{tag}:hyper"]
    net_dim, target_step, batch, repeat, lam = int(hyper[0]), int(hyper[1]), int(hyper[2]), int(hyper[3]), float(hyper[4])
    c = CASES[tag]
    state, other = g[f"{tag}:buf_state"], g[f"{tag}:buf_other"]
    D = state.shape[1]
    ag = getattr(agent_residual, c["agent"])(device=DEV)
    ag.lambda_gae_adv = lam
    if c["integrator"] is not None:
        ag.init(net_dim, D, 1, c["integrator"])
    else:
        ag.init(net_dim, D, 1)
    ag.init_residual({"init_K": np.array(c["K"]).reshape(-1, 1)})
    ag.fix_K()
    ag.act.load_state_dict({k: v.to(DEV) for k, v in _sd(g, f"{tag}:act0").items()})
    ag.cri.load_state_dict({k: v.to(DEV) for k, v in _sd(g, f"{tag}:cri0").items()})
    ag.weights_changed()
    buf = ReplayBuffer(len(state) + 8, D, 1, if_on_policy=True, device=DEV)
    buf.extend_buffer(state, other)
    idx = torch.from_numpy(g[f"{tag}:indices"])
    assert idx.shape == (int(repeat * len(state) / batch), batch)
    if mode == "two_graph":
        ag.index_hook = lambda step, L, B: idx[step]
    else:
        ag.index_table_hook = lambda n, L, B: idx[:n]
    obj_a, obj_c = ag.update_net(buf, target_step, batch, repeat)
    torch.cuda.synchronize()
    fused = ag._packed.get("fused")
    return g, ag, fused, obj_a, obj_c


@pytest.mark.parametrize("mode", ["two_graph", "one_graph"])
# wts10_256: width 256 + 30-float stacked observation (run_watertank_changing.sh:20-27); wtmod256: the modular actor at width 256
# (:11-18; tests/golden/ppo_update_mod256.npz) through the 16-tile family's modular kernels
@pytest.mark.parametrize("tag", ["ph128", "wt64", "wts10_256", "wtmod256"])
def test_hip_update_net_matches_reference_weights(tag, mode):
    g, ag, fused, obj_a, obj_c = _run(tag, mode)
    assert fused, f"{tag}: update_net did not take the fused HIP gradient path"
    st = fused.static
    if mode == "one_graph":
        assert st.graph_full is not None and st.graph_a is None, "the one-graph-per-step path was not captured"
    else:
        # (no second graph when the fused step also keeps the packed images current: nothing is left to launch after it)
        assert st.graph_a is not None and st.graph_full is None and (st.graph_b is not None or fused.images_follow_step)
    worst = 0.0
    for name, net in (("act1", ag.act), ("cri1", ag.cri)):
        want = _sd(g, f"{tag}:{name}")
        got = net.state_dict()
        assert set(got) == set(want)
        for k in want:
            w, v = want[k].numpy(), got[k].cpu().numpy()
            diff = np.abs(w - v)
            close = diff <= 3e-5
            worst = max(worst, float(diff[close].max()) if close.any() else 0.0)
            assert close.mean() >= 0.99, f"{tag} {name}.{k}: only {close.mean():.4f} of the elements within 3e-5"
            np.testing.assert_allclose(v, w, rtol=0, atol=2.3e-4, err_msg=f"{tag} {name}.{k}")
    # the weights moved by ~lr per step: make sure the comparison is not vacuous
    moved = max(float(np.abs(_sd(g, f"{tag}:act1")[k].numpy() - _sd(g, f"{tag}:act0")[k].numpy()).max())
                for k in _sd(g, f"{tag}:act0"))
    assert moved > 3e-4 and worst < 0.1 * moved
    np.testing.assert_allclose([obj_a, obj_c], g[f"{tag}:obj"], rtol=1e-3, atol=1e-4)


@pytest.mark.parametrize("tag", ["ph128", "wts10_256"])
def test_whole_update_graph_equals_per_step_graphs(tag):
    """From the second update_net on, all optimizer steps of an update are ONE captured graph (agent.use_update_graph): one
    replay per update instead of one per step.  Three updates on the reference's buffer with the reference's indices, once
    that way and once with one graph per step: the weights must agree bit for bit, the reported losses to rounding."""
    from pime_amd.elegantrl import agent_residual
    from pime_amd.elegantrl.replay import ReplayBuffer
    g = load_golden("ppo_update_wide.npz")
    hyper = g[f"{tag}:hyper"]
    net_dim, target_step, batch, repeat = int(hyper[0]), int(hyper[1]), int(hyper[2]), int(hyper[3])
    c = CASES[tag]
    state, other = g[f"{tag}:buf_state"], g[f"{tag}:buf_other"]
    D = state.shape[1]
    idx = torch.from_numpy(g[f"{tag}:indices"])
    results = []
    for whole in (True, False):
        ag = getattr(agent_residual, c["agent"])(device=DEV)
        ag.use_update_graph = whole
        if c["integrator"] is not None:
            ag.init(net_dim, D, 1, c["integrator"])
        else:
            ag.init(net_dim, D, 1)
        ag.init_residual({"init_K": np.array(c["K"]).reshape(-1, 1)})
        ag.fix_K()
        ag.act.load_state_dict({k: v.to(DEV) for k, v in _sd(g, f"{tag}:act0").items()})
        ag.cri.load_state_dict({k: v.to(DEV) for k, v in _sd(g, f"{tag}:cri0").items()})
        ag.weights_changed()
        buf = ReplayBuffer(len(state) + 8, D, 1, if_on_policy=True, device=DEV)
        buf.extend_buffer(state, other)
        ag.index_table_hook = lambda n, L, B: idx[:n]
        objs = [ag.update_net(buf, target_step, batch, repeat) for _ in range(3)]
        torch.cuda.synchronize()
        fused = ag._packed.get("fused")
        assert fused and (fused.static.graph_update is not None) == whole
        results.append((fused.flat_param.clone(), objs))
    assert torch.equal(results[0][0], results[1][0]), "weights differ between the whole-update graph and per-step graphs"
    # (the logged loss sums are float atomics over the workgroups: reproducible to rounding, not bitwise)
    np.testing.assert_allclose(np.array(results[0][1]), np.array(results[1][1]), rtol=1e-4, atol=1e-6)


# ------------------------------------------------------------------------------------------------------------------------
# Multi-workgroup batches against the reference's own gradients (tests/golden/ppo_update_multi.npz, VERDICT r02 task 3)
# ------------------------------------------------------------------------------------------------------------------------
def _multi_agent(g, case):
    from pime_amd.elegantrl import agent_residual
    from pime_amd.elegantrl.replay import ReplayBuffer
    tag = "ph128"
    hyper = g[f"{tag}:{case}:hyper"]
    net_dim, target_step, batch, repeat, lam = int(hyper[0]), int(hyper[1]), int(hyper[2]), int(hyper[3]), float(hyper[4])
    state, other = g[f"{tag}:buf_state"], g[f"{tag}:buf_other"]
    ag = agent_residual.AgentResidualIntegratorModularPPO(device=DEV)
    ag.lambda_gae_adv = lam
    ag.init(net_dim, 3, 1, 1)
    ag.init_residual({"init_K": np.array(CASES["ph128"]["K"]).reshape(-1, 1)})
    ag.fix_K()
    ag.act.load_state_dict({k: v.to(DEV) for k, v in _sd(g, f"{tag}:act0").items()})
    ag.cri.load_state_dict({k: v.to(DEV) for k, v in _sd(g, f"{tag}:cri0").items()})
    ag.weights_changed()
    buf = ReplayBuffer(len(state) + 8, 3, 1, if_on_policy=True, device=DEV)
    buf.extend_buffer(state, other)
    idx = torch.from_numpy(g[f"{tag}:{case}:indices"].astype(np.int64))
    assert idx.shape == (int(repeat * len(state) / batch), batch)
    ag.index_table_hook = lambda n, L, B: idx[:n]      # the one-graph-per-step path bench.py replays
    return ag, buf, target_step, batch, repeat, len(state)


@pytest.mark.parametrize("case", ["mw", "big"])
def test_multi_workgroup_update_against_reference_gradients(case):
    """`mw`: batch 4 096 = 16 workgroups of 256 samples per net, 4 optimizer steps.  `big`: batch 70 000 > 65 536 = 256 workgroups
    x 256 samples, so 18 workgroups take a SECOND group and accumulate into their slab; 2 optimizer steps.  Against the unmodified
    reference (make_golden.py:golden_ppo_update_multi): every parameter's .grad after the first backward() within 3e-4 of the
    tensor's largest entry, weights after the first Adam step within 2e-6, final weights 3e-5, losses 1e-3."""
    g = load_golden("ppo_update_multi.npz")
    tag = "ph128"
    # (1) exactly ONE optimizer step: repeat_times chosen so that int(repeat * buf_len / batch) == 1 (agent.py:629)
    ag, buf, target_step, batch, _, buf_len = _multi_agent(g, case)
    ag.update_net(buf, target_step, batch, 1.001 * batch / buf_len)
    torch.cuda.synchronize()
    fused = ag._packed.get("fused")
    assert fused, "update_net did not take the fused HIP gradient path"
    n_checked = 0
    for net_tag, net in (("act", ag.act), ("cri", ag.cri)):
        for name, p in net.named_parameters():
            key = f"{tag}:{case}:grad1:{net_tag}.{name}"
            if not p.requires_grad:
                assert key not in g.files
                continue
            want, got = g[key], p.grad.cpu().numpy()
            tol = 3e-4 * float(np.abs(want).max())
            assert tol > 0
            np.testing.assert_allclose(got, want, rtol=0, atol=tol, err_msg=f"first-step gradient of {net_tag}.{name}")
            n_checked += want.size
    assert n_checked == 67459, n_checked   # every trainable parameter of both nets (67 459 f32)
    for name, net in (("act_step1", ag.act), ("cri_step1", ag.cri)):
        want = _sd(g, f"{tag}:{case}:{name}")
        for k, v in net.state_dict().items():
            np.testing.assert_allclose(v.cpu().numpy(), want[k].numpy(), rtol=0, atol=2e-6, err_msg=f"after step 1: {name}.{k}")
    # (2) the whole update
    ag, buf, target_step, batch, repeat, _ = _multi_agent(g, case)
    obj_a, obj_c = ag.update_net(buf, target_step, batch, repeat)
    torch.cuda.synchronize()
    st = ag._packed["fused"].static
    assert st.graph_full is not None, "the one-graph-per-step path was not captured"
    worst = 0.0
    for name, net in (("act1", ag.act), ("cri1", ag.cri)):
        want = _sd(g, f"{tag}:{case}:{name}")
        for k, v in net.state_dict().items():
            worst = max(worst, float(np.abs(v.cpu().numpy() - want[k].numpy()).max()))
            np.testing.assert_allclose(v.cpu().numpy(), want[k].numpy(), rtol=0, atol=3e-5, err_msg=f"final: {name}.{k}")
    np.testing.assert_allclose([obj_a, obj_c], g[f"{tag}:{case}:obj"], rtol=1e-3, atol=1e-4)
