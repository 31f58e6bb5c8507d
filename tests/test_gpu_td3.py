"""The vectorised off-policy path on the GPU: the TD3 nets against the reference's own outputs
(tests/golden/nets.npz: CriticTwin `twin4:q1/q2`, Actor `actor4:forward`), the per-lane device ring, and the residual-TD3
agent on the HIP water-tank env with its update replayed from HIP graphs.  The update arithmetic itself is pinned against
the reference's weights on the CPU (tests/test_td3_golden_cpu.py); nets / losses / Adam are PyTorch-ROCm here by design
(BASELINE.json north_star: "the residual actor-critic update ... and the replay sampler stay on-device in PyTorch-ROCm")."""
import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _sd_nets(g, tag):
    return {k[len(tag) + 1:]: torch.from_numpy(g[k].copy()).to(DEV) for k in g.files if k.startswith(tag + ".")}


def test_td3_nets_reproduce_reference_outputs_on_gpu():
    from pime_amd.elegantrl.net import Actor, CriticTwin
    g = load_golden("nets.npz")
    x4, a1 = torch.from_numpy(g["x4"]).to(DEV), torch.from_numpy(g["a1"]).to(DEV)
    twin = CriticTwin(32, 4, 1).to(DEV)
    twin.load_state_dict(_sd_nets(g, "twin4"), strict=True)
    det = Actor(32, 4, 1).to(DEV)
    det.load_state_dict(_sd_nets(g, "actor4"), strict=True)
    with torch.no_grad():
        q1, q2 = twin.get_q1_q2(x4, a1)
        np.testing.assert_allclose(q1.cpu().numpy(), g["twin4:q1"], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(q2.cpu().numpy(), g["twin4:q2"], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(det(x4).cpu().numpy(), g["actor4:forward"], rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("fused", [True, False])
def test_td3_update_matches_reference_on_gpu(fused):
    """The reference's update_net golden (weights of all four nets after 6 optimizer steps, 1e-5) through AgentTD3.update_net on the
    GPU: `fused` = the hand-written optimizer step (pime_td3_step: four launches per step), else the PyTorch modules it replaces.
    The sampled rows and the smoothing noise come from torch's CPU generator in the golden and from the device here, so the
    generator's draws (randint, then randn_like, per step -- agent.py:363-365) are replayed: as tables through `draw_hook` on the
    fused path, by patching torch.randint / torch.randn_like on the module path."""
    from pime_amd import ops
    from pime_amd.elegantrl.agent import AgentTD3
    from pime_amd.elegantrl.replay import ReplayBuffer
    g = load_golden("td3_update.npz")
    net_dim, target_step, batch, repeat = (int(v) for v in g["td3:hyper"][:4])
    ag = AgentTD3(device=DEV)
    ag.use_fused_update = fused
    ag.init(net_dim, 4, 1)
    for net, tag in ((ag.act, "act0"), (ag.cri, "cri0"), (ag.act_target, "act0"), (ag.cri_target, "cri0")):
        net.load_state_dict({k[len("td3:" + tag) + 1:]: torch.from_numpy(g[k].copy()).to(DEV) for k in g.files
                             if k.startswith(f"td3:{tag}.")})
    buf = ReplayBuffer(len(g["td3:state"]) + 8, 4, 1, if_on_policy=False, device=DEV)
    buf.extend_buffer(g["td3:state"], g["td3:other"])
    cpu_gen = torch.Generator().manual_seed(77)
    if fused:
        def hook(n_steps, B):
            idx, eps = [], []
            for _ in range(n_steps):
                idx.append(torch.randint(len(g["td3:state"]) - 1, (B,), generator=cpu_gen))
                eps.append(torch.randn((B, 1), generator=cpu_gen)[:, 0])
            idx = torch.stack(idx)
            return idx, idx + 1, torch.stack(eps)
        ag.draw_hook = hook
        obj_a, obj_c = ag.update_net(buf, target_step, batch, repeat)
        assert isinstance(ag._fused_td3, ops.FusedTD3), "the update did not run on the fused step"
    else:
        orig_randint, orig_randn_like = torch.randint, torch.randn_like
        torch.randint = lambda high, size, device=None, **k: orig_randint(high, size, generator=cpu_gen).to(DEV)
        torch.randn_like = lambda t, **k: torch.randn(t.shape, generator=cpu_gen).to(DEV)
        try:
            obj_a, obj_c = ag.update_net(buf, target_step, batch, repeat)
        finally:
            torch.randint, torch.randn_like = orig_randint, orig_randn_like
        assert ag._fused_td3 is None
    for tag, net in (("act1", ag.act), ("cri1", ag.cri), ("act_target1", ag.act_target), ("cri_target1", ag.cri_target)):
        for k, v in net.state_dict().items():
            np.testing.assert_allclose(v.cpu().numpy(), g[f"td3:{tag}.{k}"], rtol=0, atol=1e-5, err_msg=f"{tag}.{k}")
    np.testing.assert_allclose([obj_a, obj_c], g["td3:obj"], rtol=1e-3, atol=1e-5)


def test_residual_td3_on_the_hip_env_with_graph_replay():
    from pime_amd import gym_control
    from pime_amd.elegantrl.agent_residual import AgentResidualTD3
    from pime_amd.elegantrl.replay import VecReplayBuffer
    from pime_amd.elegantrl.run import get_episode_return_vec, make_buffer
    N = 1024
    env = gym_control.make_vec(gym_control.WT_INTEGRATOR, N, device=DEV, seed=5, reward_type="distance", max_step=50)
    torch.manual_seed(0)
    ag = AgentResidualTD3(device=DEV)
    ag.init(128, env.state_dim, 1)
    ag.init_residual({"init_K": env.K.reshape(-1, 1)})
    buf = make_buffer(ag, env, 2 ** 17)
    assert isinstance(buf, VecReplayBuffer)
    r0 = get_episode_return_vec(env, ag.eval_policy).mean()       # the prior P controller alone
    steps = ag.explore_env(env, buf, 60 * N, 1.0, 0.99)
    assert steps == 60 * N and buf.stored_slots == 60
    done_rows = (buf.other[:60, :, 1] == 0).sum().item()
    assert done_rows == N            # 60 lock-steps of 50-step episodes: every lane finished exactly one
    w0 = torch.cat([p.detach().reshape(-1) for p in ag.act.parameters()]).clone()
    oa, oc = ag.update_net(buf, 60 * N, 512, 1)                   # 60 optimizer steps: 2 eager, capture, 58 replays
    torch.cuda.synchronize()
    assert ag._fused_td3 is not None and ag._fused_td3 is not False, "the update must run on the fused TD3 step"
    assert np.isfinite(oa) and np.isfinite(oc)
    w1 = torch.cat([p.detach().reshape(-1) for p in ag.act.parameters()])
    assert torch.isfinite(w1).all() and not torch.equal(w0, w1)
    # a few more rounds: the residual must not destroy the prior controller's return (sanity of the composed agent)
    graph = None
    for k in range(6):
        ag.explore_env(env, buf, 50 * N, 1.0, 0.99)
        ag.update_net(buf, 50 * N, 512, 1)
        if k == 1:   # (k = 0 is the first call with 50 optimizer steps: eager; the second one captures)
            graph = ag._fused_td3.tables["graph"]
            assert graph is not None, "the second update_net call of a shape must capture the whole update into ONE HIP graph"
    assert ag._fused_td3.tables["graph"] is graph, "the update graph must survive from one update_net call to the next (device-side sampler bounds)"
    r1 = get_episode_return_vec(env, ag.eval_policy).mean()
    assert np.isfinite(r1) and r1 > 2.0 * r0, f"return collapsed: prior {r0:.1f} -> {r1:.1f}"   # returns are negative
    env.close()


def test_config2_residual_td3_4096_lanes_replays_through_oracle():
    """BASELINE.json config 2 as worded: "Water-tank env, 4 096 vectorised instances, residual TD3, 1 MI355X".  The env side of
    AgentResidualTD3.explore_env at 4 096 lanes x 210 lock-steps (one in-kernel auto-reset with ensemble resampling at step
    200) is replayed through OracleWT (fp64, nonlinear_watertank.py:800-826,890-939): the ring buffer's recorded residual
    actions + the prior term give the env action; observations / rewards 2e-4 (mixed mode: 20 Euler sub-steps in f32), done
    masks exact, reset observations bit-equal.  Then 210 optimizer steps at batch 4 096 from the two persistent HIP graphs.
    (The update arithmetic itself is pinned to the reference's weights by test_td3_update_matches_reference_on_gpu.)"""
    import oracle
    from pime_amd import gym_control
    from pime_amd.elegantrl.agent_residual import AgentResidualTD3
    from pime_amd.elegantrl.run import make_buffer
    N, seed, offset, T, steps = 4096, 11, 8192, 200, 210
    env = gym_control.make_vec(gym_control.WT_INTEGRATOR, N, device=DEV, state_mode="mixed", seed=seed, env_offset=offset,
                               reward_type="distance")
    assert env.max_step == T
    torch.manual_seed(0)
    ag = AgentResidualTD3(device=DEV)
    ag.init(128, env.state_dim, 1)
    ag.init_residual({"init_K": env.K.reshape(-1, 1)})
    with torch.no_grad():
        ag.act.net[-1].weight.normal_(0, 0.05)   # a non-trivial residual
    buf = make_buffer(ag, env, 2 ** 21)
    assert ag._fused_explore(env) is not None, "config 2 must explore through the fused kernel (pime_rollout_offpolicy)"
    assert ag.explore_env(env, buf, steps * N, 1.0, 0.99) == steps * N and buf.stored_slots == steps
    torch.cuda.synchronize()
    # the stored action is clamp(tanh(actor(s)) + 0.1 * eps, -1, 1) with eps the oracle's Philox stream-2 draw (agent.py:303-305)
    sd = {k: v.detach().cpu().numpy() for k, v in ag.act.state_dict().items()}
    for t in (0, 1, 57, 199, 200, 209):
        s_t, a_t = buf.state[t].cpu().numpy(), buf.other[t, :, 2].cpu().numpy()
        mean = oracle.critic_forward(s_t, sd)[:, 0]
        want = np.clip(np.tanh(mean) + np.float32(0.1) * oracle.explore_noise(ag._rollout_seed, offset, N, 1, t), -1.0, 1.0)
        np.testing.assert_allclose(a_t, want, rtol=0, atol=5e-5, err_msg=f"stored action, step {t}")
    state, other = buf.state[:steps + 1], buf.other[:steps]
    with torch.no_grad():   # the env action exactly as explore_vec_env composed it (float32, same device, same ops)
        a_env = torch.stack([other[t, :, 2:3] + state[t] @ ag.act.priorK for t in range(steps)])[:, :, 0].double().cpu().numpy()
    assert float(other[:, :, 2].abs().max()) <= 1.0
    state, other = state.cpu().numpy(), other.cpu().numpy()
    ref = oracle.OracleWT(N, max_steps=T, reward_type="distance", seed=seed, env_offset=offset)
    np.testing.assert_array_equal(state[0], ref.reset())
    for t in range(steps - 1):   # slot t + 1 is stored for t < steps - 1 (the newest step's successor is still in the agent)
        obs, _, rew, d = ref.step(a_env[t], auto_reset=True)
        assert bool(d.all()) == (t == T - 1) and bool(d.any()) == bool(d.all())
        np.testing.assert_array_equal(other[t, :, 1] == 0, d)
        np.testing.assert_allclose(other[t, :, 0], rew, rtol=2e-4, atol=2e-4, err_msg=f"reward, step {t}")
        if t == T - 1:
            np.testing.assert_array_equal(state[t + 1], obs)    # first observation of the next episode: Philox draws bit-equal
            continue
        np.testing.assert_allclose(state[t + 1], obs, rtol=2e-4, atol=2e-4, err_msg=f"observation, step {t}")
        for name, col in (("h1", 0), ("h2", 1), ("I", 3)):      # re-sync the fp64 oracle to the kernel's f32 state
            ref.set(name, state[t + 1][:, col].astype(np.float64))
    np.testing.assert_allclose(env.get_field("a1"), ref.get("a1"), rtol=1e-7)
    oa, oc = ag.update_net(buf, steps * N, 4096, 1)
    torch.cuda.synchronize()
    assert np.isfinite(oa) and np.isfinite(oc) and ag._fused_td3
    env.close()


@pytest.mark.parametrize("N", [512, 4608])   # 4 608 lanes: beyond one tile per compute unit -> the 16-lane-tile (non-QUAD) instantiation
def test_fused_offpolicy_explore_matches_lock_step_launches(N):
    """The one-launch exploration against the lock-step-by-lock-step path it replaces (torch actor, pime_env_step, torch copies
    into the ring; its torch.randn replaced by the noise the kernel drew), two calls so that the episodes continue across them,
    48 slots for 70 lock-steps so that the ring wraps: same transitions to float32 rounding of the policy forward."""
    from pime_amd import gym_control
    from pime_amd.elegantrl.agent_residual import AgentResidualTD3
    from pime_amd.elegantrl.replay import VecReplayBuffer
    steps = 70

    def run(fused, noise_from=None):
        env = gym_control.make_vec(gym_control.WT_INTEGRATOR, N, device=DEV, state_mode="mixed", seed=6, reward_type="distance",
                                   max_step=25)
        torch.manual_seed(0)
        ag = AgentResidualTD3(device=DEV)
        ag.use_fused_rollout = fused
        ag.init(64, env.state_dim, 1)
        ag.init_residual({"init_K": env.K.reshape(-1, 1)})
        with torch.no_grad():
            ag.act.net[-1].weight.normal_(0, 0.05)
        buf = VecReplayBuffer(48 * N, N, env.state_dim, 1, DEV)
        orig, it = torch.randn_like, iter(range(steps))
        if noise_from is not None:   # eps = (a - tanh(mean)) / 0.1 reproduces the stored action exactly, clipped or not
            f_states, f_other = noise_from

            def replay_noise(a, **k):
                t = next(it)
                return (f_other[t][:, 2].reshape(a.shape) - torch.tanh(ag.act.net(f_states[t]))) / ag.explore_noise
            torch.randn_like = replay_noise
        states, other = [], []
        try:
            for chunk in (30, 40):
                base = buf.next_slot
                assert ag.explore_env(env, buf, chunk * N, 0.5, 0.98) == chunk * N
                for j in range(chunk):
                    states.append(buf.state[(base + j) % buf.slots].clone())
                    other.append(buf.other[(base + j) % buf.slots].clone())
        finally:
            torch.randn_like = orig
        torch.cuda.synchronize()
        assert buf.if_full and buf.next_slot == steps % 48
        assert (ag._fused_explore(env) is not None) == fused
        env.close()
        return states, other

    s_f, o_f = run(True)
    s_s, o_s = run(False, noise_from=(s_f, o_f))
    n_done = 0
    for t in range(steps):
        torch.testing.assert_close(s_s[t], s_f[t], rtol=2e-4, atol=2e-4)
        torch.testing.assert_close(o_s[t], o_f[t], rtol=2e-4, atol=2e-4)
        assert torch.equal(o_s[t][:, 1], o_f[t][:, 1]), "masks (episode ends) must agree exactly"
        n_done += int((o_f[t][:, 1] == 0).sum())
    assert n_done == 2 * N      # 70 lock-steps of 25-step episodes: every lane ended two


def test_td3_target_q_on_the_forward_kernel_matches_the_torch_modules():
    """AgentTD3._target_packs (opt-in, PIME_TD3_FUSED_TARGETS=1): next_a = act_target.get_action(next_s) and
    min(cri_target.get_q1_q2(next_s, next_a)) served by three pime_mlp_forward launches -- the twin heads share the trunk through an
    identity third layer -- against the torch modules on the same batch and the same policy-noise draws (agent.py:363-367,
    net.py:305-332).  Also after a soft update: the images are re-packed behind it."""
    from pime_amd.elegantrl.agent import AgentTD3
    torch.manual_seed(3)
    D, B = 4, 4096
    ag = AgentTD3(device=DEV)
    ag.init(128, D, 1)
    with torch.no_grad():   # targets that differ from the online nets, heads away from their tiny initial scale
        for p in list(ag.act_target.parameters()) + list(ag.cri_target.parameters()):
            p.add_(torch.randn_like(p) * 0.05)
    next_s = torch.randn(B, D, device=DEV) * 2 + 1

    def label(fused):
        ag.use_fused_targets, ag._tpacks = fused, None
        torch.manual_seed(11)   # the same policy-noise draws on both paths
        with torch.no_grad():
            tp = ag._target_packs()
            assert (tp is not None) == fused
            if tp is not None:
                a = tp[0](next_s).tanh().unsqueeze(1)
                noise = (torch.randn_like(a) * ag.policy_noise).clamp(-0.5, 0.5)
                next_a = (a + noise).clamp(-1.0, 1.0)
                sa = torch.cat((next_s, next_a), 1)
                return next_a, torch.min(tp[1](sa), tp[2](sa)).unsqueeze(1)
            next_a = ag.act_target.get_action(next_s, ag.policy_noise)
            return next_a, torch.min(*ag.cri_target.get_q1_q2(next_s, next_a))

    a0, q0 = label(False)
    a1, q1 = label(True)
    assert a1.shape == a0.shape and q1.shape == q0.shape
    np.testing.assert_allclose(a1.cpu().numpy(), a0.cpu().numpy(), rtol=0, atol=2e-6)
    np.testing.assert_allclose(q1.cpu().numpy(), q0.cpu().numpy(), rtol=2e-5, atol=2e-6)
    # a soft update moves the targets; _one_update re-packs the images right behind it
    ag.soft_update(ag.cri_target, ag.cri, 0.3)
    ag.soft_update(ag.act_target, ag.act, 0.3)
    for pk in ag._tpacks:
        pk.repack()
    with torch.no_grad():   # the re-packed images (not rebuilt ones) serve the moved targets
        sa = torch.cat((next_s, a1), 1)
        want = torch.min(*ag.cri_target.get_q1_q2(next_s, a1))
        got = torch.min(ag._tpacks[1](sa), ag._tpacks[2](sa)).unsqueeze(1)
        pre = ag.act_target.net(next_s)[:, 0]
        np.testing.assert_allclose(ag._tpacks[0](next_s).cpu().numpy(), pre.cpu().numpy(), rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(got.cpu().numpy(), want.cpu().numpy(), rtol=2e-5, atol=2e-6)
