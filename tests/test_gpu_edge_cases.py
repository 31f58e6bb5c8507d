"""Degenerate sizes through the fused entry points: one env lane (a single live lane in a 32- or 64-lane tile), minibatches smaller
than one 256-sample group, a batch of one sample, zero optimizer steps, a one-step episode -- each against the oracle or torch
autograd, not just "does not crash"."""
import numpy as np
import pytest
import torch

from rollout_replay import DEV, make_agent, replay_through_oracle

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("md", [128, 256])
def test_single_lane_rollout_update_and_evaluation(md):
    import oracle
    from pime_amd import gym_control
    from pime_amd.elegantrl.run import get_episode_return_vec, make_buffer
    seed, off = 3, 12345
    env = gym_control.make_vec(gym_control.PH_V35, 1, device=DEV, state_mode="mixed", seed=seed, env_offset=off)
    ag = make_agent("ResidualIntegratorModularPPO", env, md)
    assert ag._fused_rollout_ok(env)
    buf = make_buffer(ag, env, 3 * 50)
    assert ag.explore_env(env, buf, 150, 1.0, 0.99) == 150
    ref = oracle.OraclePH(1, oracle.ph_table(), seed=seed, env_offset=off)
    replay_through_oracle(ag, "ResidualIntegratorModularPPO", env, buf, ref, 3, off, True)
    oa, oc = ag.update_net(buf, 150, 33, 2)           # 9 optimizer steps of 33 samples: one partly filled tile
    assert np.isfinite(oa) and np.isfinite(oc) and ag._packed.get("fused")
    assert ag.update_net(buf, 150, 4096, 2) == (0.0, 0.0)   # int(2 * 150 / 4096) == 0 optimizer steps (agent.py:629)
    if md == 128:
        r = get_episode_return_vec(env, ag.act, fused=ag.fused_eval_policy(env))
        assert r.shape == (1,) and np.isfinite(r).all()
    env.close()


def test_batch_of_one_sample_matches_autograd():
    """B = 1: the unbiased std of one target is NaN in torch (agent.py:652 would poison the critic); the kernels define the scale
    as 1 / (0 + 1e-5) there -- so only the ACTOR's gradients are compared with autograd, the critic's with the scale divided out."""
    from pime_amd import ops
    from test_gpu_ppo_fused import _data, _make
    act, cri = _make("modular", 128, 3, seed=2)
    state, action, logprob, adv, r_sum = _data(64, 3, act, seed=1)
    idx = torch.tensor([17], device=DEV)
    fused = ops.FusedPPOGrad(act, cri, 1)
    fused.zero_grad()
    scale = torch.zeros(1, device=DEV)
    fused(state, action.reshape(-1).contiguous(), logprob, adv, r_sum, idx, 0.2, 0.02, scale)
    torch.cuda.synchronize()
    got = {n: p.grad.clone() for n, p in act.named_parameters() if p.requires_grad}
    got_c = {n: p.grad.clone() for n, p in cri.named_parameters()}
    for p in list(act.parameters()) + list(cri.parameters()):
        p.grad = None
    s, a = state[idx], action[idx]
    new_lp = act.compute_logprob(s, a)
    ratio = (new_lp - logprob[idx]).exp()
    sur = torch.min(adv[idx] * ratio, adv[idx] * ratio.clamp(0.8, 1.2))
    obj = -sur.mean() + (new_lp.exp() * new_lp).mean() * 0.02
    obj.backward()
    for n, p in act.named_parameters():
        if p.requires_grad:
            tol = 3e-4 * float(p.grad.abs().max()) + 1e-7
            assert float((p.grad - got[n]).abs().max()) <= tol, n
    torch.nn.functional.smooth_l1_loss(cri(s).squeeze(1), r_sum[idx]).backward()
    assert float(scale) == pytest.approx(1e5, rel=1e-5)
    for n, p in cri.named_parameters():
        tol = 3e-4 * float(p.grad.abs().max()) + 1e-7
        assert float((p.grad - got_c[n] / float(scale)).abs().max()) <= tol, n


def test_one_step_episodes():
    """max_episode_steps = 1: every step ends an episode and auto-resets in the kernel; 40 'episodes' in one launch."""
    import oracle
    from pime_amd import gym_control
    from pime_amd.elegantrl.replay import TrajectoryBuffer
    N, seed = 96, 8
    env = gym_control.make_vec(gym_control.PH_V35, N, device=DEV, state_mode="mixed", seed=seed, max_episode_steps=1)
    assert env.max_step == 1
    ag = make_agent("ResidualPPO", env, 64)
    buf = TrajectoryBuffer(40, N, 3, 1, DEV)
    assert ag.explore_env(env, buf, 40 * N, 1.0, 0.99) == 40 * N
    assert bool(buf.done[:40].all()) and bool((buf.mask[:40] == 0).all())
    ref = oracle.OraclePH(N, oracle.ph_table(), seed=seed, max_steps=1)
    obs = ref.reset()
    np.testing.assert_array_equal(buf.state[0].cpu().numpy(), obs)
    for t in range(40):
        act = oracle.residual_action(buf.action[t, :, 0].cpu().numpy(), buf.state[t].cpu().numpy(), ag._rollout_priorK())
        obs, _, rew, d = ref.step(act, auto_reset=True)
        assert d.all()
        np.testing.assert_allclose(buf.reward[t].cpu().numpy(), rew, rtol=2e-5, atol=2e-5)
        np.testing.assert_array_equal(buf.state[t + 1].cpu().numpy(), obs)     # the next episode's first observation
    env.close()
