"""Ragged lane counts: the fused kernels map 32 lanes to a wave and 64 (rollouts) / 256 (gradient groups) to a workgroup; a lane count
that is no multiple of either leaves a partly filled last tile whose idle lanes shadow the last env (compute, never store).  Every
fused entry point is run at N = 1 000 (31 full tiles + 8 lanes) and N = 33, against the oracle / the launch-by-launch path, and the
memory right behind every output buffer is checked for stray writes."""
import numpy as np
import pytest
import torch

from rollout_replay import DEV, make_agent, replay_through_oracle

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("N", [1000, 33])
def test_fused_rollout_with_a_partly_filled_tile(N):
    import oracle
    from pime_amd import gym_control
    from pime_amd.elegantrl.replay import TrajectoryBuffer
    seed, off = 5, 77
    env = gym_control.make_vec(gym_control.PH_V35, N, device=DEV, state_mode="mixed", seed=seed, env_offset=off)
    ag = make_agent("ResidualIntegratorModularPPO", env, 128)
    assert ag._fused_rollout_ok(env)
    T = env.max_step
    buf = TrajectoryBuffer(2 * T, N, 3, 1, DEV)
    # guard words right behind every trajectory tensor: the kernel must not write past lane N - 1 of the last slot
    guards = {}
    for name in ("state", "action", "noise", "reward", "done"):
        t = getattr(buf, name)
        big = torch.full((t.numel() + 64,), 7, dtype=t.dtype, device=DEV)
        view = big[:t.numel()].view(t.shape)
        view.zero_()
        setattr(buf, name, view)
        guards[name] = big
    assert ag.explore_env(env, buf, 2 * N * T, 1.0, 0.99) == 2 * N * T
    torch.cuda.synchronize()
    for name, big in guards.items():
        assert bool((big[-64:] == 7).all()), f"stray write behind the {name} buffer"
    ref = oracle.OraclePH(N, oracle.ph_table(), seed=seed, env_offset=off)
    replay_through_oracle(ag, "ResidualIntegratorModularPPO", env, buf, ref, 2, off, True)
    oa, oc = ag.update_net(buf, 2 * N * T, 4096, 2)      # batches of 4 096 out of 2 N T rows: partly filled last group as well
    assert np.isfinite(oa) and np.isfinite(oc)
    env.close()


@pytest.mark.parametrize("N", [1000, 33])
def test_fused_evaluation_and_offpolicy_exploration_with_a_partly_filled_tile(N):
    from pime_amd import gym_control
    from pime_amd.elegantrl.agent_residual import AgentResidualTD3
    from pime_amd.elegantrl.replay import VecReplayBuffer
    from pime_amd.elegantrl.run import get_episode_return_vec
    envs = [gym_control.make_vec(gym_control.WT_INTEGRATOR, N, device=DEV, state_mode="mixed", seed=9, reward_type="distance",
                                 max_step=40) for _ in range(2)]
    ag = make_agent("ResidualIntegratorModularPPO", envs[0], 64)
    fused = ag.fused_eval_policy(envs[0])
    assert fused is not None
    got = get_episode_return_vec(envs[0], ag.act, fused=fused)
    slow = get_episode_return_vec(envs[1], ag.act)
    assert got.shape == (N,)
    np.testing.assert_allclose(got, slow, rtol=1e-4, atol=1e-3)
    # off-policy exploration into a guarded ring
    env = envs[0]
    torch.manual_seed(0)
    td3 = AgentResidualTD3(device=DEV)
    td3.init(64, env.state_dim, 1)
    td3.init_residual({"init_K": env.K.reshape(-1, 1)})
    buf = VecReplayBuffer(50 * N, N, env.state_dim, 1, DEV)
    big_s = torch.full((buf.state.numel() + 64,), 7.0, device=DEV)
    big_o = torch.full((buf.other.numel() + 64,), 7.0, device=DEV)
    buf.state = big_s[:buf.state.numel()].view(buf.state.shape).zero_()
    buf.other = big_o[:buf.other.numel()].view(buf.other.shape).zero_()
    buf.buf_state, buf.buf_other = buf.state.view(buf.max_len, -1), buf.other.view(buf.max_len, -1)
    assert td3._fused_explore(env) is not None
    assert td3.explore_env(env, buf, 45 * N, 1.0, 0.99) == 45 * N
    torch.cuda.synchronize()
    assert bool((big_s[-64:] == 7).all()) and bool((big_o[-64:] == 7).all()), "stray write behind the ring"
    assert bool((buf.other[:45, :, 1] == 0).sum() == N), "every lane ends exactly one 40-step episode in 45 lock-steps"
    assert bool((buf.other[45:] == 0).all()), "slots beyond the explored ones must be untouched"
    for e in envs:
        e.close()
