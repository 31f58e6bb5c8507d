"""Evaluation on the fused kernel (csrc/rollout_eval.hip, `pime_rollout_eval`): the evaluator's deterministic
episode (/root/reference/elegantrl/run.py:600-619) and the set-point step-response protocols (utils/test.py:1369-1407,209-349) as
ONE launch each, against (a) the reference's golden protocol records (float64 state mode, the tolerances of
tests/test_gpu_facade.py::test_batched_step_response_protocols), (b) the CPU oracle stepped with the oracle's own policy forward,
and (c) the launch-by-launch path it replaces."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from rollout_replay import DEV, make_agent, oracle_mean

pytestmark = pytest.mark.gpu


def _no_stepwise(env):
    """The fused path must not fall back: any step-per-launch call on this env fails the test."""
    def boom(*a, **k):
        raise AssertionError("the protocol / evaluation fell back to step-per-launch kernels")
    env.step = env.step_residual = boom


def test_protocols_run_as_one_launch_and_match_the_golden_records():
    from pime_amd import gym_control, protocols
    g = load_golden("ph_stepresponse.npz")
    env = gym_control.make_vec(gym_control.PH_V35, 2, device=DEV, state_mode="f64", seed=0)
    _no_stepwise(env)
    res = protocols.ph_step_response(env, plants=[g["nominal_params"], g["corner_params"]])
    for lane, tag in enumerate(("nominal", "corner")):
        np.testing.assert_allclose(res["action"][:, lane], g[tag + "_act"], rtol=0, atol=1e-12)
        np.testing.assert_allclose(res["y"][:, lane], g[tag + "_y"], rtol=0, atol=1e-11)
        np.testing.assert_allclose(res["I"][:, lane], g[tag + "_I"], rtol=0, atol=1e-10)
        np.testing.assert_allclose(res["x"][:, lane], g[tag + "_x"], rtol=1e-12)
        np.testing.assert_allclose(res["reward"][:, lane], g[tag + "_rew"], rtol=2e-7, atol=1e-6)
    env.close()
    gw = load_golden("wt_stepresponse.npz")
    env = gym_control.make_vec(gym_control.WT_INTEGRATOR, 2, device=DEV, state_mode="f64", seed=0, reward_type="distance",
                               noise_scale=0.0)
    _no_stepwise(env)
    res = protocols.wt_step_response(env, steps=500, plants=[gw["robust1_params"][:3], gw["robust3_params"][:3]])
    for lane, tag in enumerate(("robust1", "robust3")):
        np.testing.assert_allclose(res["obs"][:, lane], gw[tag + "_obs"], rtol=1e-11, atol=1e-12)
        np.testing.assert_allclose(res["action"][:, lane], gw[tag + "_act"], rtol=0, atol=1e-12)
    env.close()


@pytest.mark.parametrize("env_name,algo,md", [("PH_V35", "ResidualIntegratorModularPPO", 128), ("PH_V35", "ResidualPPO", 64),
                                              ("WT_INTEGRATOR", "ResidualIntegratorModularPPO", 128),
                                              # width 256: the streamed rollout kernel in evaluation mode
                                              ("WT_INTEGRATOR", "ResidualIntegratorModularPPO", 256), ("PH_V35", "ResidualPPO", 256)])
def test_fused_episode_returns_against_oracle_and_stepwise(env_name, algo, md):
    """One deterministic episode per lane: the fused launch vs (1) the oracle env driven by the oracle's forward of the same
    weights, step by step in float64 (per-lane returns 1e-4 relative: float32 policy mean + mixed-mode state), (2) the
    launch-by-launch evaluator path on an identically seeded env."""
    import oracle
    from pime_amd import gym_control
    from pime_amd.elegantrl.run import get_episode_return_vec
    is_ph = env_name == "PH_V35"
    N, seed, off = 2048, 13, 512
    kw = {} if is_ph else dict(reward_type="distance", max_step=80)
    envs = [gym_control.make_vec(getattr(gym_control, env_name), N, device=DEV, state_mode="mixed", seed=seed, env_offset=off, **kw)
            for _ in range(2)]
    ag = make_agent(algo, envs[0], md)
    fused = ag.fused_eval_policy(envs[0])
    assert fused is not None, "the fused evaluation kernel must serve this configuration"
    _no_stepwise(envs[0])
    got = get_episode_return_vec(envs[0], ag.act, fused=fused)
    slow = get_episode_return_vec(envs[1], ag.act)
    T = envs[0].max_step
    ref = oracle.OraclePH(N, oracle.ph_table(), seed=seed, env_offset=off) if is_ph else \
        oracle.OracleWT(N, max_steps=T, reward_type="distance", seed=seed, env_offset=off)
    sd = {k: v.detach().cpu().numpy() for k, v in ag.act.state_dict().items()}
    priorK = ag._rollout_priorK()
    obs = ref.reset()
    want = np.zeros(N)
    for _ in range(T):
        act = oracle.residual_action(oracle_mean(algo, obs, sd).astype(np.float32), obs, priorK)
        obs, _, rew, _ = ref.step(act)
        want += rew.astype(np.float32).astype(np.float64)
    # pH: a float32-rounding difference of the policy mean (3e-5) moves C*x*1e5 by ~0.08 of a titration cell per step, so over an
    # episode some lanes read a neighbouring cell now and then (|dy| <= 0.03 at the steepest point, ~0.001 elsewhere) and on the
    # steep part of the curve the closed loop carries that on: ~1 % of the lanes differ by 1e-4 .. 2e-3 of their return (observed),
    # the rest agree to rounding.  The tank is a smooth map: every lane to 1e-4.
    for other in (want, slow):
        rel = np.abs(got - other) / np.abs(other)
        if is_ph:
            assert (rel <= 1e-4).mean() >= 0.97 and rel.max() <= 0.05, (float((rel <= 1e-4).mean()), float(rel.max()))
        else:
            np.testing.assert_allclose(got, other, rtol=1e-4, atol=1e-3)
    assert np.median(np.abs(got - want) / np.abs(want)) < 2e-5
    for e in envs:
        e.close()


@pytest.mark.parametrize("algo,md,N", [("ResidualIntegratorModularPPO", 128, 2048), ("ResidualPPO", 64, 2048),
                                       ("ResidualIntegratorModularPPO", 128, 6144)])   # 6 144 lanes: the 16-lane-tile (non-QUAD) instantiation
def test_fused_ph_evaluation_replays_cell_exact_through_the_oracle(algo, md, N):
    """The pH evaluation episode pinned lane by lane (the return comparison above is statistical for pH
    because a float32 rounding difference of the policy MEAN can move a lane into the neighbouring titration cell).  The kernel's
    trace mode records every step's observation, env action, reward and plant state; the recorded ACTIONS are replayed through
    OraclePH (as tests/rollout_replay.py does for the training rollout): every lane reads the oracle's titration cell at every step
    (float32 y bit-equal), x to 1e-12 relative, rewards and returns to 2e-5; and the recorded action itself is the oracle's forward of
    the recorded observation + prior term to 3e-5 at every step."""
    import oracle
    from pime_amd import gym_control
    seed, off = 13, 512
    env = gym_control.make_vec(gym_control.PH_V35, N, device=DEV, state_mode="mixed", seed=seed, env_offset=off)
    ag = make_agent(algo, env, md)
    fused = ag.fused_eval_policy(env)
    assert fused is not None and env.eval_supported(fused[0], trace=True)
    _no_stepwise(env)
    T = env.max_step
    env.reset()
    ret, tr = env.rollout_eval(fused[0], fused[1], T, want_trace=True)
    torch.cuda.synchronize()
    ret, tr = ret.cpu().numpy(), tr.cpu().numpy()      # tr [T, 6, N]: y, r, I before the step | env action, reward, x after
    ref = oracle.OraclePH(N, oracle.ph_table(), seed=seed, env_offset=off)
    sd = {k: v.detach().cpu().numpy() for k, v in ag.act.state_dict().items()}
    priorK = ag._rollout_priorK()
    obs = ref.reset()
    want_ret = np.zeros(N)
    for t in range(T):
        seen = tr[t, 0:3].T.astype(np.float32)                              # what the policy saw (float32 observation)
        np.testing.assert_array_equal(seen[:, 0], obs[:, 0].astype(np.float32), err_msg=f"titration cell, step {t}")
        np.testing.assert_allclose(seen, obs, rtol=2e-5, atol=2e-5, err_msg=f"observation, step {t}")
        want_act = oracle.residual_action(oracle_mean(algo, seen, sd).astype(np.float32), seen, priorK)
        np.testing.assert_allclose(tr[t, 3], want_act, rtol=0, atol=3e-5, err_msg=f"policy forward + prior term, step {t}")
        obs, _, rew, _ = ref.step(tr[t, 3])                                  # the RECORDED env action
        np.testing.assert_allclose(tr[t, 5], ref.get("x"), rtol=1e-12, err_msg=f"plant state x, step {t}")
        np.testing.assert_allclose(tr[t, 4], rew, rtol=2e-5, atol=2e-5, err_msg=f"reward, step {t}")
        want_ret += rew.astype(np.float32).astype(np.float64)
    np.testing.assert_allclose(ret, want_ret, rtol=2e-5, atol=1e-4)
    np.testing.assert_allclose(ret, tr[:, 4].sum(0), rtol=1e-12)
    env.close()


def test_evaluator_uses_the_fused_path_and_explore_resets_afterwards():
    """train_and_evaluate's evaluator on the shared vectorised env: evaluation = reset + one launch; the next explore_env must
    start from a fresh reset (the evaluation leaves the lanes mid-episode)."""
    from pime_amd import gym_control
    from pime_amd.elegantrl.run import Evaluator, make_buffer
    N = 512
    env = gym_control.make_vec(gym_control.PH_V35, N, device=DEV, state_mode="mixed", seed=2)
    ag = make_agent("ResidualIntegratorModularPPO", env, 128)
    buf = make_buffer(ag, env, N * 50)
    ag.explore_env(env, buf, N * 50, 1.0, 0.99)
    assert env.fresh
    ev = Evaluator(cwd="/tmp", agent_id=0, eval_times1=N, eval_times2=N, eval_gap=1, env=env, device=DEV, is_main=False)
    calls = []
    orig = env.rollout_eval
    env.rollout_eval = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    r = ev._returns(ev._policy(ag), N, ag)
    assert calls == [1] and r.shape == (N,) and np.isfinite(r).all()
    assert not env.fresh
    ep0 = env.get_field("episode").copy()
    ag.explore_env(env, buf, N * 50, 1.0, 0.99)      # resets (episode + 1), rolls out one episode, auto-resets (episode + 1)
    assert (env.get_field("episode") == ep0 + 2).all() and env.fresh
    env.close()


def test_stacking10_width_256_evaluation_is_one_launch():
    """The reference's live water-tank configuration (ResidualPPO, net_dim 256, Stacking10; run_watertank_changing.sh:20-27): the
    evaluator's episode through the width-256 kernel's evaluation mode against the launch-by-launch path, and its TRACE mode
    (VERDICT r03 task 6): the traced levels / rewards / actions of the same launch against the step-per-launch records."""
    from pime_amd import gym_control
    from pime_amd.elegantrl.run import get_episode_return_vec
    N = 1000
    envs = [gym_control.make_vec(gym_control.WT_STACKING.format(10), N, device=DEV, state_mode="mixed", seed=4, reward_type="distance",
                                 max_step=60) for _ in range(3)]
    ag = make_agent("ResidualPPO", envs[0], 256)
    fused = ag.fused_eval_policy(envs[0])
    assert fused is not None and envs[0].eval_supported(fused[0], trace=True), "the width-256 evaluation must serve a trace"
    assert not envs[0].eval_supported(fused[0], trace=True, schedule=True)    # (no protocol schedule on a Stacking observation)
    _no_stepwise(envs[0])
    got = get_episode_return_vec(envs[0], ag.act, fused=fused)
    slow = get_episode_return_vec(envs[1], ag.act)
    np.testing.assert_allclose(got, slow, rtol=1e-4, atol=1e-3)
    # trace: one launch records (h1, h2, r, I after the step | reward, env action) of every step
    envs[2].reset()
    ret, tr = envs[2].rollout_eval(fused[0], fused[1], 60, want_trace=True)
    torch.cuda.synchronize()
    tr = tr.cpu().numpy()
    np.testing.assert_allclose(ret.cpu().numpy(), got, rtol=1e-6, atol=1e-5)
    np.testing.assert_allclose(tr[:, 4].sum(0), got, rtol=1e-6, atol=1e-4)
    assert np.isfinite(tr).all() and (tr[:, 0] >= 0).all() and (np.abs(tr[:, 5]) < 5).all()
    np.testing.assert_allclose(tr[-1, 0], envs[2].get_field("h1"), rtol=1e-6)
    np.testing.assert_allclose(tr[-1, 1], envs[2].get_field("h2"), rtol=1e-6)
    for e in envs:
        e.close()


def test_width_256_step_response_protocol_is_one_launch_and_matches_the_golden_records():
    """utils/test.py:209-349 (r = 3, 6, 9, 4, 2 x 500 steps, noise 0, the robust_test.py plants) through a WIDTH-256 agent on the
    Integrator observation -- ResidualIntegratorModularPPO net_dim 256, the commented block of run_watertank_changing.sh:11-18 --
    whose output layer is zero (agent_residual.py:45-50: the initial policy IS the prior controller): one launch of the streamed
    kernel's evaluation mode with the set-point schedule and the trace, against the reference's golden protocol records.  The kernel
    keeps float32 state (PIME_STATE_MIXED): levels to 2e-3 of the 10-unit range over 2 500 steps, actions to 2e-3."""
    from pime_amd import gym_control, protocols
    gw = load_golden("wt_stepresponse.npz")
    env = gym_control.make_vec(gym_control.WT_INTEGRATOR, 2, device=DEV, state_mode="mixed", seed=0, reward_type="distance",
                               noise_scale=0.0)
    torch.manual_seed(0)
    from pime_amd.utils import MODELS
    ag = MODELS["residualintegratormodularppo"](device=DEV)
    ag.init(256, env.state_dim, 1, 1)
    ag.init_residual({"init_K": env.K.reshape(-1, 1)})      # zero output layer: residual 0
    fused = ag.fused_eval_policy(env)
    assert fused is not None and fused[0].md == 256 and env.eval_supported(fused[0], trace=True, schedule=True)
    _no_stepwise(env)
    res = protocols.wt_step_response(env, steps=500, plants=[gw["robust1_params"][:3], gw["robust3_params"][:3]], agent=ag)
    for lane, tag in enumerate(("robust1", "robust3")):
        np.testing.assert_allclose(res["obs"][:, lane, :3], gw[tag + "_obs"][:, :3], rtol=2e-3, atol=2e-2)
        np.testing.assert_allclose(res["action"][:, lane], gw[tag + "_act"], rtol=0, atol=1e-2)
    env.close()


def test_step_response_segments_draw_fresh_process_noise():
    """the fused protocol restarted t at every set-point segment without starting a new episode, so the water tank's
    process noise -- Philox keyed on (lane, episode, t) -- repeated segment 0's sequence in every segment.  A boundary now counts
    as the env.reset() it stands for: the fused protocol must equal the step-per-launch protocol (which does reset per segment, i.e.
    draws episode e + k's noise in segment k) lane by lane, with a process noise 5x the registered one so that a shared sequence
    would show at once."""
    from pime_amd import gym_control, protocols
    N, steps = 64, 40
    kw = dict(device=DEV, state_mode="f64", seed=3, reward_type="distance", noise_scale=0.05)
    env = gym_control.make_vec(gym_control.WT_INTEGRATOR, N, **kw)
    fused = protocols.wt_step_response(env, setpoints=(5.0, 5.0, 5.0), steps=steps)
    env.close()
    env = gym_control.make_vec(gym_control.WT_INTEGRATOR, N, **kw)
    slow = protocols.wt_step_response(env, policy=protocols._prior(env), setpoints=(5.0, 5.0, 5.0), steps=steps)
    np.testing.assert_allclose(fused["obs"], slow["obs"], rtol=1e-9, atol=1e-9)
    env.close()
