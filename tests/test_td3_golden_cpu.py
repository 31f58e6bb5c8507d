"""AgentTD3.update_net (twin critics, target policy smoothing, delayed soft target updates) against the REFERENCE's own
update (/root/reference/elegantrl/agent.py:276-341, run by tests/golden/make_golden.py:golden_td3_update): same start,
same flat ring buffer, torch seeded alike right before the call -> the same sampled rows and smoothing noise, so actor,
critic and both target nets must come out equal (2e-6: CPU float32, same op order).  Plus the host logic of the
vectorised off-policy path: the per-lane device ring and its (i, i + N) successor rule, and the residual-TD3 composition
(which has no reference counterpart, SURVEY.md fact 5: its PARTS are what this file and nets.npz pin)."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle.cpu_stack import OracleBackend, OracleVecEnv


def _sd(g, prefix):
    return {k[len(prefix) + 1:]: torch.from_numpy(g[k].copy()) for k in g.files if k.startswith(prefix + ".")}


def test_td3_update_matches_reference():
    from pime_amd.elegantrl.agent import AgentTD3
    from pime_amd.elegantrl.replay import ReplayBuffer
    g = load_golden("td3_update.npz")
    net_dim, target_step, batch, repeat = (int(v) for v in g["td3:hyper"][:4])
    ag = AgentTD3(backend=OracleBackend(), device="cpu")
    ag.init(net_dim, 4, 1)
    ag.act.load_state_dict(_sd(g, "td3:act0"))
    ag.cri.load_state_dict(_sd(g, "td3:cri0"))
    ag.act_target.load_state_dict(_sd(g, "td3:act0"))
    ag.cri_target.load_state_dict(_sd(g, "td3:cri0"))
    buf = ReplayBuffer(len(g["td3:state"]) + 8, 4, 1, if_on_policy=False, device="cpu")
    buf.extend_buffer(g["td3:state"], g["td3:other"])
    torch.manual_seed(77)
    obj_a, obj_c = ag.update_net(buf, target_step, batch, repeat)
    for tag, net in (("act1", ag.act), ("cri1", ag.cri), ("act_target1", ag.act_target), ("cri_target1", ag.cri_target)):
        want = _sd(g, f"td3:{tag}")
        got = net.state_dict()
        assert set(got) == set(want)
        for k in want:
            np.testing.assert_allclose(got[k].numpy(), want[k].numpy(), rtol=0, atol=2e-6, err_msg=f"{tag}.{k}")
    np.testing.assert_allclose([obj_a, obj_c], g["td3:obj"], rtol=1e-5, atol=1e-6)


def test_vec_replay_ring_successor_rule():
    from pime_amd.elegantrl.replay import VecReplayBuffer
    N, D = 8, 3
    b = VecReplayBuffer(5 * N, N, D, 1, "cpu")
    assert b.slots == 5
    for t in range(7):   # wraps: slots 5, 6 overwrite 0, 1
        s = torch.full((N, D), float(t)) + torch.arange(N).reshape(N, 1) * 0.01
        b.append_step(s, torch.full((N,), -float(t)), torch.full((N,), 0.99), torch.full((N, 1), 0.5))
    b.update_now_len_before_sample()
    assert b.if_full and b.next_slot == 2 and b.now_len == 5 * N
    torch.manual_seed(0)
    r, m, a, s, s2 = b.sample_batch(4096)
    t0, t1 = s[:, 0].round(), s2[:, 0].round()
    lane0, lane1 = ((s[:, 0] - t0) * 100).round(), ((s2[:, 0] - t1) * 100).round()
    assert torch.equal(lane0, lane1), "successor must be the SAME lane"
    assert torch.equal(t1, t0 + 1), "successor must be the next time step"
    assert set(t0.tolist()) == {2.0, 3.0, 4.0, 5.0}, "newest step (6) has no successor and must not be sampled"
    np.testing.assert_allclose(r[:, 0].numpy(), -t0.numpy())


def test_residual_td3_composition_and_vector_loop():
    """Zero-initialised residual => the env sees the prior controller (+ clipped exploration noise); the update moves the
    actor, soft-updates both targets, and the buffer holds the RESIDUAL action."""
    from pime_amd.elegantrl.agent_residual import AgentResidualTD3
    from pime_amd.elegantrl.run import make_buffer
    from pime_amd.utils import IF_ONPOLICY, MODELS
    assert MODELS["residualtd3"] is AgentResidualTD3 and IF_ONPOLICY["residualtd3"] is False
    N = 64
    env = OracleVecEnv("wt", N, seed=3, reward_type="distance", max_steps=30)
    torch.manual_seed(1)
    ag = AgentResidualTD3(backend=OracleBackend(), device="cpu")
    ag.init(32, env.state_dim, 1)
    ag.init_residual({"init_K": env.K.reshape(-1, 1)})
    obs = env.reset()
    with torch.no_grad():
        np.testing.assert_allclose(ag.eval_policy(obs).numpy(), (obs @ torch.tensor(-env.K, dtype=torch.float32).reshape(-1, 1)).numpy(),
                                   atol=1e-7)   # policy == prior at initialisation
        assert float(ag.act(obs).abs().max()) == 0.0
    buf = make_buffer(ag, env, 40 * N)
    steps = ag.explore_env(env, buf, 35 * N, 1.0, 0.99)
    assert steps == 35 * N and buf.stored_slots == 35
    act_rows = buf.other[:35, :, 2]
    assert float(act_rows.abs().max()) <= 1.0 and float(act_rows.std()) == pytest.approx(0.1, rel=0.15)   # explore noise only
    assert float((buf.other[:35, :, 1] == 0).float().sum()) == N      # every lane ended exactly one 30-step episode
    before = {k: v.clone() for k, v in ag.act.state_dict().items()}
    tgt_before = {k: v.clone() for k, v in ag.cri_target.state_dict().items()}
    oa, oc = ag.update_net(buf, 35 * N, 128, 1)      # 35 optimizer steps
    assert np.isfinite(oa) and np.isfinite(oc)
    assert any(not torch.equal(v, before[k]) for k, v in ag.act.state_dict().items() if k != "priorK")
    assert torch.equal(ag.act.priorK, before["priorK"])
    assert any(not torch.equal(v, tgt_before[k]) for k, v in ag.cri_target.state_dict().items())


def _td3_train_args(env, tmp_path, env_eval=None):
    from pime_amd.elegantrl.agent_residual import AgentResidualTD3
    from pime_amd.elegantrl.run import Arguments
    N = env.num_envs
    env.env_name, env.target_return = "wt-oracle", 1e9
    args = Arguments(if_on_policy=False)
    args.agent = AgentResidualTD3(backend=OracleBackend(), device="cpu")
    args.env, args.env_eval = env, env_eval
    args.cwd, args.if_remove = str(tmp_path / "run"), False
    args.net_dim, args.batch_size, args.repeat_times = 32, 64, 1
    args.target_step, args.max_memo = 7 * N, 64 * N       # 7 lock-steps per explore call: the evaluations fall mid-episode
    args.break_step = 6 * 7 * N
    args.eval_gap, args.eval_times1, args.eval_times2 = 2, N, N
    args.num_threads, args.random_seed = 1, 3
    args.residual_kwargs = {"init_K": env.K.reshape(-1, 1)}
    args.if_residual, args.fix_K = True, True
    return args


def _assert_successor_rows(buf, stored, integral_max=25.0):
    """Every stored transition with mask != 0 must be followed, one slot later in the same lane, by ITS env successor: same
    episode (same set-point r) and the integrator advanced by that step's error, clip(I + (r - h2'), +-25)
    (nonlinear_watertank.py:822-825).  A row whose successor slot holds a reset observation must carry mask 0."""
    s, m = buf.state[:stored].double(), buf.other[:stored, :, 1]
    cont = m[:-1] != 0
    assert cont.any()
    same_r = s[1:, :, 2] == s[:-1, :, 2]
    want_I = (s[:-1, :, 3] + (s[:-1, :, 2] - s[1:, :, 1])).clamp(-integral_max, integral_max)
    ok_I = (s[1:, :, 3] - want_I).abs() <= 1e-5
    assert bool((same_r & ok_I)[cont].all()), "a stored transition continues into a row that is not its successor"


def test_td3_train_loop_keeps_successor_rows_across_evaluations(tmp_path):
    """the evaluator resets the env it is given.  (a) train_and_evaluate gives an off-policy agent on a
    vectorised env its own evaluation env (a clone), so the running episodes are never cut; (b) if a caller shares the env
    anyway, explore_vec_env notices the foreign reset, starts new episodes and cuts the newest stored step off (mask 0)."""
    from pime_amd.elegantrl.run import get_episode_return_vec, make_buffer, train_and_evaluate
    N = 16
    env = OracleVecEnv("wt", N, seed=3, reward_type="distance", max_steps=30)
    ag, buf = train_and_evaluate(_td3_train_args(env, tmp_path))
    stored = buf.stored_slots
    assert stored == 6 * 7 and not buf.if_full
    _assert_successor_rows(buf, stored)
    ends = (buf.other[:stored, :, 1] == 0).sum().item()
    assert ends == N * (stored // 30), "with its own evaluation env no episode of the training env is cut short"
    # (b) a shared env: evaluate on the training env between two explore calls
    env2 = OracleVecEnv("wt", N, seed=4, reward_type="distance", max_steps=30)
    args = _td3_train_args(env2, tmp_path)
    ag2 = args.agent
    ag2.init(32, env2.state_dim, 1)
    ag2.init_residual(args.residual_kwargs)
    buf2 = make_buffer(ag2, env2, 64 * N)
    ag2.explore_env(env2, buf2, 7 * N, 1.0, 0.99)
    get_episode_return_vec(env2, ag2.eval_policy)          # resets env2 and runs it a full episode
    ag2.explore_env(env2, buf2, 7 * N, 1.0, 0.99)
    assert buf2.stored_slots == 14
    assert bool((buf2.other[6, :, 1] == 0).all()), "the step in front of the foreign reset must be cut (mask 0)"
    assert bool((buf2.other[:6, :, 1] != 0).all())
    _assert_successor_rows(buf2, 14)


def test_td3_first_update_with_one_lock_step(tmp_path):
    """target_step < 2 * num_envs used to store ONE lock-step and then assert in the sampler."""
    from pime_amd.elegantrl.agent_residual import AgentResidualTD3
    from pime_amd.elegantrl.run import make_buffer
    N = 16
    env = OracleVecEnv("wt", N, seed=3, reward_type="distance", max_steps=30)
    ag = AgentResidualTD3(backend=OracleBackend(), device="cpu")
    ag.init(32, env.state_dim, 1)
    ag.init_residual({"init_K": env.K.reshape(-1, 1)})
    buf = make_buffer(ag, env, 8 * N)
    assert ag.explore_env(env, buf, N, 1.0, 0.99) == 2 * N and buf.stored_slots == 2
    oa, oc = ag.update_net(buf, N, 32, 1)
    assert np.isfinite(oa) and np.isfinite(oc)
    assert ag.explore_env(env, buf, N, 1.0, 0.99) == N and buf.stored_slots == 3
