"""BASELINE.json config 5 end to end on one GPU: "Mixed pH + water-tank batch, domain-randomised ensemble sweep, fp16 state".

One rank's slice of the 8-GPU job (SURVEY.md section 8d cfg 5: 131 072 lanes / 8 ranks = 16 384 per rank, half pH and half water
tank): 8 192 pH lanes + 8 192 Integrator-tank lanes in `state_mode="mixed16"` (integrated error stored as binary16; observation
and reward rows binary16; float32 / float64 arithmetic), ensemble ranges 1.5x the registered widths, a non-zero lane offset
(rank 3).  The two halves run side by side on two HIP streams, each through the FUSED rollout kernel (`pime_rollout_h`: policy
forward on the matrix cores + exploration noise + residual env step + in-kernel auto-reset, binary16 rows straight into the
trajectory buffer), then each agent's `update_net` consumes its binary16 trajectory (widened once per update) on the fused
gradient kernels -- the path `bench.py --workload mixed16` times.

Parity: what the kernel stored is replayed through the fp64 oracle with the recorded pre-tanh actions, under the binary16
STORAGE tolerance of tests/test_gpu_env_fp16.py (include/pime_hip.h: every stored word = the float32 value rounded to nearest
binary16, relative 2^-11):
  * exploration noise bit-equal to the oracle's Philox draw; policy mean 3e-5 on the binary16 observation the kernel saw;
  * pH: y is the oracle's titration cell rounded once (x and the plant never touch binary16), r one rounding; the integrated
    error one rounding per step on top of the stored value it continued from (the oracle is re-synchronised to the stored I, as
    the device continues from it); reward within the rounding of its magnitude;
  * tank: levels rtol 2^-11 + 2e-3 (float32 Euler sub-steps, one rounding), I and reward likewise;
  * first observation after the in-kernel auto-reset (new ensemble draw from the widened ranges): the oracle's float32
    observation rounded once, bit for bit."""
import numpy as np
import pytest
import torch

from rollout_replay import DEV, make_agent, oracle_mean

pytestmark = pytest.mark.gpu
H = 2.0 ** -11
LANES, RANK, SEED = 8192, 3, 17
WIDE_PH = dict(qww_V=(0.0045, 0.0165), qc_V=(0.00125, 0.00275))
WIDE_WT = dict(a1=(0.0012, 0.0027), a2=(0.0012, 0.0027), Kp=(0.045, 0.195))
ALGO = "ResidualIntegratorModularPPO"


def _replay_half(ag, env, buf, ref, episodes, offset, is_ph):
    import oracle
    N, T = env.num_envs, env.max_step
    n = episodes * T
    assert buf.state.dtype == torch.float16 and buf.reward.dtype == torch.float16
    priorK = ag._rollout_priorK()
    sd = {k: v.detach().cpu().numpy() for k, v in ag.act.state_dict().items()}
    sigma = np.float32(np.exp(sd["a_std_log"][0, 0]))
    state = buf.state[:n + 1].cpu().numpy()            # float16
    action = buf.action[:n, :, 0].cpu().numpy()
    noise = buf.noise[:n, :, 0].cpu().numpy()
    reward = buf.reward[:n].cpu().numpy().astype(np.float64)
    done = buf.done[:n].cpu().numpy()
    obs = ref.reset()
    np.testing.assert_array_equal(state[0], obs.astype(np.float16))
    ref.set("I", state[0][:, -1].astype(np.float64))
    off_cell = 0
    for t in range(n):
        ep, tt = divmod(t, T)
        seen = state[t].astype(np.float32)             # the binary16 observation the policy and the prior term saw
        want_eps = oracle.explore_noise(ag._rollout_seed, offset, N, 1 + ep, tt)
        assert (noise[t] != want_eps).mean() <= 1e-4
        np.testing.assert_allclose(noise[t], want_eps, rtol=1.2e-7, atol=0)
        np.testing.assert_allclose(action[t] - noise[t] * sigma, oracle_mean(ALGO, seen, sd), rtol=3e-5, atol=3e-5)
        act = oracle.residual_action(action[t], seen, priorK)
        obs, _, rew, d = ref.step(act, auto_reset=True)
        assert bool(d.all()) == (tt == T - 1)
        np.testing.assert_array_equal(done[t].astype(bool), d)
        got = state[t + 1].astype(np.float64)
        np.testing.assert_allclose(reward[t], rew, rtol=2 * H, atol=3e-3, err_msg=f"reward, step {t}")
        if tt == T - 1:      # first observation of the next episode: the oracle's float32 row rounded once
            np.testing.assert_array_equal(state[t + 1], obs.astype(np.float16))
        elif is_ph:
            dy = np.abs(got[:, 0] - obs[:, 0])
            off_cell += int((dy > H * np.abs(obs[:, 0]) + 1e-6).sum())
            np.testing.assert_allclose(got[:, 1], obs[:, 1], rtol=H, atol=0)
            np.testing.assert_allclose(got[:, 2], obs[:, 2], rtol=H, atol=1e-5, err_msg=f"integrated error, step {t}")
        else:
            np.testing.assert_allclose(got[:, :3], obs[:, :3], rtol=H, atol=2e-3, err_msg=f"levels / set-point, step {t}")
            np.testing.assert_allclose(got[:, 3], obs[:, 3], rtol=H, atol=2e-3 + 1e-5, err_msg=f"integrated error, step {t}")
        ref.set("I", got[:, -1])                       # the device continues from the STORED (rounded) integrator
    assert off_cell == 0, f"{off_cell} pH lane-steps read another titration cell than the oracle"


def test_config5_mixed16_rank_slice_rollout_and_update():
    import oracle
    from pime_amd import gym_control
    from pime_amd.elegantrl.run import make_buffer
    off = RANK * LANES
    ph = gym_control.make_vec(gym_control.PH_V35, LANES, device=DEV, state_mode="mixed16", seed=SEED, env_offset=off, **WIDE_PH)
    from pime_amd.vec_env import VecWaterTank
    wt = VecWaterTank(LANES, device=DEV, state_mode="mixed16", seed=SEED, env_offset=off, reward_type="distance", **WIDE_WT)
    assert ph.trajectory_dtype == wt.trajectory_dtype == torch.float16 and wt.max_step == 200
    ag_ph, ag_wt = make_agent(ALGO, ph, 128), make_agent(ALGO, wt, 128)
    assert ag_ph._fused_rollout_ok(ph) and ag_wt._fused_rollout_ok(wt), "config 5 must take the fused rollout kernel"
    b_ph, b_wt = make_buffer(ag_ph, ph, 2 * LANES * 50), make_buffer(ag_wt, wt, LANES * 200)
    s_ph, s_wt = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    with torch.cuda.stream(s_ph):      # the two halves of the batch advance side by side
        assert ag_ph.explore_env(ph, b_ph, 2 * LANES * 50, 1.0, 0.99) == 2 * LANES * 50
    with torch.cuda.stream(s_wt):
        assert ag_wt.explore_env(wt, b_wt, LANES * 200, 1.0, 0.99) == LANES * 200
    torch.cuda.synchronize()
    r_ph = oracle.OraclePH(LANES, oracle.ph_table(), seed=SEED, env_offset=off)
    r_ph.set_ranges(WIDE_PH["qww_V"], WIDE_PH["qc_V"])
    r_wt = oracle.OracleWT(LANES, max_steps=200, reward_type="distance", seed=SEED, env_offset=off)
    r_wt.set_ranges(WIDE_WT["a1"], WIDE_WT["a2"], WIDE_WT["Kp"])
    _replay_half(ag_ph, ph, b_ph, r_ph, 2, off, True)
    _replay_half(ag_wt, wt, b_wt, r_wt, 1, off, False)
    np.testing.assert_array_equal(ph.get_field("qww_V"), r_ph.get("qww_V"))       # the widened ensemble, resampled in-kernel
    assert ph.get_field("qww_V").min() < 0.005 and ph.get_field("qww_V").max() > 0.015
    np.testing.assert_allclose(wt.get_field("Kp"), r_wt.get("Kp"), rtol=1e-7)
    I_dev = ph.get_field("I")
    assert np.array_equal(I_dev, I_dev.astype(np.float16).astype(np.float64)), "the handle's I must be a binary16 value"
    # the update consumes the binary16 trajectory (widened once) on the fused gradient kernels; identical to an update on a
    # float32 buffer holding the same (widened) rows
    from pime_amd.elegantrl.replay import TrajectoryBuffer
    for ag, env, buf, n in ((ag_ph, ph, b_ph, 2 * LANES * 50), (ag_wt, wt, b_wt, LANES * 200)):
        # a float32 TrajectoryBuffer holding the widened rows, and an identical agent (weights, fresh optimizer state) to update on it
        wide = TrajectoryBuffer(buf.horizon, buf.num_envs, buf.state_dim, buf.action_dim, DEV)
        wide.state.copy_(buf.state.float()); wide.reward.copy_(buf.reward.float())
        wide.mask.copy_(buf.mask); wide.action.copy_(buf.action); wide.noise.copy_(buf.noise); wide.done.copy_(buf.done)
        wide.length = buf.length
        twin = make_agent(ALGO, env, 128)
        twin.act.load_state_dict(ag.act.state_dict()); twin.cri.load_state_dict(ag.cri.state_dict())
        twin.weights_changed()
        w0 = torch.cat([p.detach().reshape(-1) for p in ag.act.parameters()]).clone()
        torch.manual_seed(5)
        oa, oc = ag.update_net(buf, n, 65536, 2)
        torch.manual_seed(5)
        oa2, oc2 = twin.update_net(wide, n, 65536, 2)
        torch.cuda.synchronize()
        assert ag._packed.get("fused") and twin._packed.get("fused"), "update_net did not take the fused HIP gradient path"
        w1 = torch.cat([p.detach().reshape(-1) for p in list(ag.act.parameters()) + list(ag.cri.parameters())])
        w2 = torch.cat([p.detach().reshape(-1) for p in list(twin.act.parameters()) + list(twin.cri.parameters())])
        assert np.isfinite(oa) and np.isfinite(oc) and torch.isfinite(w1).all()
        assert not torch.equal(w0, torch.cat([p.detach().reshape(-1) for p in ag.act.parameters()]))
        assert torch.equal(w1, w2), \
            "the update from the binary16 buffer must be bit-equal to the update from a float32 buffer holding the widened rows"
        np.testing.assert_allclose([oa, oc], [oa2, oc2], rtol=1e-5)   # (the LOGGED loss sums are float atomics: order-dependent last bits)
    ph.close(); wt.close()


@pytest.mark.parametrize("tiles", ["wide", "narrow", "quad"])
def test_config5_stepwise_half_rollout_matches_fused(tiles, monkeypatch):
    """The launch-by-launch form (policy forward + pime_env_step_residual_h per lock-step, what a shape without a fused rollout
    takes) writes the same binary16 trajectory as the fused kernel when fed the same exploration noise.  32-lane tiles run the
    forward kernel's own MFMA chain; the default 16-lane tiles sum in another order, and a last-bit difference of a policy mean now
    and then moves a binary16 row by one ulp (1e-3 relative), which the next policy means see: looser bounds there."""
    monkeypatch.setenv("PIME_ROLLOUT_NARROW", {"wide": "0", "narrow": "1", "quad": "2"}[tiles])
    from pime_amd import gym_control
    from pime_amd.elegantrl.run import make_buffer
    N = 1024
    bufs = []
    for fused in (True, False):
        env = gym_control.make_vec(gym_control.PH_V35, N, device=DEV, state_mode="mixed16", seed=4, **WIDE_PH)
        ag = make_agent(ALGO, env, 128)
        ag.use_fused_rollout = fused
        buf = make_buffer(ag, env, N * 50)
        if not fused:
            ag.noise_hook = lambda t, shape: bufs[0].noise[t].reshape(shape)
        ag.explore_env(env, buf, N * 50, 1.0, 0.99)
        torch.cuda.synchronize()
        bufs.append(buf)
        env.close()
    a, b = bufs
    assert torch.equal(a.done, b.done)
    # the step-wise policy mean comes from mlp_forward_kernel, the fused one from the rollout kernel's own chain: same weights,
    # same order of operations up to the MFMA accumulation order -> actions agree to rounding, rows to a binary16 ulp
    assert float((a.action - b.action).abs().max()) <= (3e-5 if tiles == "wide" else 5e-4)
    same = (a.state[:50] == b.state[:50]).float().mean().item()
    assert same > (0.995 if tiles == "wide" else 0.99), f"only {same:.4f} of the binary16 rows are bit-equal"
    np.testing.assert_allclose(a.state[:50].float().cpu().numpy(), b.state[:50].float().cpu().numpy(), rtol=2 * H, atol=0.03)
