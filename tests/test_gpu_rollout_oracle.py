"""The fused on-device rollout (csrc/rollout.hip, `pime_rollout`: the kernel bench.py times) against the CPU oracle.

One launch per episode does policy forward -> exploration noise -> residual composition -> env step (+ in-kernel
auto-reset) -> trajectory writes; here every piece of what it stored is re-derived on the host from the oracle:

  * exploration noise eps[t, lane]: bit-for-bit the oracle's Philox stream-2 Box-Muller draw (oracle_explore_noise);
  * policy mean: a_pre - eps*sigma against the oracle's double-accumulated forward of the same weights on the
    state the kernel saw, at EVERY step (not only t = 0);
  * env: the recorded pre-tanh actions replayed through OraclePH / OracleWT (fp64, reference semantics
    /root/reference/gym_control/envs/ph.py:320-348,409-445; nonlinear_watertank.py:800-826,890-939) via the
    reference's composition agent_residual.py:61, over TWO episodes so the in-kernel auto-reset and ensemble
    resampling are covered.

Tolerances (state_mode "mixed": f32 state words, f64 x/A/B/C): pH obs/reward 2e-5 rel and EVERY lane reads the oracle's
titration cell at every step (round 3: the residual tanh is the float64 tanh rounded once to float32 on the device and in
the oracle, so the env action is the same float64 number on both sides; a lane whose float64 tanh differs in the last bit
AND sits on a float32 rounding boundary may read the neighbouring cell: allowed on <= 1e-4 of the lanes per episode, observed
0); water tank 2e-4 (20 Euler sub-steps in f32)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _agent(algo, env, md):
    from pime_amd.utils import MODELS
    torch.manual_seed(0)
    ag = MODELS[algo.lower()](device=DEV)
    if "modular" in algo.lower():
        ag.init(md, env.state_dim, 1, 1)
    else:
        ag.init(md, env.state_dim, 1)
    if "residual" in algo.lower():
        ag.init_residual({"init_K": env.K.reshape(-1, 1)})
    with torch.no_grad():
        ag.act.net[-1].weight.normal_(0, 0.1)   # a non-trivial residual (zero-init would make the policy the prior alone)
        ag.act.net[-1].bias.normal_(0, 0.05)
    ag.weights_changed()
    return ag


def _oracle_mean(algo, obs, sd):
    import oracle
    if "modular" in algo.lower():
        return oracle.modular_actor_mean(obs, sd)[:, 0]
    return oracle.plain_actor_mean(obs, sd)[:, 0]


@pytest.mark.parametrize("env_name,algo,N,md", [
    ("PH_V35", "ResidualIntegratorModularPPO", 16384, 128),   # the bench configuration
    ("PH_V35", "ResidualPPO", 16384, 128),
    ("PH_V35", "PPO", 4096, 64),                              # plain PPO: the env sees tanh(a_pre), no prior term
    ("WT_INTEGRATOR", "ResidualIntegratorModularPPO", 4096, 128),   # BASELINE config 2 workload
    ("WT_INTEGRATOR", "ResidualPPO", 2048, 64),
    ("WT_STACKING10", "ResidualPPO", 2048, 128),                    # run_watertank_changing.sh:20-27 observation (30 floats)
    ("WT_STACKING4", "ResidualPPO", 1024, 64),
    ("WT_STACKING1", "PPO", 1024, 128),
])
def test_fused_rollout_replays_through_oracle(env_name, algo, N, md):
    import oracle
    from pime_amd import gym_control
    from pime_amd.elegantrl.run import make_buffer
    is_ph = env_name == "PH_V35"
    stack = int(env_name[len("WT_STACKING"):]) if env_name.startswith("WT_STACKING") else 0
    env_id = gym_control.WT_STACKING.format(stack) if stack else getattr(gym_control, env_name)
    seed, offset = 21, 4096     # a non-zero lane offset: the Philox counter word is the GLOBAL lane id
    kw = {} if is_ph else dict(reward_type="distance", max_step=60)
    env = gym_control.make_vec(env_id, N, device=DEV, state_mode="mixed", seed=seed, env_offset=offset, **kw)
    T = env.max_step
    ag = _agent(algo, env, md)
    assert ag._fused_rollout_ok(env), "the fused rollout path must serve this configuration"
    buf = make_buffer(ag, env, 2 * N * T)
    steps = ag.explore_env(env, buf, 2 * N * T, 1.0, 0.99)
    assert steps == 2 * N * T
    torch.cuda.synchronize()

    if is_ph:
        ref = oracle.OraclePH(N, oracle.ph_table(), seed=seed, env_offset=offset)
    else:
        ref = oracle.OracleWT(N, max_steps=T, reward_type="distance", num_stack=stack, seed=seed, env_offset=offset)
    priorK = ag._rollout_priorK()
    sd = {k: v.detach().cpu().numpy() for k, v in ag.act.state_dict().items()}
    sigma = np.float32(np.exp(sd["a_std_log"][0, 0]))
    state = buf.state[:2 * T + 1].cpu().numpy()
    action = buf.action[:2 * T, :, 0].cpu().numpy()
    noise = buf.noise[:2 * T, :, 0].cpu().numpy()
    reward = buf.reward[:2 * T].cpu().numpy()
    done = buf.done[:2 * T].cpu().numpy()

    obs = ref.reset()
    np.testing.assert_array_equal(state[0], obs)   # Philox reset draws + LUT: float32 obs bit-equal
    alive = np.ones(N, dtype=bool)
    cell_exact = []
    rtol = 2e-5 if is_ph else 2e-4
    for t in range(2 * T):
        ep, tt = divmod(t, T)
        # exploration noise: the kernel's Philox stream-2 Box-Muller draw, bit for bit
        want_eps = oracle.explore_noise(ag._rollout_seed, offset, N, ep + 1, tt)
        # (device ocml vs host libm log/cos/sqrt may differ in the last float64 bit, which survives the rounding to float32
        #  about once in 2^29 draws: allow a 1-ulp float32 difference on at most one draw in 10^4, count the rest as exact)
        neq = noise[t] != want_eps
        assert neq.mean() <= 1e-4, f"exploration noise differs on {neq.sum()} lanes at step {t}"
        np.testing.assert_allclose(noise[t], want_eps, rtol=1.2e-7, atol=0, err_msg=f"exploration noise, step {t}")
        # policy mean on the state the kernel saw (the kernel's own previous output), at every step
        mean = _oracle_mean(algo, state[t], sd)
        got_mean = action[t] - noise[t] * sigma
        np.testing.assert_allclose(got_mean, mean, rtol=3e-5, atol=3e-5, err_msg=f"policy mean, step {t}")
        # env step of the reference composition on the RECORDED action and the recorded observation
        act = oracle.residual_action(action[t], state[t], priorK)
        obs, _, rew, d = ref.step(act, auto_reset=True)
        assert bool(d.all()) == (tt == T - 1) and bool(d.any()) == bool(d.all())
        np.testing.assert_array_equal(done[t].astype(bool), d)
        if tt == T - 1:   # the obs row is the first observation of the next episode (in-kernel auto-reset, new ensemble draw)
            # y of the last step is not stored (the slot holds the next episode's first observation): checked through the reward
            ok = np.abs(reward[t] - rew) <= rtol * (1.0 + np.abs(rew))
            if is_ph:
                alive &= ok
            else:
                assert ok[alive].all()
            cell_exact.append(alive.mean())
            np.testing.assert_array_equal(state[t + 1], obs)
            alive[:] = True
            continue
        if is_ph:
            alive &= np.abs(state[t + 1][:, 0] - obs[:, 0]) <= 1e-5
        np.testing.assert_allclose(reward[t][alive], rew[alive], rtol=rtol, atol=rtol)
        np.testing.assert_allclose(state[t + 1][alive], obs[alive], rtol=rtol, atol=rtol)
        if not is_ph:   # the oracle continues from ITS state: re-sync it to the kernel's f32 state so errors do not compound
            D = state.shape[2]   # Stacking: the newest frame is the last one
            cols = (("h1", D - 3), ("h2", D - 2)) if stack else (("h1", 0), ("h2", 1), ("I", 3))
            for name, col in cols:
                ref.set(name, state[t + 1][:, col].astype(np.float64))
    assert min(cell_exact) >= 1.0 - 1e-4, f"only {min(cell_exact):.5f} of the lanes stayed cell-exact over an episode"
    # ensemble params were resampled by the in-kernel reset of episode 2 exactly as the oracle's
    if is_ph:
        np.testing.assert_allclose(env.get_field("qww_V"), ref.get("qww_V"), rtol=0, atol=0)
        np.testing.assert_allclose(env.get_field("A"), ref.get("A"), rtol=1e-15)
    else:
        np.testing.assert_allclose(env.get_field("a1"), ref.get("a1"), rtol=1e-7)
        np.testing.assert_allclose(env.get_field("Kp"), ref.get("Kp"), rtol=1e-7)
    env.close()
