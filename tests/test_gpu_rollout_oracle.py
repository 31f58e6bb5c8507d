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

from rollout_replay import DEV, make_agent, replay_through_oracle

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("tiles", ["wide", "narrow", "quad"])   # 32- / 16-lane tiles per wave, one 16-lane tile per workgroup (csrc/rollout.hip)
@pytest.mark.parametrize("env_name,algo,N,md", [
    ("PH_V35", "ResidualIntegratorModularPPO", 16384, 128),   # the bench configuration
    ("PH_V35", "ResidualPPO", 16384, 128),
    ("PH_V35", "PPO", 4096, 64),                              # plain PPO: the env sees tanh(a_pre), no prior term
    ("WT_INTEGRATOR", "ResidualIntegratorModularPPO", 4096, 128),   # BASELINE config 2 workload
    ("WT_INTEGRATOR", "ResidualPPO", 2048, 64),
    ("WT_STACKING10", "ResidualPPO", 2048, 128),                    # run_watertank_changing.sh:20-27 observation (30 floats)
    ("WT_STACKING4", "ResidualPPO", 1024, 64),
    ("WT_STACKING1", "PPO", 1024, 128),
    ("PH_V35", "ResidualIntegratorModularPPO", 2048, 64),          # the modular actor's 64 -> 32 towers (one 32-feature tile each)
    ("WT_INTEGRATOR", "ResidualIntegratorModularPPO", 1000, 64),   # ... and a ragged lane count
    # width 256 (round 3): the streamed 16-tile rollout kernel (csrc/mlp16.hip: rollout16_kernel)
    ("WT_STACKING10", "ResidualPPO", 2048, 256),                    # run_watertank_changing.sh:20-27 as it is run
    ("WT_INTEGRATOR", "ResidualIntegratorModularPPO", 1024, 256),   # run_watertank_changing.sh:11-18
    ("PH_V35", "ResidualIntegratorModularPPO", 1000, 256),          # ragged lane count: 15 full 64-lane workgroups + 40 lanes
    ("PH_V35", "ResidualPPO", 512, 256),
    ("WT_STACKING4", "PPO", 256, 256),
])
def test_fused_rollout_replays_through_oracle(env_name, algo, N, md, tiles, monkeypatch):
    import oracle
    if md == 256:   # the streamed 16-tile family: one tile per wave, or (plain actor) one tile per workgroup
        if tiles == "wide":
            pytest.skip("width 256 has no 32-lane tiling")
        monkeypatch.setenv("PIME_ROLLOUT_NARROW", {"narrow": "1", "quad": "2"}[tiles])
    else:   # the library picks 16-lane tiles (one per workgroup up to 4 096 lanes) by itself; every tiling must replay
        monkeypatch.setenv("PIME_ROLLOUT_NARROW", {"wide": "0", "narrow": "1", "quad": "2"}[tiles])
    from pime_amd import gym_control
    from pime_amd.elegantrl.run import make_buffer
    is_ph = env_name == "PH_V35"
    stack = int(env_name[len("WT_STACKING"):]) if env_name.startswith("WT_STACKING") else 0
    env_id = gym_control.WT_STACKING.format(stack) if stack else getattr(gym_control, env_name)
    seed, offset = 21, 4096     # a non-zero lane offset: the Philox counter word is the GLOBAL lane id
    kw = {} if is_ph else dict(reward_type="distance", max_step=60)
    env = gym_control.make_vec(env_id, N, device=DEV, state_mode="mixed", seed=seed, env_offset=offset, **kw)
    T = env.max_step
    ag = make_agent(algo, env, md)
    assert ag._fused_rollout_ok(env), "the fused rollout path must serve this configuration"
    buf = make_buffer(ag, env, 2 * N * T)
    steps = ag.explore_env(env, buf, 2 * N * T, 1.0, 0.99)
    assert steps == 2 * N * T
    torch.cuda.synchronize()
    if is_ph:
        ref = oracle.OraclePH(N, oracle.ph_table(), seed=seed, env_offset=offset)
    else:
        ref = oracle.OracleWT(N, max_steps=T, reward_type="distance", num_stack=stack, seed=seed, env_offset=offset)
    replay_through_oracle(ag, algo, env, buf, ref, 2, offset, is_ph, stack)
    # ensemble params were resampled by the in-kernel reset of episode 2 exactly as the oracle's
    if is_ph:
        np.testing.assert_allclose(env.get_field("qww_V"), ref.get("qww_V"), rtol=0, atol=0)
        np.testing.assert_allclose(env.get_field("A"), ref.get("A"), rtol=1e-15)
    else:
        np.testing.assert_allclose(env.get_field("a1"), ref.get("a1"), rtol=1e-7)
        np.testing.assert_allclose(env.get_field("Kp"), ref.get("Kp"), rtol=1e-7)
    env.close()


@pytest.mark.parametrize("env_name,algo,N,md", [("PH_V35", "ResidualIntegratorModularPPO", 3000, 128),
                                               ("WT_INTEGRATOR", "ResidualIntegratorModularPPO", 1024, 64),
                                               ("WT_STACKING4", "ResidualPPO", 2048, 128),
                                               ("WT_STACKING10", "ResidualPPO", 1000, 256)])   # the streamed family's layer16q
def test_quad_and_narrow_tilings_give_the_same_bits(env_name, algo, N, md, monkeypatch):
    """One 16-lane tile per workgroup (QUAD: each wave a quarter of a layer's output tiles) runs the same MFMA sequence per
    accumulator as one tile per wave (NARROW): actions, observations, rewards and done flags are bit-identical -- which is what keeps a
    lane's trajectory independent of the launch size (<= 4 096 lanes take QUAD, larger launches NARROW)."""
    from pime_amd import gym_control
    from pime_amd.elegantrl.run import make_buffer
    stack = int(env_name[len("WT_STACKING"):]) if env_name.startswith("WT_STACKING") else 0
    env_id = gym_control.WT_STACKING.format(stack) if stack else getattr(gym_control, env_name)
    kw = {} if env_name == "PH_V35" else dict(reward_type="distance", max_step=40)
    bufs = []
    for mode in ("1", "2"):
        monkeypatch.setenv("PIME_ROLLOUT_NARROW", mode)
        env = gym_control.make_vec(env_id, N, device=DEV, state_mode="mixed", seed=9, env_offset=77, **kw)
        ag = make_agent(algo, env, md)
        buf = make_buffer(ag, env, 2 * N * env.max_step)
        ag.explore_env(env, buf, 2 * N * env.max_step, 1.0, 0.99)
        torch.cuda.synchronize()
        bufs.append(buf)
        env.close()
    a, b = bufs
    T = a.length
    assert T == b.length and T > 0
    for name in ("action", "noise", "reward", "done"):
        assert torch.equal(getattr(a, name)[:T], getattr(b, name)[:T]), name
    assert torch.equal(a.state[:T + 1], b.state[:T + 1])
