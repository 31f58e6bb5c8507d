"""GPU parity tests proper: the HIP path (through the C ABI, via pime_amd.vec_env) against the CPU oracle and
the reference-generated golden vectors.  Run on the GPU box: pytest -m gpu.

Tolerances
  state_mode f64   : x (reaction invariant), obs and reward agree with the fp64 oracle to 1e-12 relative
                     (only exp/expm1/sqrt/log differ between device and host libm); the LUT index is identical.
  state_mode mixed : float32 state words => 2e-5 relative on obs / reward per step; pH LUT index identical
                     because x, A, B, C stay float64.
"""
import numpy as np
import pytest
import torch

import oracle
from conftest import load_golden

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def _t(a, dtype=torch.float64):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=dtype, device=DEV)


def _np(t):
    return t.detach().cpu().numpy()


@pytest.fixture(scope="module")
def table():
    return oracle.ph_table()


# ------------------------------------------------------------------------------------------------ golden replays
@pytest.mark.parametrize("mode", ["f64", "mixed"])
@pytest.mark.parametrize("tag", ["v35", "noib", "dist", "sparse", "punish"])
def test_ph_golden_rollouts(tag, mode):
    """All 16 seeds x 2 policies of the reference-generated pH rollouts as ONE 32-lane launch sequence."""
    from pime_amd.vec_env import VecPH, CallbackDraws
    g = load_golden("ph_rollouts.npz")
    pre = [f"{tag}_{pol}_" for pol in ("prior", "resid")]
    params = np.concatenate([g[p + "params"] for p in pre])
    x0 = np.concatenate([g[p + "x0"] for p in pre])
    r = np.concatenate([g[p + "r"] for p in pre])
    act = np.concatenate([g[p + "act"] for p in pre])      # [32, 50]
    want_obs = np.concatenate([g[p + "obs"] for p in pre])
    want_rew = np.concatenate([g[p + "rew"] for p in pre])
    want_x = np.concatenate([g[p + "x"] for p in pre])
    obs0 = np.concatenate([g[p + "obs0"] for p in pre])
    N = len(x0)
    kw = dict(v35={}, noib=dict(integral_bound=False), dist=dict(reward_type="distance"),
              sparse=dict(reward_type="sparse"),
              punish=dict(action_punishment=0.1, action_change_punishment=0.2, integral_punish=0.05))[tag]
    draws = CallbackDraws(lambda i: (params[i, 0], params[i, 1], x0[i], r[i]))
    env = VecPH(N, device=DEV, state_mode=mode, draws=draws, **kw)
    obs = env.reset()
    np.testing.assert_array_equal(_np(obs), obs0.astype(np.float32))
    tol = dict(rtol=1e-12, atol=1e-12) if mode == "f64" else dict(rtol=2e-5, atol=2e-5)
    for t in range(50):
        obs, rew, done = env.step(_t(act[:, t]), auto_reset=False)
        np.testing.assert_allclose(env.get_field("x"), want_x[:, t], rtol=1e-12, atol=1e-12)
        if mode == "f64":
            np.testing.assert_array_equal(_np(obs), want_obs[:, t].astype(np.float32))
            np.testing.assert_allclose(env.get_field("y"), want_obs[:, t, 0], rtol=0, atol=4e-15)
            np.testing.assert_allclose(env.get_field("I"), want_obs[:, t, 2], **tol)
        else:
            np.testing.assert_allclose(_np(obs), want_obs[:, t], rtol=2e-5, atol=2e-5)
        np.testing.assert_allclose(_np(rew), want_rew[:, t], rtol=2e-5 if mode == "mixed" else 1e-6, atol=2e-5)
        assert bool(_np(done).all()) == (t == 49)
    env.close()


@pytest.mark.parametrize("mode", ["f64", "mixed"])
@pytest.mark.parametrize("tag", ["dist", "sq", "sparse", "zero"])
def test_wt_golden_rollouts(tag, mode):
    from pime_amd.vec_env import VecWaterTank, CallbackDraws
    g = load_golden("wt_rollouts.npz")
    pre = [f"{tag}_{pol}_" for pol in ("prior", "resid")]
    cat = lambda k: np.concatenate([g[p + k] for p in pre])  # noqa: E731
    params, obs0, act, noise = cat("params"), cat("obs0"), cat("act"), cat("noise")
    want_obs, want_rew = cat("obs"), cat("rew")
    N = len(params)
    kw = dict(dist=dict(reward_type="distance"), sq=dict(reward_type="square_distance"),
              sparse=dict(reward_type="sparse"), zero=dict(reward_type="distance", noise_scale=0.0))[tag]
    step = {"t": 0}
    draws = CallbackDraws(lambda i: (*params[i], *obs0[i, :3]), lambda i: noise[i, step["t"]])
    env = VecWaterTank(N, device=DEV, state_mode=mode, draws=draws, **kw)
    obs = env.reset()
    np.testing.assert_array_equal(_np(obs), obs0.astype(np.float32))
    tol = dict(rtol=1e-12, atol=1e-12) if mode == "f64" else dict(rtol=3e-5, atol=3e-5)
    for t in range(200):
        step["t"] = t
        obs, rew, done = env.step(_t(act[:, t]), auto_reset=False)
        if mode == "f64":
            np.testing.assert_allclose(env.get_field("h1"), want_obs[:, t, 0], **tol)
            np.testing.assert_allclose(env.get_field("h2"), want_obs[:, t, 1], **tol)
            np.testing.assert_allclose(env.get_field("I"), want_obs[:, t, 3], **tol)
            np.testing.assert_allclose(_np(obs), want_obs[:, t].astype(np.float32), rtol=1.2e-7, atol=0)
        else:
            np.testing.assert_allclose(_np(obs), want_obs[:, t], rtol=1e-4, atol=1e-4)
        np.testing.assert_allclose(_np(rew), want_rew[:, t], rtol=1e-4 if mode == "mixed" else 1e-6, atol=1e-4 if mode == "mixed" else 1e-6)
        assert bool(_np(done).all()) == (t == 199)
    env.close()


@pytest.mark.parametrize("S", [1, 4, 10])
def test_wt_stacking_golden(S):
    from pime_amd.vec_env import VecWaterTank, CallbackDraws
    g = load_golden("wt_stacking.npz")
    p = f"s{S}_"
    E = g[p + "params"].shape[0]
    step = {"t": 0}
    draws = CallbackDraws(lambda i: (*g[p + "params"][i], *g[p + "obs0"][i][-3:]), lambda i: g[p + "noise"][i, step["t"]])
    env = VecWaterTank(E, device=DEV, state_mode="f64", draws=draws, reward_type="distance", num_stack=S)
    assert env.obs_dim == 3 * S
    np.testing.assert_array_equal(env.K, g[p + "K"])
    obs = env.reset()
    np.testing.assert_array_equal(_np(obs), g[p + "obs0"].astype(np.float32))
    for t in range(24):
        step["t"] = t
        obs, rew, _ = env.step(_t(g[p + "act"][:, t]), auto_reset=False)
        np.testing.assert_allclose(_np(obs), g[p + "obs"][:, t].astype(np.float32), rtol=1.2e-7, atol=0)
        np.testing.assert_allclose(_np(rew), g[p + "rew"][:, t], rtol=1e-6, atol=1e-6)
    env.close()


def test_ph_chain_autoreset_golden():
    """Three chained reference episodes through the in-kernel auto-reset with injected MT19937 draws."""
    from pime_amd.vec_env import VecPH, CallbackDraws
    g = load_golden("ph_rollouts.npz")
    ep = {"k": 0}
    draws = CallbackDraws(lambda i: (*g["chain_params"][ep["k"]], g["chain_x0"][ep["k"]], g["chain_r"][ep["k"]]))
    env = VecPH(1, device=DEV, state_mode="f64", draws=draws)
    obs = env.reset()
    for k in range(3):
        np.testing.assert_array_equal(_np(obs)[0], g["chain_obs0"][k].astype(np.float32))
        for t in range(50):
            ep["k"] = min(k + 1, 2)
            obs, rew, done = env.step(_t(g["chain_act"][k, t:t + 1]), auto_reset=True)
            np.testing.assert_allclose(_np(rew)[0], g["chain_rew"][k, t], rtol=1e-6, atol=1e-6)
            assert bool(_np(done)[0]) == (t == 49)
            if t < 49:
                np.testing.assert_array_equal(_np(obs)[0], g["chain_obs"][k, t].astype(np.float32))
    env.close()


# ------------------------------------------------------------------------------------------------ oracle, seeded inputs
@pytest.mark.parametrize("mode", ["f64", "mixed"])
@pytest.mark.parametrize("n_resample", [1, 3, 0])
def test_ph_philox_vs_oracle(table, mode, n_resample):
    """In-kernel Philox resets + auto-reset + resample-every-n over 3 episodes, 4 096 lanes, random actions:
    identical draws (bitwise), identical LUT index, x to 1e-12."""
    from pime_amd.vec_env import VecPH
    N, seed, off = 4096, 1234567, 77
    env = VecPH(N, device=DEV, state_mode=mode, seed=seed, env_offset=off, resample_every=max(n_resample, 1))
    ref = oracle.OraclePH(N, table, resample_every=max(n_resample, 1), seed=seed, env_offset=off)
    obs = env.reset()
    want = ref.reset()
    if n_resample == 0:
        env.set_reset_all(False)
        ref2 = oracle.OraclePH(N, table, resample_every=0, seed=seed, env_offset=off)
        for f in ("qww_V", "qc_V", "x", "r"):
            ref2.set(f, ref.get(f))
        ref2.set("episode", ref.get("episode"))
        ref = ref2
    np.testing.assert_array_equal(_np(obs), want)
    for f in ("qww_V", "qc_V", "x", "r"):
        np.testing.assert_allclose(env.get_field(f), ref.get(f), rtol=1e-7 if (mode == "mixed" and f == "r") else 0, atol=0)
    np.testing.assert_allclose(env.get_field("A"), ref.get("A"), rtol=4e-16)
    np.testing.assert_allclose(env.get_field("B"), ref.get("B"), rtol=2e-15)
    rng = np.random.RandomState(5)
    for t in range(150):
        a = rng.uniform(-1.3, 1.3, N)  # exercises the clip
        obs, rew, done = env.step(_t(a), auto_reset=True)
        w_obs, w_obs64, w_rew, w_done = ref.step(a, auto_reset=True)
        np.testing.assert_array_equal(_np(done).astype(bool), w_done)
        np.testing.assert_allclose(env.get_field("x"), ref.get("x"), rtol=1e-12, atol=1e-12)
        if mode == "f64":
            np.testing.assert_array_equal(_np(obs)[:, 0], w_obs[:, 0])  # same LUT cell on every lane
            np.testing.assert_allclose(_np(obs), w_obs, rtol=1.2e-7, atol=0)
            np.testing.assert_allclose(_np(rew), w_rew, rtol=1e-6, atol=1e-6)
        else:
            np.testing.assert_allclose(_np(obs), w_obs, rtol=3e-5, atol=3e-5)
            np.testing.assert_allclose(_np(rew), w_rew, rtol=3e-5, atol=3e-5)
    np.testing.assert_array_equal(env.get_field("episode"), ref.get("episode"))
    np.testing.assert_array_equal(env.get_field("qww_V"), ref.get("qww_V"))
    env.close()


@pytest.mark.parametrize("mode", ["f64", "mixed"])
@pytest.mark.parametrize("num_stack", [0, 4])
def test_wt_philox_vs_oracle(mode, num_stack):
    from pime_amd.vec_env import VecWaterTank
    N, seed, off = 2048, 99, 5
    env = VecWaterTank(N, device=DEV, state_mode=mode, seed=seed, env_offset=off, reward_type="distance",
                       num_stack=num_stack, max_step=40)
    ref = oracle.OracleWT(N, max_steps=40, reward_type="distance", num_stack=num_stack, seed=seed, env_offset=off)
    obs = env.reset()
    np.testing.assert_allclose(_np(obs), ref.reset(), rtol=0, atol=0)
    rng = np.random.RandomState(6)
    tol = dict(rtol=1e-10, atol=1e-11) if mode == "f64" else dict(rtol=2e-4, atol=2e-4)
    for t in range(90):
        a = rng.uniform(-1.5, 1.5, N)  # the water tank does not clip the action
        obs, rew, done = env.step(_t(a), auto_reset=True)
        w_obs, w_obs64, w_rew, w_done = ref.step(a, auto_reset=True)
        np.testing.assert_array_equal(_np(done).astype(bool), w_done)
        np.testing.assert_allclose(env.get_field("h1"), ref.get("h1"), **tol)
        np.testing.assert_allclose(env.get_field("h2"), ref.get("h2"), **tol)
        np.testing.assert_allclose(_np(obs), w_obs, rtol=2e-7 if mode == "f64" else 2e-4, atol=0 if mode == "f64" else 2e-4)
        np.testing.assert_allclose(_np(rew), w_rew, rtol=1e-6 if mode == "f64" else 2e-4, atol=1e-6 if mode == "f64" else 2e-4)
    env.close()


def test_residual_step_matches_composed_action(table):
    """pime_env_step_residual == step(tanh(a_pre) + obs @ priorK) (agent_residual.py:61).

    np.tanh of the float32 action is the float64 tanh rounded once to float32 on the device and in the oracle (round 3), so the
    two env actions are the same float64 number and every pH lane reads the same titration cell: no lane may be off."""
    from pime_amd.vec_env import VecPH, VecWaterTank
    for Env, kw in ((VecPH, {}), (VecWaterTank, dict(reward_type="distance")), (VecWaterTank, dict(num_stack=4))):
        N = 1024
        e1 = Env(N, device=DEV, state_mode="f64", seed=3, **kw)
        e2 = Env(N, device=DEV, state_mode="f64", seed=3, **kw)
        o1, o2 = e1.reset().clone(), e2.reset().clone()
        g = torch.Generator(device="cpu").manual_seed(0)
        off_cell = 0
        for t in range(20):
            a_pre = (torch.randn(N, generator=g) * 0.6).to(DEV)
            act = oracle.residual_action(_np(a_pre), _np(o2), -e2.K)
            n1, r1, d1 = e1.step_residual(a_pre, o1)
            n2, r2, d2 = e2.step(_t(act))
            if Env is VecPH:
                np.testing.assert_allclose(e1.get_field("x"), e2.get_field("x"), rtol=1e-6, atol=1e-9)
                off_cell += int((np.abs(_np(n1)[:, 0] - _np(n2)[:, 0]) > 1e-6).sum())
                np.testing.assert_allclose(_np(n1), _np(n2), rtol=2e-6, atol=2e-6)
                np.testing.assert_allclose(_np(r1), _np(r2), rtol=2e-5, atol=2e-5)
                o1, o2 = n1.clone(), n2.clone()
            else:
                np.testing.assert_allclose(_np(n1), _np(n2), rtol=1e-5, atol=1e-5)
                np.testing.assert_allclose(_np(r1), _np(r2), rtol=1e-4, atol=1e-4)
                o1, o2 = n1.clone(), n2.clone()
        assert off_cell == 0
        e1.close(); e2.close()


def test_field_io_and_set_params(table):
    from pime_amd.vec_env import VecPH
    env = VecPH(8, device=DEV, state_mode="f64", seed=1)
    env.reset()
    env.set_params(0.01, 0.002)
    a, b, c = oracle.ph_zoh(0.01, 0.002)
    np.testing.assert_allclose(env.get_field("A"), a, rtol=1e-15)
    np.testing.assert_allclose(env.get_field("B"), b, rtol=1e-15)
    np.testing.assert_array_equal(env.get_field("C"), c)
    mask = np.array([1, 0, 0, 0, 0, 0, 0, 1], dtype=np.uint8)
    env.set_field("x", 12.5, mask)
    x = env.get_field("x")
    assert x[0] == 12.5 and x[7] == 12.5 and x[1] != 12.5
    k = int(np.rint(0.002 * 12.5 * 1e5))
    assert env.get_field("y")[0] == table[k]
    obs = _np(env.observe())
    assert obs[0, 0] == np.float32(table[k])
    env.close()


# ------------------------------------------------------------------------------------------------ full-size properties
def test_ph_full_size_properties():
    """BASELINE config 3 shape (16 384 lanes x 50 steps) in the bench's mixed mode: size-independent invariants."""
    from pime_amd.vec_env import VecPH
    N = 16384
    env = VecPH(N, device=DEV, state_mode="mixed", seed=11)
    obs0 = env.reset().clone()
    assert torch.all((obs0[:, 1] >= 3) & (obs0[:, 1] <= 11)) and torch.all(obs0[:, 2] == 0)
    qww0, qc0 = env.get_changable_parameters()
    assert qww0.min() >= 0.005 and qww0.max() <= 0.015 and qc0.min() >= 0.0015 and qc0.max() <= 0.0025
    assert abs(qww0.mean() - 0.01) < 2e-4 and abs(qc0.mean() - 0.002) < 2e-5  # U(lo,hi) means over 16 384 lanes
    g = torch.Generator(device=DEV).manual_seed(0)
    obs = obs0
    ret = torch.zeros(N, device=DEV)
    for t in range(50):
        a_pre = torch.randn(N, device=DEV, generator=g) * 0.6065
        nxt, rew, done = env.step_residual(a_pre, obs)
        ret += rew
        assert torch.all(rew <= 0)
        assert bool(done.all()) == (t == 49) and bool(done.any()) == (t == 49)
        if t < 49:
            assert torch.all(nxt[:, 1] == obs[:, 1])              # the goal is constant inside an episode
            assert torch.all(nxt[:, 2].abs() <= 25.0)             # integrator bound (ph.py:341)
            assert torch.all((nxt[:, 0] > 0) & (nxt[:, 0] < 11.8))  # inside the titration curve
        obs = nxt.clone()
    # after the auto-reset: new goals, zero integrators, new ensemble params on every lane (resample_every = 1)
    assert torch.all(obs[:, 2] == 0) and not torch.equal(obs[:, 1], obs0[:, 1])
    qww1, _ = env.get_changable_parameters()
    assert np.mean(qww1 != qww0) > 0.999
    assert np.all(env.get_field("episode") == 1) and np.all(env.get_field("t") == 0)
    env.close()


@pytest.mark.parametrize("N", [1, 63, 65, 257])
def test_ragged_lane_counts_and_masked_reset(table, N):
    """Lane counts that are not multiples of the wave size, and a partial (masked) reset: untouched lanes keep their
    state, reset lanes start a new episode with new Philox draws (episode counter advanced only for them)."""
    from pime_amd.vec_env import VecPH, VecWaterTank
    env = VecPH(N, device=DEV, state_mode="f64", seed=21)
    ref = oracle.OraclePH(N, table, seed=21)
    np.testing.assert_array_equal(_np(env.reset()), ref.reset())
    rng = np.random.RandomState(N)
    for t in range(7):
        a = rng.uniform(-1, 1, N)
        obs, rew, done = env.step(_t(a), auto_reset=False)
        w_obs, _, w_rew, _ = ref.step(a)
        np.testing.assert_array_equal(_np(obs), w_obs)
    mask = (rng.uniform(size=N) < 0.5).astype(np.uint8)
    if N == 1:
        mask[:] = 1
    before = _np(env.observe()).copy()
    obs = _np(env.reset(mask=mask))
    want = ref.reset(mask=mask)
    sel = mask.astype(bool)
    np.testing.assert_array_equal(obs[sel], want[sel])
    np.testing.assert_array_equal(_np(env.observe())[~sel], before[~sel])
    np.testing.assert_array_equal(env.get_field("episode"), ref.get("episode"))
    np.testing.assert_array_equal(env.get_field("t")[sel], 0)
    assert np.all(env.get_field("t")[~sel] == 7)
    env.close()
    wt = VecWaterTank(N, device=DEV, state_mode="f64", seed=4, reward_type="distance")
    rw = oracle.OracleWT(N, reward_type="distance", seed=4)
    np.testing.assert_array_equal(_np(wt.reset()), rw.reset())
    a = rng.uniform(-1, 1, N)
    obs, rew, done = wt.step(_t(a), auto_reset=False)
    w_obs, _, w_rew, _ = rw.step(a)
    np.testing.assert_allclose(_np(obs), w_obs, rtol=2e-7)
    wt.close()


def test_mixed_ph_and_tank_batch_wide_ensemble(table):
    """BASELINE.json config 5 as a parity case (cfg 5): a pH batch and a water-tank batch advanced
    side by side on two HIP streams, ensemble ranges 1.5x wider than the registered ones (domain-randomised sweep), in-kernel
    Philox resets and per-episode resampling, against the oracle with the same ranges.  State storage is float32
    (PIME_STATE_MIXED); the binary16 storage of that config is tests/test_gpu_env_fp16.py and tests/test_gpu_config5.py.  The pH range keeps
    C*x inside the 100 000-entry titration table (the reference raises IndexError beyond it)."""
    from pime_amd.vec_env import VecPH, VecWaterTank
    N, seed = 2048, 4242
    qww, qc = (0.0045, 0.0165), (0.00125, 0.00275)
    a1 = a2 = (0.0012, 0.0027)
    kp = (0.045, 0.195)
    s_ph, s_wt = torch.cuda.Stream(device=DEV), torch.cuda.Stream(device=DEV)
    with torch.cuda.stream(s_ph):
        ph = VecPH(N, device=DEV, state_mode="mixed", seed=seed, qww_V=qww, qc_V=qc)
        o_ph = ph.reset()
    with torch.cuda.stream(s_wt):
        wt = VecWaterTank(N, device=DEV, state_mode="mixed", seed=seed + 1, reward_type="distance", max_step=30, a1=a1, a2=a2,
                          Kp=kp)
        o_wt = wt.reset()
    r_ph = oracle.OraclePH(N, table, seed=seed)
    r_ph.set_ranges(qww, qc)
    r_wt = oracle.OracleWT(N, max_steps=30, reward_type="distance", seed=seed + 1)
    r_wt.set_ranges(a1, a2, kp)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(_np(o_ph), r_ph.reset())
    np.testing.assert_allclose(_np(o_wt), r_wt.reset(), rtol=0, atol=0)
    for f, rng_ in (("qww_V", qww), ("qc_V", qc)):
        v = ph.get_field(f)
        np.testing.assert_array_equal(v, r_ph.get(f))
        assert v.min() >= rng_[0] and v.max() <= rng_[1] and v.max() - v.min() > 0.9 * (rng_[1] - rng_[0])
    rng = np.random.RandomState(11)
    for t in range(110):   # > 2 pH episodes, > 3 tank episodes: resampled plants on every lane
        a_ph, a_wt = rng.uniform(-1.2, 1.2, N), rng.uniform(-1.5, 1.5, N)
        with torch.cuda.stream(s_ph):
            g_ph = ph.step(_t(a_ph), auto_reset=True)
        with torch.cuda.stream(s_wt):
            g_wt = wt.step(_t(a_wt), auto_reset=True)
        torch.cuda.synchronize()
        w_obs, _, w_rew, w_done = r_ph.step(a_ph, auto_reset=True)
        np.testing.assert_array_equal(_np(g_ph[2]).astype(bool), w_done)
        np.testing.assert_allclose(ph.get_field("x"), r_ph.get("x"), rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(_np(g_ph[0]), w_obs, rtol=3e-5, atol=3e-5)
        np.testing.assert_allclose(_np(g_ph[1]), w_rew, rtol=3e-5, atol=3e-5)
        w_obs, _, w_rew, w_done = r_wt.step(a_wt, auto_reset=True)
        np.testing.assert_array_equal(_np(g_wt[2]).astype(bool), w_done)
        np.testing.assert_allclose(_np(g_wt[0]), w_obs, rtol=2e-4, atol=2e-4)
        np.testing.assert_allclose(_np(g_wt[1]), w_rew, rtol=2e-4, atol=2e-4)
    np.testing.assert_array_equal(ph.get_field("qww_V"), r_ph.get("qww_V"))
    np.testing.assert_allclose(wt.get_field("Kp"), r_wt.get("Kp"), rtol=1e-7)
    ph.close(); wt.close()
