"""Pins the CPU oracle (oracle/pime_oracle.c) against vectors produced by the unmodified reference
(tests/golden/make_golden.py).  fp64 restatement => tolerance 1e-12 absolute unless stated."""
import numpy as np
import pytest

import oracle
from conftest import load_golden

TOL = 1e-12


def test_philox_known_answers():
    # Random123 kat_vectors, philox4x32-10
    kat = [
        ((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
        ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
        ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
         (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
    ]
    for ctr, key, want in kat:
        assert tuple(int(v) for v in oracle.philox4x32_10(ctr, key)) == want


def test_ph_table(ph_table_oracle):
    g = load_golden("ph_table.npz")
    got = ph_table_oracle
    assert got.shape == (100000,)
    # The reference takes -log10 through numpy's ufunc, which on AVX-512 hosts is Intel SVML and differs from
    # glibc's log10 by 1 ulp on ~8 % of the entries (measured in this container); the Newton iterates use
    # scalar ** -> libm pow and agree.  So the table is reproducible to 1 ulp, not bitwise, ACROSS HOSTS of the
    # reference itself; the oracle is held to 2 ulp on every sampled entry (max |pH| = 11.7 -> ulp 1.8e-15).
    np.testing.assert_allclose(got[g["k"]], g["pH"], rtol=4.5e-16, atol=4e-18)
    assert np.mean(got[g["k"]] == g["pH"]) > 0.9
    # SURVEY.md 8(a1) anchors
    np.testing.assert_allclose([got[0], got[1000], got[99999]],
                               [11.702028825841987, 9.24882913017368, 0.008778344598710155], rtol=4.5e-16)
    assert np.all(np.diff(got) < 0) and np.abs(np.diff(got)).max() < 0.02963  # monotone; steepest cell (k=1500)


def test_ph_zoh():
    t = load_golden("ph_zoh.npz")["table"]
    for qww, qc, A, B, C in t:
        a, b, c = oracle.ph_zoh(qww, qc)
        assert abs(a - A) <= 4e-16 * A and abs(b - B) <= 2e-15 * B and c == C


def _ph_env(table, tag):
    kw = dict(v35={}, noib=dict(integral_bound=False), dist=dict(reward_type="distance"),
              sparse=dict(reward_type="sparse"), punish={})[tag]
    e = oracle.OraclePH(1, table, **kw)
    if tag == "punish":
        e.set_punish(integral=0.05, action=0.1, action_change=0.2)
    return e


@pytest.mark.parametrize("tag", ["v35", "noib", "dist", "sparse", "punish"])
@pytest.mark.parametrize("pol", ["prior", "resid"])
def test_ph_rollouts(ph_table_oracle, tag, pol):
    g = load_golden("ph_rollouts.npz")
    p = f"{tag}_{pol}_"
    S = g[p + "x0"].shape[0]
    for s in range(S):
        e = _ph_env(ph_table_oracle, tag)
        draws = np.array([[*g[p + "params"][s], g[p + "x0"][s], g[p + "r"][s]]])
        obs0 = e.reset(draws=draws)
        np.testing.assert_allclose(obs0[0], g[p + "obs0"][s].astype(np.float32), rtol=0, atol=0)
        for t in range(50):
            obs, obs64, rew, done = e.step(g[p + "act"][s, t])
            np.testing.assert_allclose(e.get("x")[0], g[p + "x"][s, t], rtol=1e-13, atol=TOL)
            np.testing.assert_allclose(obs64[0], g[p + "obs"][s, t], rtol=0, atol=1e-11)
            np.testing.assert_allclose(rew[0], g[p + "rew"][s, t], rtol=1e-13, atol=1e-11)
            assert bool(done[0]) == bool(g[p + "done"][s, t])
        # the residual action composition (agent_residual.py:61) reproduces the recorded env actions
        if pol == "resid":
            K = np.array([-0.02, 0.02, 0.035])
            obs_seen = np.concatenate([g[p + "obs0"][s][None], g[p + "obs"][s, :-1]]).astype(np.float32)
            act = oracle.residual_action(g[p + "a_pre"][s], obs_seen, -K)
            np.testing.assert_allclose(act, g[p + "act"][s], rtol=0, atol=3e-7)  # tanhf vs numpy f32 tanh: 1-2 ulp


def test_ph_chain_autoreset(ph_table_oracle):
    """Three back-to-back episodes with auto-reset: same trajectory as reset-per-episode in the reference."""
    g = load_golden("ph_rollouts.npz")
    e = oracle.OraclePH(1, ph_table_oracle)
    E = g["chain_x0"].shape[0]
    draws = [np.array([[*g["chain_params"][k], g["chain_x0"][k], g["chain_r"][k]]]) for k in range(E)]
    obs = e.reset(draws=draws[0])
    for k in range(E):
        np.testing.assert_array_equal(obs[0], g["chain_obs0"][k].astype(np.float32))
        for t in range(50):
            nxt = draws[k + 1] if k + 1 < E else draws[0]
            obs, obs64, rew, done = e.step(g["chain_act"][k, t], auto_reset=True, reset_draws=nxt)
            np.testing.assert_allclose(rew[0], g["chain_rew"][k, t], rtol=1e-13, atol=1e-11)
            assert bool(done[0]) == (t == 49)
            if t < 49:
                np.testing.assert_allclose(obs64[0], g["chain_obs"][k, t], rtol=0, atol=1e-11)


def test_ph_keep_params(ph_table_oracle):
    """set_reset_all(False) == resample_every 0: plant params survive reset (ph.py:428-445)."""
    g = load_golden("ph_rollouts.npz")
    e = oracle.OraclePH(1, ph_table_oracle, resample_every=0)
    e.set("qww_V", g["keep_params"][0][0])
    e.set("qc_V", g["keep_params"][0][1])
    assert np.all(g["keep_params"][0] == g["keep_params"][1])
    for k in range(2):
        bogus = np.array([[0.123, 0.456, g["keep_x0"][k], g["keep_r"][k]]])  # params in the draw row are ignored
        e.reset(draws=bogus)
        for t in range(50):
            _, obs64, rew, _ = e.step(g["keep_act"][k, t])
            np.testing.assert_allclose(obs64[0], g["keep_obs"][k, t], rtol=0, atol=1e-11)


@pytest.mark.parametrize("tag", ["nominal", "corner"])
def test_ph_stepresponse(ph_table_oracle, tag):
    """utils/test.py:1369-1407 protocol with the prior PI controller."""
    g = load_golden("ph_stepresponse.npz")
    K = np.array([-0.02, 0.02, 0.035])
    e = oracle.OraclePH(1, ph_table_oracle, resample_every=0)
    e.set("qww_V", g[tag + "_params"][0])
    e.set("qc_V", g[tag + "_params"][1])
    x = 0.0
    i = 0
    for r in [10., 6, 3, 8, 5]:
        e.reset(draws=np.array([[0, 0, 1.0, 5.0]]))
        e.set("x", x)
        e.set("r", r)
        obs = np.array([e.get("y")[0], r, 0.0], dtype=np.float32)
        for n in range(50):
            a = float(obs @ (-K))
            assert abs(a - g[tag + "_act"][i]) < 1e-12
            np.testing.assert_allclose(e.get("y")[0], g[tag + "_y"][i], atol=1e-11)
            o32, obs64, rew, _ = e.step(a)
            obs = o32[0]
            np.testing.assert_allclose(e.get("x")[0], g[tag + "_x"][i], rtol=1e-13)
            np.testing.assert_allclose(rew[0], g[tag + "_rew"][i], rtol=1e-13, atol=1e-11)
            i += 1
        x = e.get("x")[0]


@pytest.mark.parametrize("tag,kw", [("dist", dict(reward_type="distance")), ("sq", dict(reward_type="square_distance")),
                                    ("sparse", dict(reward_type="sparse")),
                                    ("zero", dict(reward_type="distance", noise_scale=0.0))])
@pytest.mark.parametrize("pol", ["prior", "resid"])
def test_wt_rollouts(tag, kw, pol):
    g = load_golden("wt_rollouts.npz")
    p = f"{tag}_{pol}_"
    for s in range(g[p + "params"].shape[0]):
        e = oracle.OracleWT(1, **kw)
        o0 = g[p + "obs0"][s]
        obs = e.reset(draws=np.array([[*g[p + "params"][s], o0[0], o0[1], o0[2]]]))
        np.testing.assert_array_equal(obs[0], o0.astype(np.float32))
        for t in range(200):
            obs, obs64, rew, done = e.step(g[p + "act"][s, t], noise=g[p + "noise"][s, t][None])
            np.testing.assert_allclose(obs64[0], g[p + "obs"][s, t], rtol=1e-13, atol=TOL)
            np.testing.assert_allclose(rew[0], g[p + "rew"][s, t], rtol=1e-13, atol=TOL)
            assert bool(done[0]) == bool(g[p + "done"][s, t])
        if pol == "resid":
            K = np.array([0., 0.4, -0.4, 0.])
            seen = np.concatenate([o0[None], g[p + "obs"][s, :-1]]).astype(np.float32)
            act = oracle.residual_action(g[p + "a_pre"][s], seen, -K)
            np.testing.assert_allclose(act, g[p + "act"][s], rtol=0, atol=3e-7)


def test_wt_chain_and_keep():
    g = load_golden("wt_rollouts.npz")
    e = oracle.OracleWT(1)
    E = g["chain_params"].shape[0]
    dr = [np.array([[*g["chain_params"][k], *g["chain_obs0"][k][:3]]]) for k in range(E)]
    obs = e.reset(draws=dr[0])
    for k in range(E):
        for t in range(200):
            obs, obs64, rew, done = e.step(g["chain_act"][k, t], noise=g["chain_noise"][k, t][None], auto_reset=True,
                                           reset_draws=dr[(k + 1) % E])
            np.testing.assert_allclose(rew[0], g["chain_rew"][k, t], rtol=1e-13, atol=TOL)
            if t < 199:
                np.testing.assert_allclose(obs64[0], g["chain_obs"][k, t], rtol=1e-13, atol=TOL)
            else:
                assert done[0]
                np.testing.assert_array_equal(obs[0], g["chain_obs0"][(k + 1) % E].astype(np.float32))
    # reset_changable_parameters + if_reset_all False
    e = oracle.OracleWT(1, resample_every=0)
    for f, v in zip(("a1", "a2", "Kp"), g["keep_params"][0]):
        e.set(f, v)
    e.reset(draws=np.array([[9, 9, 9, *g["keep_obs0"][0][:3]]]))
    for t in range(200):
        _, obs64, rew, _ = e.step(g["keep_act"][0, t], noise=g["keep_noise"][0, t][None])
        np.testing.assert_allclose(obs64[0], g["keep_obs"][0, t], rtol=1e-13, atol=TOL)


@pytest.mark.parametrize("tag", ["nominal", "robust1", "robust3"])
def test_wt_stepresponse(tag):
    g = load_golden("wt_stepresponse.npz")
    a1, a2, Kp, T = g[tag + "_params"]
    T = int(T)
    K = np.array([0., 0.4, -0.4, 0.])
    e = oracle.OracleWT(1, max_steps=T, resample_every=0, noise_scale=0.0)
    for f, v in zip(("a1", "a2", "Kp"), (a1, a2, Kp)):
        e.set(f, v)
    h1 = h2 = 0.0
    i = 0
    for r in [3., 6., 9., 4., 2.]:
        e.reset(draws=np.array([[0, 0, 0, h1, h2, r]]))
        obs = np.array([h1, h2, r, 0.0], dtype=np.float32)
        for n in range(T):
            a = float(obs @ (-K))
            o32, obs64, rew, _ = e.step(a, noise=np.zeros((1, 2)))
            obs = o32[0]
            np.testing.assert_allclose(obs64[0], g[tag + "_obs"][i], rtol=1e-12, atol=TOL)
            np.testing.assert_allclose(rew[0], g[tag + "_rew"][i], rtol=1e-12, atol=TOL)
            i += 1
        h1, h2 = e.get("h1")[0], e.get("h2")[0]


@pytest.mark.parametrize("S", [1, 4, 10])
def test_wt_stacking(S):
    g = load_golden("wt_stacking.npz")
    p = f"s{S}_"
    e = oracle.OracleWT(1, num_stack=S)
    assert e.obs_dim == 3 * S
    for k in range(2):
        o0 = g[p + "obs0"][k]
        assert o0.shape == (3 * S,)
        obs = e.reset(draws=np.array([[*g[p + "params"][k], *o0[-3:]]]))
        np.testing.assert_array_equal(obs[0], o0.astype(np.float32))
        for t in range(24):
            obs, obs64, rew, done = e.step(g[p + "act"][k, t], noise=g[p + "noise"][k, t][None])
            np.testing.assert_allclose(obs64[0], g[p + "obs"][k, t], rtol=1e-13, atol=TOL)
            np.testing.assert_allclose(rew[0], g[p + "rew"][k, t], rtol=1e-13, atol=TOL)


@pytest.mark.parametrize("lam", [0.97, 0.99])
def test_gae(lam):
    g = load_golden("gae.npz")
    # golden arrays are [lane, T] (episode-major, the order the reference buffer is filled); oracle is [T, lane]
    r_sum, adv = oracle.gae(g["reward"].T, g["mask"].T, g["value"].T, lam)
    np.testing.assert_allclose(r_sum.T, g[f"r_sum_{lam}"], rtol=2e-6, atol=2e-6)
    a = adv.T.astype(np.float64)
    want = g[f"adv_{lam}"]
    # the reference normalises over the whole buffer with torch's unbiased std (agent.py:707)
    a = (a - a.mean()) / (a.std(ddof=1) + 1e-5)
    np.testing.assert_allclose(a, want, rtol=2e-5, atol=2e-5)


def test_gae_plain():
    g = load_golden("gae.npz")
    r_sum, adv = oracle.gae(g["reward"].T, g["mask"].T, g["value"].T, 0.0, use_gae=False)
    np.testing.assert_allclose(r_sum.T, g["r_sum_noGAE"], rtol=2e-6, atol=2e-6)
    a = adv.T.astype(np.float64)
    a = (a - a.mean()) / (a.std(ddof=1) + 1e-5)
    np.testing.assert_allclose(a, g["adv_noGAE"], rtol=2e-5, atol=2e-5)


def _sd(g, tag):
    return {k[len(tag) + 1:]: g[k] for k in g.files if k.startswith(tag + ".")}


def test_nets_forward():
    g = load_golden("nets.npz")
    v = oracle.critic_forward(g["x3"], _sd(g, "critic3"))
    np.testing.assert_allclose(v, g["critic3:forward"], rtol=2e-5, atol=2e-5)
    for tag, x in (("modular3", g["x3"]), ("modular4", g["x4"])):
        sd = _sd(g, tag)
        mean = oracle.modular_actor_mean(x, sd).astype(np.float64)
        std = np.exp(sd["a_std_log"].astype(np.float64))
        np.testing.assert_allclose(mean + g["eps"] * std, g[f"{tag}:action"], rtol=2e-5, atol=2e-5)
        fwd = np.tanh(mean) + x.astype(np.float64) @ sd["priorK"].astype(np.float64)
        np.testing.assert_allclose(fwd, g[f"{tag}:forward"], rtol=2e-5, atol=2e-5)
        lp = -(sd["a_std_log"] + np.log(np.sqrt(2 * np.pi)) + 0.5 * ((mean - g["a1"]) / std) ** 2).sum(1)
        np.testing.assert_allclose(lp, g[f"{tag}:logprob"], rtol=1e-4, atol=1e-4)
    sd = _sd(g, "resid3")
    mean = oracle.plain_actor_mean(g["x3"], sd).astype(np.float64)
    np.testing.assert_allclose(np.tanh(mean) + g["x3"] @ sd["priorK"], g["resid3:forward"], rtol=2e-5, atol=2e-5)
    sd = _sd(g, "ppo3")
    mean = oracle.plain_actor_mean(g["x3"], sd).astype(np.float64)
    np.testing.assert_allclose(np.tanh(mean), g["ppo3:forward"], rtol=2e-5, atol=2e-5)


# ---- TD3 optimizer step (oracle/td3.py) against the reference's own update_net (elegantrl/agent.py:276-376) ------------------------
def _td3_sd(g, prefix):
    return {k[len(prefix) + 1:]: g[k] for k in g.files if k.startswith(prefix + ".")}


def test_td3_oracle_matches_reference_weights_after_six_steps():
    """td3_update.npz: the reference draws sample_batch's rows and the smoothing noise from torch's CPU generator seeded 77, in the
    order randint, randn_like per step; the same draws are replayed here.  float64 restatement vs the reference's float32 run:
    2e-6 on every weight of the four nets after 6 optimizer steps (three delayed soft updates)."""
    import torch
    from oracle import td3
    g = load_golden("td3_update.npz")
    net_dim, target_step, batch, repeat = (int(v) for v in g["td3:hyper"][:4])
    o = td3.Td3(_td3_sd(g, "td3:act0"), _td3_sd(g, "td3:act0"), _td3_sd(g, "td3:cri0"), _td3_sd(g, "td3:cri0"),
                lr=float(g["td3:hyper"][4]), tau=float(g["td3:hyper"][5]), policy_noise=float(g["td3:hyper"][7]),
                update_freq=int(g["td3:hyper"][8]))
    state, other = g["td3:state"], g["td3:other"]
    gen = torch.Generator().manual_seed(77)
    obj = None
    for i in range(target_step * repeat):
        idx = torch.randint(len(state) - 1, (batch,), generator=gen).numpy()
        eps = torch.randn((batch, 1), generator=gen).numpy()[:, 0]
        obj = o.step(i, state, other, idx, idx + 1, eps)
    for tag, net in (("act1", o.act), ("cri1", o.cri), ("act_target1", o.act_t), ("cri_target1", o.cri_t)):
        want = _td3_sd(g, f"td3:{tag}")
        for k in want:
            np.testing.assert_allclose(net[k], want[k], rtol=0, atol=2e-6, err_msg=f"{tag}.{k}")
    np.testing.assert_allclose([obj[0], obj[1] / 2], g["td3:obj"], rtol=1e-5, atol=1e-6)


def test_td3_oracle_matches_reference_gradients_at_batch_4096():
    """td3_update_multi.npz: width 128, batch 4 096, the reference's recorded rows and noise.  First-step .grad of every critic and
    actor parameter within 2e-6 of its tensor's largest entry (float64 vs the reference's float32 autograd), weights of all four nets
    after step 1 and after step 4 within 2e-6."""
    from oracle import td3
    g = load_golden("td3_update_multi.npz")
    o = td3.Td3(_td3_sd(g, "td3m:act0"), _td3_sd(g, "td3m:act_target0"), _td3_sd(g, "td3m:cri0"), _td3_sd(g, "td3m:cri_target0"),
                lr=float(g["td3m:hyper"][4]), tau=float(g["td3m:hyper"][5]), policy_noise=float(g["td3m:hyper"][7]),
                update_freq=int(g["td3m:hyper"][8]))
    state, other, idx, noise = g["td3m:state"], g["td3m:other"], g["td3m:indices"].astype(np.int64), g["td3m:noise"]
    assert idx.shape == (4, 4096) and noise.shape == (4, 4096)
    obj = None
    for i in range(4):
        obj = o.step(i, state, other, idx[i], idx[i] + 1, noise[i])
        if i == 0:
            for net_tag, grads in (("cri", obj[2]), ("act", obj[3])):
                for k, got in grads.items():
                    want = g[f"td3m:grad1:{net_tag}.{k}"]
                    np.testing.assert_allclose(got.reshape(want.shape), want, rtol=0, atol=2e-6 * max(np.abs(want).max(), 1e-3),
                                               err_msg=f"grad {net_tag}.{k}")
            for tag, net in (("act_step1", o.act), ("cri_step1", o.cri), ("act_target_step1", o.act_t), ("cri_target_step1", o.cri_t)):
                want = _td3_sd(g, f"td3m:{tag}")
                for k in want:
                    np.testing.assert_allclose(net[k], want[k], rtol=0, atol=2e-6, err_msg=f"{tag}.{k}")
    for tag, net in (("act1", o.act), ("cri1", o.cri), ("act_target1", o.act_t), ("cri_target1", o.cri_t)):
        want = _td3_sd(g, f"td3m:{tag}")
        for k in want:
            np.testing.assert_allclose(net[k], want[k], rtol=0, atol=2e-6, err_msg=f"{tag}.{k}")
    np.testing.assert_allclose([obj[0], obj[1] / 2], g["td3m:obj"], rtol=1e-5, atol=1e-6)
