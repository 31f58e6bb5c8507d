#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/ by running the UNMODIFIED reference.

Container-only.  The reference (/root/reference) is imported here, never copied
and never shipped: only the small input/output vectors it produces are committed.
Its missing third-party deps (gym 0.18.0, control 0.9.1, pyserial) and its
missing in-repo `elegantrl/logger.py` are replaced by the stand-ins under
tests/golden/shims/ (written for this harness; see their docstrings).

Run (from the repo root; takes ~2 min, most of it the reference's own titration
table loop, /root/reference/gym_control/envs/ph.py:72-84):

    PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg python tests/golden/make_golden.py

Every array is float64 unless the reference itself produced float32.
What each fixture pins is listed in tests/golden/README.md.
"""
import hashlib
import json
import os
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("PIME_REFERENCE", "/root/reference")
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
sys.path.insert(0, os.path.join(HERE, "shims"))
os.environ.setdefault("MPLBACKEND", "Agg")

import numpy as np  # noqa: E402
import torch  # noqa: E402

torch.set_num_threads(1)

# --- the reference's own missing module: elegantrl/logger.py ---------
import elegantrl  # noqa: E402

_stub = types.ModuleType("elegantrl.logger")
_stub.record = lambda *a, **k: None
_stub.dump = lambda *a, **k: None
_stub.configure = lambda *a, **k: None
_stub.Figure = type("Figure", (), {})
_stub.make_output_format = lambda *a, **k: None
sys.modules["elegantrl.logger"] = _stub
elegantrl.logger = _stub

import gym  # noqa: E402  (shim)
import gym_control  # noqa: E402,F401  (reference: registers the env ids)

PH_ID = "PH1DChangingParamUniformGoalIntegrator-SqaureDistance-v35"
PH_NOIB_ID = "PH1DChangingParamUniformGoalIntegrator-SqaureDistance-NoIB-v35"
WT_ID = "NonLinearWaterTankChangingParamUniformGoalIntegrator-SquareDistance-v2"
WT_STACK_ID = "NonLinearWaterTankChangingParamUniformGoalStacking{}-SquareDistance-v2"


def save(name, **arrays):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrays)
    print(f"wrote {name}: {os.path.getsize(path) / 1024:.1f} KiB")


# =========================================================================================
# pH
# =========================================================================================
_ph_cache = {}


def make_ph(env_id=PH_ID, **kw):
    """gym.make is 8 s (table loop) -> cache one instance per (id, kwargs)."""
    key = (env_id, tuple(sorted(kw.items())))
    if key not in _ph_cache:
        _ph_cache[key] = gym.make(env_id, **kw)
    return _ph_cache[key]


def golden_ph_table():
    env = make_ph()
    pH = env.unwrapped.pH
    assert pH.shape == (100000,) and pH.dtype == np.float64
    idx = np.unique(np.concatenate([np.arange(0, 4097), np.arange(0, 100000, 64), [99999]]))
    save("ph_table.npz", k=idx.astype(np.int64), pH=pH[idx],
         sha256=np.frombuffer(hashlib.sha256(pH.tobytes()).digest(), dtype=np.uint8),
         MHCl_first=env.unwrapped.MHCl[:8], MHCl_last=env.unwrapped.MHCl[-8:])


def golden_ph_zoh():
    env = make_ph()
    u = env.unwrapped
    rng = np.random.RandomState(1234)
    pairs = [(0.005, 0.0015), (0.005, 0.0025), (0.015, 0.0015), (0.015, 0.0025), (0.01, 0.002)]
    pairs += [(rng.uniform(0.005, 0.015), rng.uniform(0.0015, 0.0025)) for _ in range(32)]
    keep = (u.qww_V, u.qc_V)
    out = []
    for a, c in pairs:
        u.qww_V, u.qc_V = a, c
        u.update_system()
        out.append((a, c, u.dsys.A.item(), u.dsys.B.item(), u.dsys.C.item()))
    u.qww_V, u.qc_V = keep
    u.update_system()
    save("ph_zoh.npz", table=np.array(out))  # columns qww_V, qc_V, A, B, C


def _ph_episode(env, policy, T=50):
    """One reset + T steps of the reference env.  policy(obs_f32, t) -> env action (float64)."""
    u = env.unwrapped
    obs = env.reset()
    rec = dict(params=(u.qww_V, u.qc_V), x0=float(u.state), r=float(u.r), obs0=obs.copy(),
               act=[], x=[], obs=[], rew=[], done=[])
    for t in range(T):
        a = policy(obs.astype(np.float32), t)
        obs, rew, done, _ = env.step(a)
        rec["act"].append(float(a)); rec["x"].append(float(u.state)); rec["obs"].append(obs.copy())
        rec["rew"].append(float(rew)); rec["done"].append(bool(done))
    return rec


def _pack(recs):
    out = {}
    for k in recs[0]:
        out[k] = np.array([r[k] for r in recs])
    return out


def golden_ph_rollouts():
    K = np.array([-0.02, 0.02, 0.035])
    n_seed = 16
    arrays = {}
    for tag, env_id, kw, setup in [
        ("v35", PH_ID, {}, None),
        ("noib", PH_NOIB_ID, {}, None),
        ("dist", PH_ID, {"reward_type": "distance"}, None),
        ("sparse", PH_ID, {"reward_type": "sparse"}, None),
        ("punish", PH_ID, {"action_punishment": 0.1, "action_change_punishment": 0.2},
         lambda u: setattr(u, "integral_punish", 0.05)),
    ]:
        env = make_ph(env_id, **kw)
        if setup is not None:
            setup(env.unwrapped)
        for pol in ("prior", "resid"):
            recs = []
            for s in range(n_seed):
                env.seed(s)
                np.random.seed(s)
                noise = np.random.RandomState(10_000 + s).standard_normal(50).astype(np.float32)
                if pol == "prior":
                    policy = lambda o, t: float(o @ (-K))  # noqa: E731  agent_residual.py:61, zero residual
                else:
                    # residual exploration as a zero-init actor produces it (net_residual.py:162,176-179)
                    policy = lambda o, t: float(np.tanh(noise[t] * np.float32(np.exp(-0.5))) + o @ (-K))  # noqa: E731
                rec = _ph_episode(env, policy)
                rec["a_pre"] = noise * np.float32(np.exp(-0.5))
                recs.append(rec)
            for k, v in _pack(recs).items():
                arrays[f"{tag}_{pol}_{k}"] = v
        if setup is not None:
            env.unwrapped.integral_punish = 0.0
    # three back-to-back episodes on one stream pair: pins the per-episode draw order (ph.py:410,420,424)
    env = make_ph()
    env.seed(100)
    np.random.seed(100)
    recs = [_ph_episode(env, lambda o, t: float(o @ (-K))) for _ in range(3)]
    for k, v in _pack(recs).items():
        arrays[f"chain_{k}"] = v
    # if_reset_all = False keeps the ensemble params (ph.py:428-445)
    env.seed(101)
    np.random.seed(101)
    env.reset()
    env.unwrapped.set_reset_all(False)
    recs = [_ph_episode(env, lambda o, t: float(o @ (-K))) for _ in range(2)]
    env.unwrapped.set_reset_all(True)
    for k, v in _pack(recs).items():
        arrays[f"keep_{k}"] = v
    save("ph_rollouts.npz", **arrays)


def golden_ph_stepresponse():
    """Protocol of utils/test.py:1369-1407 with the prior PI controller as the policy."""
    K = np.array([-0.02, 0.02, 0.035])
    env = make_ph()
    u = env.unwrapped
    env.seed(7)
    np.random.seed(7)
    keep = (u.qww_V, u.qc_V)
    u.set_reset_all(False)
    out = {}
    for tag, (qww, qc) in {"nominal": (0.01, 0.002), "corner": (0.015, 0.0015)}.items():
        u.qww_V, u.qc_V = qww, qc
        u.update_system()
        ys, rs, Is, acts, xs, rews = [], [], [], [], [], []
        last_state = np.zeros(1)
        for r in [10., 6, 3, 8, 5]:
            env.reset()
            env.set_state(last_state)
            state = env.set_r(r)
            for n in range(u.max_episode_steps):
                a = float(state.astype(np.float32) @ (-K))
                acts.append(a); rs.append(u.r); ys.append(u.y); Is.append(u.integrator)
                state, rew, done, info = env.step(a)
                xs.append(float(np.asarray(u.state).reshape(-1)[0])); rews.append(float(rew))
            last_state = u.state
        out.update({f"{tag}_y": np.array(ys), f"{tag}_r": np.array(rs), f"{tag}_I": np.array(Is, dtype=np.float64),
                    f"{tag}_act": np.array(acts), f"{tag}_x": np.array(xs), f"{tag}_rew": np.array(rews),
                    f"{tag}_params": np.array([qww, qc])})
    u.qww_V, u.qc_V = keep
    u.update_system()
    u.set_reset_all(True)
    save("ph_stepresponse.npz", **out)


# =========================================================================================
# water tank
# =========================================================================================
class _RecordingNormal:
    """Wraps np.random.normal so the two per-step noise draws (nonlinear_watertank.py:271-272) are recorded."""

    def __init__(self):
        self.log = []
        self._orig = np.random.normal

    def __enter__(self):
        def rec(loc=0.0, scale=1.0, size=None):
            v = self._orig(loc=loc, scale=scale, size=size)
            self.log.append(float(v))
            return v
        np.random.normal = rec
        return self

    def __exit__(self, *a):
        np.random.normal = self._orig


def _wt_episode(env, policy, T):
    obs = env.reset()
    a1, a2, Kp = env.get_changable_parameters()
    rec = dict(params=(a1, a2, Kp), obs0=obs.copy(), act=[], obs=[], rew=[], done=[], h=[], noise=[])
    with _RecordingNormal() as rn:
        for t in range(T):
            a = policy(obs.astype(np.float32), t)
            obs, rew, done, _ = env.step(a)
            rec["act"].append(float(a)); rec["obs"].append(obs.copy()); rec["rew"].append(float(rew))
            rec["done"].append(bool(done)); rec["h"].append((float(env.h1), float(env.h2)))
        rec["noise"] = np.array(rn.log).reshape(T, 2)
    return rec


def golden_wt_rollouts():
    K4 = np.array([0., 0.4, -0.4, 0.])
    arrays = {}
    T = 200
    for tag, kw in [("dist", dict(reward_type="distance", r=4.0)),          # what train.py makes (train.py:98-101)
                    ("sq", dict()),                                          # registered default
                    ("sparse", dict(reward_type="sparse", r=4.0)),
                    ("zero", dict(noise_scale=0., reward_type="distance", r=4.0))]:
        env = gym.make(WT_ID, **kw)
        for pol in ("prior", "resid"):
            recs = []
            for s in range(8 if tag == "dist" else 3):
                env.seed(s)
                np.random.seed(s)
                noise = np.random.RandomState(20_000 + s).standard_normal(T).astype(np.float32)
                if pol == "prior":
                    policy = lambda o, t: float(o @ (-K4))  # noqa: E731
                else:
                    policy = lambda o, t: float(np.tanh(noise[t] * np.float32(np.exp(-0.5))) + o @ (-K4))  # noqa: E731
                rec = _wt_episode(env, policy, T)
                rec["a_pre"] = noise * np.float32(np.exp(-0.5))
                recs.append(rec)
            for k, v in _pack(recs).items():
                arrays[f"{tag}_{pol}_{k}"] = v
    # two chained episodes on one global stream (draw order nonlinear_watertank.py:891-893,912-913,810-811)
    env = gym.make(WT_ID, reward_type="distance", r=4.0)
    env.seed(100)
    np.random.seed(100)
    recs = [_wt_episode(env, lambda o, t: float(o @ (-K4)), T) for _ in range(2)]
    for k, v in _pack(recs).items():
        arrays[f"chain_{k}"] = v
    # if_reset_all False + reset_changable_parameters (nonlinear_watertank.py:899-900)
    env.reset_changable_parameters(0.0024, 0.0019, 0.12)
    env.unwrapped.if_reset_all = False
    np.random.seed(101)
    recs = [_wt_episode(env, lambda o, t: float(o @ (-K4)), T) for _ in range(1)]
    for k, v in _pack(recs).items():
        arrays[f"keep_{k}"] = v
    save("wt_rollouts.npz", **arrays)


def golden_wt_stepresponse():
    """utils/test.py:209-349 (r = 3,6,9,4,2; 200 steps each; carries h1,h2) on the robust-test plants
    of utils/robust_test.py:4-46, noise_scale = 0 (train.py --env_zero_noise)."""
    K4 = np.array([0., 0.4, -0.4, 0.])
    env = gym.make(WT_ID, noise_scale=0., reward_type="distance", r=4.0)
    env.seed(3)
    np.random.seed(3)
    env.unwrapped.if_reset_all = False
    out = {}
    for tag, (a1, a2, Kp, T) in {"nominal": (0.0019, 0.0019, 0.12, 200),
                                  "robust1": (0.0024, 0.0019, 0.12, 500),
                                  "robust3": (0.0024, 0.0015, 0.07, 500)}.items():
        env.reset_changable_parameters(a1, a2, Kp)
        env.unwrapped.max_step = T
        obs_l, act_l, rew_l = [], [], []
        h1 = h2 = 0.
        for r in [3., 6., 9., 4., 2.]:
            env.reset()
            env.set_state(h1, h2)
            state = env.set_r(r)
            for n in range(T):
                a = float(state.astype(np.float32) @ (-K4))
                state, rew, done, _ = env.step(a)
                obs_l.append(state.copy()); act_l.append(a); rew_l.append(float(rew))
            h1, h2 = env.h1, env.h2
        out.update({f"{tag}_obs": np.array(obs_l), f"{tag}_act": np.array(act_l), f"{tag}_rew": np.array(rew_l),
                    f"{tag}_params": np.array([a1, a2, Kp, T])})
    save("wt_stepresponse.npz", **out)


def golden_wt_stacking():
    arrays = {}
    for S in (1, 4, 10):
        env = gym.make(WT_STACK_ID.format(S), reward_type="distance", r=4.0)
        K = env.K
        env.seed(5)
        np.random.seed(5)
        recs = [_wt_episode(env, lambda o, t: float(o @ (-K)), 24) for _ in range(2)]
        for k, v in _pack(recs).items():
            arrays[f"s{S}_{k}"] = v
        arrays[f"s{S}_K"] = np.asarray(K, dtype=np.float64)
    save("wt_stacking.npz", **arrays)


# =========================================================================================
# agent side
# =========================================================================================
def _sd_to_np(prefix, sd):
    return {f"{prefix}.{k}": v.detach().cpu().numpy().copy() for k, v in sd.items()}


def golden_gae():
    from elegantrl.agent import AgentPPO
    ag = AgentPPO()
    ag.device = torch.device("cpu")
    ag.lambda_gae_adv = 0.97
    rng = np.random.RandomState(42)
    T, N = 50, 8
    rew = rng.standard_normal((N, T)).astype(np.float32) * 3 - 2
    val = rng.standard_normal((N, T)).astype(np.float32)
    mask = np.full((N, T), 0.99, dtype=np.float32)
    mask[:, -1] = 0.0
    mask[3, 20] = 0.0  # an early termination inside a lane
    flat = lambda x: torch.as_tensor(x.reshape(-1))  # noqa: E731  episode-major, as the reference buffer is filled
    out = {}
    for lam in (0.97, 0.99):
        ag.lambda_gae_adv = lam
        r_sum, adv = ag.compute_reward_gae(N * T, flat(rew), flat(mask), flat(val).unsqueeze(1))
        out[f"r_sum_{lam}"] = r_sum.numpy().reshape(N, T)
        out[f"adv_{lam}"] = adv.numpy().reshape(N, T)
    r_sum, adv = ag.compute_reward_adv(N * T, flat(rew), flat(mask), flat(val).unsqueeze(1))
    out["r_sum_noGAE"] = r_sum.numpy().reshape(N, T)
    out["adv_noGAE"] = adv.numpy().reshape(N, T)
    save("gae.npz", reward=rew, mask=mask, value=val, **out)


def golden_nets():
    from elegantrl.net import CriticAdv, CriticTwin, Actor, ActorPPO
    from elegantrl.net_residual import ActorResidualIntegratorModularPPO, ActorResidualPPO
    torch.manual_seed(11)
    out = {}
    x3 = torch.randn(64, 3) * torch.tensor([4., 4., 10.]) + torch.tensor([7., 7., 0.])
    x4 = torch.rand(64, 4) * torch.tensor([10., 10., 10., 50.]) - torch.tensor([0., 0., 0., 25.])
    a1 = torch.randn(64, 1)
    eps = torch.randn(64, 1)
    out.update(x3=x3.numpy(), x4=x4.numpy(), a1=a1.numpy(), eps=eps.numpy())

    def actor_block(tag, act, x):
        act.priorK.data = -torch.tensor({3: [[-0.02], [0.02], [0.035]], 4: [[0.], [0.4], [-0.4], [0.]]}[x.shape[1]])
        out.update(_sd_to_np(tag, act.state_dict()))
        orig = torch.randn_like
        torch.randn_like = lambda t, **k: eps.clone()  # inject the exploration noise (net_residual.py:178)
        try:
            with torch.no_grad():
                out[f"{tag}:forward"] = act(x).numpy()
                action, noise = act.get_action_noise(x)
                out[f"{tag}:action"] = action.numpy()
                assert torch.equal(noise, eps)
                out[f"{tag}:logprob"] = act.compute_logprob(x, a1).numpy()
        finally:
            torch.randn_like = orig

    m = ActorResidualIntegratorModularPPO(128, 3, 1, 1)
    # default init leaves the output layer tiny (std 0.1); perturb so tanh is exercised
    with torch.no_grad():
        m.net[-1].weight.mul_(8.0)
    actor_block("modular3", m, x3)
    m4 = ActorResidualIntegratorModularPPO(64, 4, 1, 1)
    with torch.no_grad():
        m4.net[-1].weight.mul_(8.0)
    actor_block("modular4", m4, x4)
    r3 = ActorResidualPPO(32, 3, 1)
    with torch.no_grad():
        r3.net[-1].weight.mul_(8.0)
    actor_block("resid3", r3, x3)

    c = CriticAdv(3, 128)
    out.update(_sd_to_np("critic3", c.state_dict()))
    with torch.no_grad():
        out["critic3:forward"] = c(x3).numpy()
    ct = CriticTwin(32, 4, 1)
    out.update(_sd_to_np("twin4", ct.state_dict()))
    with torch.no_grad():
        q1, q2 = ct.get_q1_q2(x4, a1)
        out["twin4:q1"] = q1.numpy(); out["twin4:q2"] = q2.numpy()
    a = Actor(32, 4, 1)
    out.update(_sd_to_np("actor4", a.state_dict()))
    with torch.no_grad():
        out["actor4:forward"] = a(x4).numpy()
    p = ActorPPO(32, 3, 1)
    out.update(_sd_to_np("ppo3", p.state_dict()))
    with torch.no_grad():
        out["ppo3:forward"] = p(x3).numpy()
        out["ppo3:logprob"] = p.compute_logprob(x3, a1).numpy()
    save("nets.npz", **out)


def golden_ppo_update_and_explore():
    """One full rollout chunk + one update_net of the reference ResidualIntegratorModularPPO on pH (N = 1),
    exactly as run.py:136-152,205-211 sequences it.  Records the buffer (which holds the exploration noise,
    so the rollout can be replayed with injected draws), the per-episode env draws, the state_dicts before
    and after the update and the minibatch indices the reference drew."""
    from elegantrl.agent_residual import AgentResidualIntegratorModularPPO, AgentResidualPPO
    out = {}
    _ppo_cases(out, [
        ("ph", AgentResidualIntegratorModularPPO, PH_ID, 32, 200, 64, 4, 0.99),
        ("wt", AgentResidualPPO, WT_STACK_ID.format(1), 32, 400, 128, 2, 0.97),
    ], with_eval=True)
    save("ppo_update.npz", **out)


def golden_ppo_update_wide():
    """The same explore_env + update_net recording at the widths the HIP kernels serve (the fused gradient kernels take
    net_dim 64 / 128 / 256; `ppo_update.npz` is net_dim 32): pH ModularPPO at 128 (run_ph_changing.sh:5), water-tank
    Integrator ModularPPO at 64, and the reference's live water-tank configuration, ResidualPPO on Stacking10 at
    net_dim 256 (run_watertank_changing.sh:20-27).  No evaluation episode."""
    from elegantrl.agent_residual import AgentResidualIntegratorModularPPO, AgentResidualPPO
    out = {}
    _ppo_cases(out, [
        ("ph128", AgentResidualIntegratorModularPPO, PH_ID, 128, 1000, 256, 2, 0.99),
        ("wt64", AgentResidualIntegratorModularPPO, WT_ID, 64, 400, 128, 2, 0.97),
        ("wts10_256", AgentResidualPPO, WT_STACK_ID.format(10), 256, 400, 128, 2, 0.97),
    ], with_eval=False)
    save("ppo_update_wide.npz", **out)


def golden_ppo_update_mod256():
    """The second block of the reference's water-tank script (run_watertank_changing.sh:11-18): ResidualIntegratorModularPPO,
    net_dim 256, on the Integrator observation -- the shape the HIP path serves through the 16-tile family's modular kernels
    (csrc/mlp16.hip: mlp16m_forward_kernel, ppo16m_kernel).  Same recording as `ppo_update_wide.npz`."""
    from elegantrl.agent_residual import AgentResidualIntegratorModularPPO
    out = {}
    _ppo_cases(out, [("wtmod256", AgentResidualIntegratorModularPPO, WT_ID, 256, 400, 128, 2, 0.97)], with_eval=False)
    save("ppo_update_mod256.npz", **out)


def golden_ppo_update_multi():
    """update_net at MULTI-WORKGROUP batch sizes, with the reference's own gradients (VERDICT r02 task 3).

    `ppo_update_wide.npz` has batch 128 / 256 = one workgroup of the fused HIP gradient kernel, so its slab reduction over many
    workgroups, the second-group accumulation (batch > 65 536) and the one-graph path at large B were pinned only to torch
    autograd on the same device.  Here: the reference pH ResidualIntegratorModularPPO, net_dim 128 (run_ph_changing.sh), ONE
    explore_env chunk of >= 8 192 transitions, then from the same initial weights and buffer
      mw  : update_net(batch 4 096, repeat 2)   -> 4 optimizer steps, 16 workgroups of 256 samples per net
      big : update_net(batch 70 000, repeat 20) -> 2 optimizer steps; 70 000 > 65 536 = 256 workgroups x 256 samples, so 18 of the
            workgroups take a second 256-sample group and accumulate (the reference's torch.randint draws with replacement, so a
            batch larger than the buffer is its ordinary code path, agent.py:630)
    recording per case: the minibatch indices, EVERY parameter's .grad after the first backward() (agent.py:655), the weights
    after the first optimizer.step() and after the last, and the returned losses."""
    from elegantrl.agent_residual import AgentResidualIntegratorModularPPO
    from elegantrl.env import PreprocessEnv
    from elegantrl.replay import ReplayBuffer
    out = {}
    tag, net_dim, target_step, lam = "ph128", 128, 8192, 0.99
    env = PreprocessEnv(make_ph(), if_print=False)
    seed = 5
    env.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    agent = AgentResidualIntegratorModularPPO()
    agent.lambda_gae_adv = lam
    agent.init(net_dim, env.state_dim, env.action_dim, env.n_integrator)
    agent.device = torch.device("cpu")
    agent.init_residual({"init_K": env.K.reshape(-1, 1)})
    agent.init_actor_zero()
    agent.fix_K()
    with torch.no_grad():
        agent.act.net[-1].weight.normal_(0, 0.05)
    buffer = ReplayBuffer(max_len=target_step + env.max_step, state_dim=env.state_dim, action_dim=1, if_on_policy=True,
                          if_per=False, if_gpu=True)
    sd_act0 = {k: v.clone() for k, v in agent.act.state_dict().items()}
    sd_cri0 = {k: v.clone() for k, v in agent.cri.state_dict().items()}
    out.update(_sd_to_np(f"{tag}:act0", sd_act0))
    out.update(_sd_to_np(f"{tag}:cri0", sd_cri0))
    steps = agent.explore_env(env, buffer, target_step, 1.0, 0.99)
    buffer.update_now_len_before_sample()
    out[f"{tag}:steps"] = np.array(steps)
    out[f"{tag}:buf_state"] = buffer.buf_state[:buffer.now_len].copy()
    out[f"{tag}:buf_other"] = buffer.buf_other[:buffer.now_len].copy()
    for case, batch, repeat, draw_seed in (("mw", 4096, 2, 99), ("big", 70000, 20, 98)):
        # same start for every case: the reference's own optimizer rebuild (init_actor_zero, agent.py:569-574), then the weights
        agent.init_actor_zero()
        agent.act.load_state_dict(sd_act0)
        agent.cri.load_state_dict(sd_cri0)
        idx_log, snaps = [], []
        orig_randint, orig_step = torch.randint, agent.optimizer.step

        def rec_randint(*a, **k):
            v = orig_randint(*a, **k)
            idx_log.append(v.numpy().astype(np.int32))
            return v

        def rec_step(*a, **k):
            if not snaps:   # the first backward() of the update: the reference's gradients of obj_united
                for net_tag, net in (("act", agent.act), ("cri", agent.cri)):
                    for name, p in net.named_parameters():
                        if p.grad is not None:
                            out[f"{tag}:{case}:grad1:{net_tag}.{name}"] = p.grad.detach().numpy().copy()
            r = orig_step(*a, **k)
            if not snaps:
                out.update(_sd_to_np(f"{tag}:{case}:act_step1", agent.act.state_dict()))
                out.update(_sd_to_np(f"{tag}:{case}:cri_step1", agent.cri.state_dict()))
            snaps.append(1)
            return r
        torch.randint, agent.optimizer.step = rec_randint, rec_step
        torch.manual_seed(draw_seed)
        obj_a, obj_c = agent.update_net(buffer, target_step, batch, repeat)
        torch.randint = orig_randint
        assert len(snaps) == int(repeat * buffer.now_len / batch) == len(idx_log)
        out[f"{tag}:{case}:indices"] = np.array(idx_log)
        out[f"{tag}:{case}:obj"] = np.array([obj_a, obj_c])
        out[f"{tag}:{case}:hyper"] = np.array([net_dim, target_step, batch, repeat, lam, 0.99, agent.learning_rate,
                                               agent.ratio_clip, agent.lambda_entropy])
        out.update(_sd_to_np(f"{tag}:{case}:act1", agent.act.state_dict()))
        out.update(_sd_to_np(f"{tag}:{case}:cri1", agent.cri.state_dict()))
    save("ppo_update_multi.npz", **out)


def _ppo_cases(out, cases, with_eval):
    from elegantrl.agent_residual import AgentResidualIntegratorModularPPO
    from elegantrl.env import PreprocessEnv
    from elegantrl.replay import ReplayBuffer
    for tag, Agent, env_id, net_dim, target_step, batch, repeat, lam in cases:
        modular = Agent is AgentResidualIntegratorModularPPO
        is_ph = env_id == PH_ID
        if is_ph:
            env = PreprocessEnv(make_ph(), if_print=False)
        else:
            env = PreprocessEnv(gym.make(env_id, reward_type="distance", r=4.0), if_print=False)
        seed = 3
        env.seed(seed)
        np.random.seed(seed)
        torch.manual_seed(seed)
        agent = Agent()
        agent.lambda_gae_adv = lam
        if modular:
            agent.init(net_dim, env.state_dim, env.action_dim, env.n_integrator)
        else:
            agent.init(net_dim, env.state_dim, env.action_dim)
        agent.device = torch.device("cpu")
        agent.init_residual({"init_K": env.K.reshape(-1, 1)})
        agent.init_actor_zero()
        agent.fix_K()
        # give the critic/actor something non-trivial to learn from: un-zero the output layer a little
        with torch.no_grad():
            agent.act.net[-1].weight.normal_(0, 0.05)
        buffer = ReplayBuffer(max_len=target_step + env.max_step, state_dim=env.state_dim, action_dim=1,
                              if_on_policy=True, if_per=False, if_gpu=True)
        out.update(_sd_to_np(f"{tag}:act0", agent.act.state_dict()))
        out.update(_sd_to_np(f"{tag}:cri0", agent.cri.state_dict()))

        # record per-episode env draws by wrapping reset
        draws = []
        u = env.unwrapped
        orig_reset = env.reset

        def rec_reset():
            s = orig_reset()
            if is_ph:
                draws.append((u.qww_V, u.qc_V, float(u.state), float(u.r)))
            else:
                draws.append((u.a1, u.a2, u.Kp, float(u.h1), float(u.h2), float(u.r)))
            return s
        env.reset = rec_reset
        with _RecordingNormal() as rn:
            steps = agent.explore_env(env, buffer, target_step, 1.0, 0.99)
        env.reset = orig_reset
        out[f"{tag}:steps"] = np.array(steps)
        out[f"{tag}:draws"] = np.array(draws)
        out[f"{tag}:step_noise"] = np.array(rn.log).reshape(-1, 2) if rn.log else np.zeros((0, 2))
        buffer.update_now_len_before_sample()
        out[f"{tag}:buf_state"] = buffer.buf_state[:buffer.now_len].copy()
        out[f"{tag}:buf_other"] = buffer.buf_other[:buffer.now_len].copy()

        # update_net with recorded minibatch indices
        idx_log = []
        orig_randint = torch.randint

        def rec_randint(*a, **k):
            v = orig_randint(*a, **k)
            idx_log.append(v.numpy().copy())
            return v
        torch.randint = rec_randint
        torch.manual_seed(99)
        obj_a, obj_c = agent.update_net(buffer, target_step, batch, repeat)
        torch.randint = orig_randint
        out[f"{tag}:indices"] = np.array(idx_log)
        out[f"{tag}:obj"] = np.array([obj_a, obj_c])
        out[f"{tag}:hyper"] = np.array([net_dim, target_step, batch, repeat, lam, 0.99, agent.learning_rate,
                                        agent.ratio_clip, agent.lambda_entropy])
        out.update(_sd_to_np(f"{tag}:act1", agent.act.state_dict()))
        out.update(_sd_to_np(f"{tag}:cri1", agent.cri.state_dict()))
        if not with_eval:
            continue
        # deterministic evaluation episode with the updated policy (run.py:600-619)
        from elegantrl.run import get_episode_return
        env.seed(17)
        np.random.seed(17)
        ret, n = get_episode_return(env, agent.act, torch.device("cpu"))
        out[f"{tag}:eval"] = np.array([ret, n])


def golden_td3_update():
    """AgentTD3.update_net of the reference (elegantrl/agent.py:276-341: twin critics, target policy smoothing, delayed soft
    updates) on a flat ring buffer of random transitions, CPU, torch seeded right before the call: weights of actor / critic /
    both targets before and after 6 update steps, and the returned losses.  The reference has no residual TD3 (SURVEY.md
    fact 5); this pins the TD3 pieces the build composes one from."""
    from elegantrl.agent import AgentTD3
    from elegantrl.replay import ReplayBuffer
    out = {}
    torch.manual_seed(21)
    agent = AgentTD3()
    agent.init(64, 4, 1)
    rng = np.random.RandomState(4)
    n = 500
    buf = ReplayBuffer(max_len=n + 8, state_dim=4, action_dim=1, if_on_policy=False, if_per=False, if_gpu=True)
    state = (rng.rand(n, 4) * np.array([10., 10., 10., 50.]) - np.array([0., 0., 0., 25.])).astype(np.float32)
    other = np.stack([-rng.rand(n) * 5, np.where(rng.rand(n) < 0.02, 0.0, 0.99), np.tanh(rng.randn(n))], axis=1).astype(np.float32)
    buf.extend_buffer(torch.as_tensor(state), torch.as_tensor(other))
    for tag, net in (("act0", agent.act), ("cri0", agent.cri)):
        out.update(_sd_to_np(f"td3:{tag}", net.state_dict()))
    torch.manual_seed(77)
    obj_a, obj_c = agent.update_net(buf, 3, 64, 2)
    for tag, net in (("act1", agent.act), ("cri1", agent.cri), ("act_target1", agent.act_target), ("cri_target1", agent.cri_target)):
        out.update(_sd_to_np(f"td3:{tag}", net.state_dict()))
    out["td3:state"], out["td3:other"] = state, other
    out["td3:obj"] = np.array([obj_a, obj_c])
    out["td3:hyper"] = np.array([64, 3, 64, 2, agent.learning_rate, agent.soft_update_tau, agent.explore_noise, agent.policy_noise,
                                 agent.update_freq])
    save("td3_update.npz", **out)


def golden_td3_update_multi():
    """AgentTD3.update_net at the batch size of BASELINE config 2 (4 096 = 256 workgroups of the fused HIP step), width 128, with
    the reference's own first-step gradients (VERDICT r03 task 1; what `ppo_update_multi` is for PPO).

    Reference AgentTD3 (elegantrl/agent.py:276-376), net_dim 128, state_dim 4, on a flat ring of 6 000 random transitions;
    update_net(target_step 2, batch 4 096, repeat 2) = 4 optimizer steps (steps 0 and 2 with the delayed soft updates).  Recorded:
    the sampled rows (torch.randint) and the smoothing-noise draws (torch.randn_like) of every step; the critic's .grad after the
    first obj_critic.backward() and the actor's after the first obj_actor.backward() (agent.py:317,326); all four nets after the
    first step and after the last; the per-step objectives the reference appends (agent.py:315,324) and the returned pair."""
    from elegantrl.agent import AgentTD3
    from elegantrl.replay import ReplayBuffer
    out = {}
    torch.manual_seed(31)
    agent = AgentTD3()
    agent.init(128, 4, 1)
    with torch.no_grad():   # targets that differ from the online nets, heads away from their tiny initial scale
        for net in (agent.act_target, agent.cri_target):
            for p_ in net.parameters():
                p_.add_(torch.randn_like(p_) * 0.02)
        agent.act.net[-1].weight.normal_(0, 0.1)
        agent.act_target.net[-1].weight.normal_(0, 0.1)
    rng = np.random.RandomState(9)
    n = 6000
    buf = ReplayBuffer(max_len=n + 8, state_dim=4, action_dim=1, if_on_policy=False, if_per=False, if_gpu=True)
    state = (rng.rand(n, 4) * np.array([10., 10., 10., 50.]) - np.array([0., 0., 0., 25.])).astype(np.float32)
    other = np.stack([-rng.rand(n) * 5, np.where(rng.rand(n) < 0.02, 0.0, 0.99), np.tanh(rng.randn(n))], axis=1).astype(np.float32)
    buf.extend_buffer(torch.as_tensor(state), torch.as_tensor(other))
    for tag, net in (("act0", agent.act), ("cri0", agent.cri), ("act_target0", agent.act_target), ("cri_target0", agent.cri_target)):
        out.update(_sd_to_np(f"td3m:{tag}", net.state_dict()))
    idx_log, noise_log, steps = [], [], {"cri": 0, "act": 0}
    orig_randint, orig_randn_like = torch.randint, torch.randn_like
    orig_cri_step, orig_act_step = agent.cri_optimizer.step, agent.act_optimizer.step

    def rec_randint(*a, **k):
        v = orig_randint(*a, **k)
        idx_log.append(v.numpy().astype(np.int32))
        return v

    def rec_randn_like(t, **k):
        v = orig_randn_like(t, **k)
        noise_log.append(v.numpy().reshape(-1).copy())
        return v

    def rec_cri_step(*a, **k):
        if steps["cri"] == 0:
            for name, p_ in agent.cri.named_parameters():
                out[f"td3m:grad1:cri.{name}"] = p_.grad.detach().numpy().copy()
        steps["cri"] += 1
        return orig_cri_step(*a, **k)

    def rec_act_step(*a, **k):
        if steps["act"] == 0:
            for name, p_ in agent.act.named_parameters():
                out[f"td3m:grad1:act.{name}"] = p_.grad.detach().numpy().copy()
        r = orig_act_step(*a, **k)
        steps["act"] += 1
        return r

    orig_soft = agent.soft_update
    snaps = []

    def rec_soft(tar, cur, tau):
        r = orig_soft(tar, cur, tau)
        if tar is agent.act_target and not snaps:   # end of step 0 (the last statement of a delayed step, agent.py:331)
            for tag, net in (("act_step1", agent.act), ("cri_step1", agent.cri), ("act_target_step1", agent.act_target),
                             ("cri_target_step1", agent.cri_target)):
                out.update(_sd_to_np(f"td3m:{tag}", net.state_dict()))
            snaps.append(1)
        return r
    torch.randint, torch.randn_like = rec_randint, rec_randn_like
    agent.cri_optimizer.step, agent.act_optimizer.step, agent.soft_update = rec_cri_step, rec_act_step, rec_soft
    torch.manual_seed(78)
    try:
        obj_a, obj_c = agent.update_net(buf, 2, 4096, 2)
    finally:
        torch.randint, torch.randn_like = orig_randint, orig_randn_like
    assert steps == {"cri": 4, "act": 4} and len(idx_log) == 4 and len(noise_log) == 4 and snaps
    for tag, net in (("act1", agent.act), ("cri1", agent.cri), ("act_target1", agent.act_target), ("cri_target1", agent.cri_target)):
        out.update(_sd_to_np(f"td3m:{tag}", net.state_dict()))
    out["td3m:state"], out["td3m:other"] = state, other
    out["td3m:indices"], out["td3m:noise"] = np.array(idx_log), np.array(noise_log, dtype=np.float32)
    out["td3m:obj"] = np.array([obj_a, obj_c])
    out["td3m:hyper"] = np.array([128, 2, 4096, 2, agent.learning_rate, agent.soft_update_tau, agent.explore_noise, agent.policy_noise,
                                  agent.update_freq])
    save("td3_update_multi.npz", **out)


def main():
    only = set(sys.argv[1:])
    jobs = dict(ph_table=golden_ph_table, ph_zoh=golden_ph_zoh, ph_rollouts=golden_ph_rollouts,
                ph_stepresponse=golden_ph_stepresponse, wt_rollouts=golden_wt_rollouts,
                wt_stepresponse=golden_wt_stepresponse, wt_stacking=golden_wt_stacking,
                gae=golden_gae, nets=golden_nets, ppo_update=golden_ppo_update_and_explore,
                ppo_update_wide=golden_ppo_update_wide, ppo_update_mod256=golden_ppo_update_mod256,
                ppo_update_multi=golden_ppo_update_multi,
                td3_update=golden_td3_update, td3_update_multi=golden_td3_update_multi)
    for name, fn in jobs.items():
        if only and name not in only:
            continue
        fn()
    meta = dict(numpy=np.__version__, torch=torch.__version__, reference=REF,
                note="generated by tests/golden/make_golden.py from the unmodified reference")
    with open(os.path.join(HERE, "golden_meta.json"), "w") as f:
        json.dump(meta, f, indent=1)


if __name__ == "__main__":
    main()
