"""Batched evaluation protocols: the reference's fixed set-point step responses and robustness sweeps, run for a whole
grid of plants at once (one env lane per plant) instead of one Python loop per plant.

replaces (protocols only, not the matplotlib plotting around them):
  pH          test_ph_policy_uniform_integrator   /root/reference/utils/test.py:1369-1407  (r = 10,6,3,8,5 x 50 steps)
              params_ph[1] plant grid             utils/test.py:1225-1236
  water tank  test_policy_uniform_integrator      utils/test.py:209-349                    (r = 3,6,9,4,2 x max_step)
              robust_test_nonlinear_watertank     utils/robust_test.py:4-46                (3 plants, max_step 500)

`policy` maps a float32 observation batch [N, D] to env actions [N]; None selects the prior controller -obs @ K
(the reference's `get_linear_action`).  All lanes share the set-point sequence; each lane keeps its own plant.

With `policy=None` (prior controller) or `agent=<a residual PPO agent>` the whole protocol is ONE launch of the fused evaluation
kernel (csrc/rollout_eval.hip, `pime_rollout_eval` with a set-point schedule: no per-step launches, no per-step device-to-host
reads; round 2 read four state fields back per step); an arbitrary `policy` callable keeps the step-per-launch loop.
"""
import numpy as np
import torch

PH_SETPOINTS = (10., 6., 3., 8., 5.)
WT_SETPOINTS = (3., 6., 9., 4., 2.)
PH_PARAM_GRID = ((0.005, 0.0025), (0.005, 0.0015), (0.015, 0.0025), (0.015, 0.0015), (0.001, 0.002), (0.001, 0.0022),
                 (0.001, 0.0018), (0.0007, 0.002), (0.0013, 0.002))          # utils/test.py:1225-1236 (qww_V, qc_V)
WT_ROBUST_PLANTS = ((0.0024, 0.0019, 0.12), (0.0024, 0.0015, 0.12), (0.0024, 0.0015, 0.07))  # robust_test.py:13-44


def _prior(env):
    k = torch.as_tensor(-env.K, dtype=torch.float64, device=env.device)
    return lambda obs: obs.double() @ k


def _fused_policy(env, policy, agent):
    """(packed actor or None, priorK) when the fused evaluation kernel can run the protocol, else None."""
    if policy is not None or not hasattr(env, "eval_supported"):
        return None
    if agent is None:
        return (None, -env.K) if env.eval_supported(None, trace=True, schedule=True) else None
    fused = agent.fused_eval_policy(env) if hasattr(agent, "fused_eval_policy") else None
    return fused if fused is not None and env.eval_supported(fused[0], trace=True, schedule=True) else None


def ph_step_response(env, policy=None, setpoints=PH_SETPOINTS, steps=50, plants=None, agent=None):
    """env: VecPH.  plants: optional [N, 2] (qww_V, qc_V) written before the run (the plant IS rebuilt, unlike the
    reference's set_params -- SURVEY.md App. C.3).  Returns dict of [len(setpoints)*steps, N] float64 arrays
    y, r, I, action, reward (and x) exactly in the order the reference protocol appends them."""
    fused = _fused_policy(env, policy, agent)
    env.set_reset_all(False)
    env.set_max_step(2 ** 30)                      # the protocol ignores TimeLimit's done and runs `steps` per segment
    if plants is not None:
        plants = np.asarray(plants, dtype=np.float64)
        env.set_params(plants[:, 0], plants[:, 1])
    if fused is not None and len(setpoints) <= 16:
        env.reset()
        env.set_field("x", np.zeros(env.num_envs))          # the protocol starts from state 0 (utils/test.py:1375)
        _, tr = env.rollout_eval(fused[0], fused[1], len(setpoints) * steps, setpoints=setpoints, seg_len=steps, want_trace=True)
        tr = tr.cpu().numpy()
        return {"y": tr[:, 0], "r": tr[:, 1], "I": tr[:, 2], "action": tr[:, 3], "reward": tr[:, 4], "x": tr[:, 5]}
    policy = policy or (agent.act if agent is not None else _prior(env))
    out = {k: [] for k in ("y", "r", "I", "action", "reward", "x")}
    last_x = np.zeros(env.num_envs)
    for r in setpoints:
        env.reset()
        env.set_field("x", last_x)
        env.set_field("r", float(r))
        obs = env.observe().clone()
        for _ in range(steps):
            a = policy(obs)
            out["y"].append(env.get_field("y")); out["r"].append(env.get_field("r")); out["I"].append(env.get_field("I"))
            out["action"].append(a.detach().double().cpu().numpy().reshape(-1))
            nxt, rew, _ = env.step(a.detach(), auto_reset=False)
            out["reward"].append(rew.double().cpu().numpy()); out["x"].append(env.get_field("x"))
            obs = nxt.clone()
        last_x = env.get_field("x")
    return {k: np.stack(v) for k, v in out.items()}


def wt_step_response(env, policy=None, setpoints=WT_SETPOINTS, steps=None, plants=None, agent=None):
    """env: VecWaterTank (Integrator observation).  plants: optional [N, 3] (a1, a2, Kp).  Returns obs [S*steps, N, D],
    action and reward [S*steps, N]; tank levels are carried from one set-point segment to the next."""
    fused = _fused_policy(env, policy, agent)
    steps = steps or env.max_step
    env.set_reset_all(False)
    env.set_max_step(max(steps, env.max_step))
    if plants is not None:
        plants = np.asarray(plants, dtype=np.float64)
        env.reset_changable_parameters(plants[:, 0], plants[:, 1], plants[:, 2])
    if fused is not None and len(setpoints) <= 16 and env.num_stack == 0:
        env.reset()
        env.set_field("h1", np.zeros(env.num_envs)); env.set_field("h2", np.zeros(env.num_envs))   # utils/test.py:219-221
        _, tr = env.rollout_eval(fused[0], fused[1], len(setpoints) * steps, setpoints=setpoints, seg_len=steps, want_trace=True)
        tr = tr.cpu().numpy()
        return {"obs": np.ascontiguousarray(np.transpose(tr[:, :4], (0, 2, 1))), "reward": tr[:, 4], "action": tr[:, 5]}
    policy = policy or (agent.act if agent is not None else _prior(env))
    out = {k: [] for k in ("obs", "action", "reward")}
    h1 = h2 = np.zeros(env.num_envs)
    for r in setpoints:
        env.reset()
        env.set_field("h1", h1); env.set_field("h2", h2); env.set_field("r", float(r))
        obs = env.observe().clone()
        for _ in range(steps):
            a = policy(obs)
            nxt, rew, _ = env.step(a.detach(), auto_reset=False)
            out["action"].append(a.detach().double().cpu().numpy().reshape(-1))
            out["obs"].append(np.stack([env.get_field(f) for f in ("h1", "h2", "r", "I")], axis=1))
            out["reward"].append(rew.double().cpu().numpy())
            obs = nxt.clone()
        h1, h2 = env.get_field("h1"), env.get_field("h2")
    return {k: np.stack(v) for k, v in out.items()}
