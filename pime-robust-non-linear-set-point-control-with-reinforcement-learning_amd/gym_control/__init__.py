"""Env ids and constructor kwargs of the reference's registry (/root/reference/gym_control/__init__.py:3-142),
registered with pime_amd.gym_compat so that `gym.make(id, **overrides)` (train.py:87-103) works unchanged.

Every id builds a ONE-instance facade over the HIP kernels (envs/*.py); the throughput path does not go through
gym.make but constructs pime_amd.vec_env.VecPH / VecWaterTank directly (see `make_vec`)."""
import numpy as np

from .. import gym_compat as gym

PH_V35 = "PH1DChangingParamUniformGoalIntegrator-SqaureDistance-v35"
PH_NOIB_V35 = "PH1DChangingParamUniformGoalIntegrator-SqaureDistance-NoIB-v35"
WT_GOAL = "NonLinearWaterTankChangingParamUniformGoal-SquareDistance-v2"
WT_INTEGRATOR = "NonLinearWaterTankChangingParamUniformGoalIntegrator-SquareDistance-v2"
WT_STACKING = "NonLinearWaterTankChangingParamUniformGoalStacking{}-SquareDistance-v2"

_PH_KW = dict(P_control_K=np.array([-0.02, 0.02, 0.035]), P_control_L=None, reward_type="square_distance",
              max_episode_steps=50, MHCl=np.arange(0., 1, step=0.00001))
_WT_KW = dict(reset_from_last_state=False, max_step=200, a1=[0.0015, 0.0024], a2=[0.0015, 0.0024], A1=1, A2=1,
              Kp=[0.07, 0.17], G=980, sample_t=2, n_discrete=20, controller_type="P", reward_type="square_distance",
              gamma=0.99)


def _stack_K(num_stack):
    K = np.zeros(3 * num_stack)
    K[-3:] = np.array([0., 0.4, -0.4])
    return K


def _register_all():
    E = "pime_amd.gym_control.envs"
    specs = [
        (PH_V35, f"{E}:PH1DChangingParamUniformGoalIntegrator", dict(_PH_KW), 50),
        (PH_NOIB_V35, f"{E}:PH1DChangingParamUniformGoalIntegrator_NoBound", dict(_PH_KW), 50),
        # registered with a 4-element K for a 3-float observation in the reference too
        (WT_GOAL, f"{E}:NonLinearWaterTankChangingParamUniformGoal",
         dict(_WT_KW, P_control_K=np.array([0., 0.4, -0.4, 0.])), None),
        (WT_INTEGRATOR, f"{E}:NonLinearWaterTankChangingParamUniformGoalIntegrator",
         dict(_WT_KW, P_control_K=np.array([0., 0.4, -0.4, 0.])), None),
    ]
    for S in (4, 10, 1):
        specs.append((WT_STACKING.format(S), f"{E}:NonLinearWaterTankChangingParamUniformGoalStacking",
                      dict(_WT_KW, P_control_K=_stack_K(S), num_stack=S), None))
    for env_id, entry, kwargs, limit in specs:
        if env_id not in gym.registry.env_specs:
            gym.register(id=env_id, entry_point=entry, kwargs=kwargs, max_episode_steps=limit)


_register_all()


def make_vec(env_id, num_envs, device="cuda", **overrides):
    """Vectorised counterpart of gym.make(env_id): the same registered configuration on N lanes."""
    from ..vec_env import VecPH, VecWaterTank
    spec = gym.spec(env_id)
    kw = dict(spec._kwargs)
    vec_kw = {k: overrides.pop(k) for k in ("state_mode", "seed", "env_offset", "draws", "resample_every") if k in overrides}
    if env_id.startswith("PH1D"):   # ensemble ranges: class constants in the reference (ph.py:357-358), sweepable here
        vec_kw.update({k: overrides.pop(k) for k in ("qww_V", "qc_V") if k in overrides})
    kw.update(overrides)
    if env_id.startswith("PH1D"):
        return VecPH(num_envs, device=device, reward_type=kw["reward_type"], max_episode_steps=kw["max_episode_steps"],
                     integral_bound="NoIB" not in env_id, P_control_K=kw["P_control_K"],
                     MHCl_step=float(kw["MHCl"][1] - kw["MHCl"][0]), MHCl_len=len(kw["MHCl"]),
                     action_punishment=kw.get("action_punishment", 0.), action_change_punishment=kw.get("action_change_punishment", 0.),
                     **vec_kw)
    num_stack = kw.get("num_stack", 0) if "Stacking" in env_id else (1 if env_id == WT_GOAL else 0)
    return VecWaterTank(num_envs, device=device, reward_type=kw["reward_type"], max_step=kw["max_step"], num_stack=num_stack,
                        a1=kw["a1"], a2=kw["a2"], Kp=kw["Kp"], A1=kw["A1"], A2=kw["A2"], G=kw["G"], sample_t=kw["sample_t"],
                        n_discrete=kw["n_discrete"], noise_scale=kw.get("noise_scale", 0.01), P_control_K=kw["P_control_K"],
                        **vec_kw)
