"""Env adapter (interface of /root/reference/elegantrl/env.py:11-88,194-245): float32 observations,
action * action_max, and the metadata the run loop reads (env_name, state_dim, action_dim, action_max, max_step,
if_discrete, target_return).  Vectorised envs (pime_amd.vec_env) already speak float32 device tensors and carry
this metadata themselves, so `PreprocessEnv` passes them through untouched."""
import numpy as np

from .. import gym_compat as gym


def get_gym_env_info(env, if_print=True):
    assert isinstance(env, gym.Env), "expected a gym-style env (pime_amd.gym_compat.Env)"
    env_name = env.unwrapped.spec.id
    shape = env.observation_space.shape
    state_dim = shape[0] if len(shape) == 1 else shape
    target_return = getattr(env, "target_return", None)
    if target_return is None:
        target_return = getattr(env.spec, "reward_threshold", None)
    if target_return is None:
        target_return = 2 ** 16
    max_step = getattr(env, "max_step", None)              # water tank: its own attribute
    if max_step is None:
        max_step = getattr(env, "_max_episode_steps", None)  # pH: gym's TimeLimit (registered max_episode_steps=50)
    if max_step is None:
        max_step = 2 ** 10
    if_discrete = isinstance(env.action_space, gym.Discrete)
    if if_discrete:
        action_dim, action_max = env.action_space.n, 1
    elif isinstance(env.action_space, gym.Box):
        action_dim = env.action_space.shape[0]
        action_max = float(env.action_space.high[0])
        assert not any(env.action_space.high + env.action_space.low), "action space must be symmetric"
    else:
        raise RuntimeError("set if_discrete / action_dim / action_max manually for this action space")
    if if_print:
        print(f"\n| env_name:  {env_name}, action space if_discrete: {if_discrete}"
              f"\n| state_dim: {state_dim:4}, action_dim: {action_dim}, action_max: {action_max}"
              f"\n| max_step:  {max_step:4}, target_return: {target_return}")
    return env_name, state_dim, action_dim, action_max, max_step, if_discrete, target_return


class PreprocessEnv(gym.Wrapper):
    def __new__(cls, env, if_print=True, data_type=np.float32):
        if hasattr(env, "num_envs"):   # vectorised env: nothing to adapt
            return env
        return super().__new__(cls)

    def __init__(self, env, if_print=True, data_type=np.float32):
        env = gym.make(env) if isinstance(env, str) else env
        super().__init__(env)
        self.data_type = data_type
        (self.env_name, self.state_dim, self.action_dim, self.action_max, self.max_step, self.if_discrete,
         self.target_return) = get_gym_env_info(env, if_print)

    def reset(self):
        return self.env.reset().astype(self.data_type)

    def step(self, action):
        state, reward, done, info = self.env.step(action * self.action_max)
        return state.astype(self.data_type), reward, done, info
