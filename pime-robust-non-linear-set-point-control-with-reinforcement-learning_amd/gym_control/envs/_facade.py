"""Shared machinery of the one-instance gym-style facades: a single-lane float64 env handle on the GPU whose
random draws are taken from the SAME numpy streams, at the SAME call sites, as the reference's Python envs."""
import numpy as np
import torch

from ... import gym_compat as gym
from ...vec_env import CallbackDraws


def device_property(field, cast=float):
    """Python attribute <-> device state word of lane 0 (the harness reads and writes env.h1, env.a1, ... directly:
    utils/robust_test.py:13-44, utils/test.py:1255-1406)."""
    def getter(self):
        return cast(self._vec.get_field(field)[0])

    def setter(self, value):
        self._vec.set_field(field, float(np.asarray(value).reshape(-1)[0]))
    return property(getter, setter)


class SingleEnvFacade(gym.Env):
    metadata = {"render.modes": ["human"]}
    _vec = None

    def _finish_init(self, vec, obs_low, obs_high, seed):
        self._vec = vec
        self.observation_space = gym.Box(low=obs_low, high=obs_high, dtype=np.float32)
        self.action_space = gym.Box(low=-np.ones(1), high=np.ones(1), dtype=np.float32)
        self.min_action, self.max_action = -1, 1
        self.seed(seed)
        self._device_action = torch.zeros(1, dtype=torch.float64, device=vec.device)

    def _draws(self, episode_fn, noise_fn=None):
        return CallbackDraws(lambda lane: episode_fn(), None if noise_fn is None else (lambda lane: noise_fn()))

    def seed(self, seed=None):
        self.np_random, seed = gym.np_random(seed)
        return [seed]

    def _obs64(self):
        raise NotImplementedError

    def _step_device(self, action):
        a = float(np.asarray(action, dtype=np.float64).reshape(-1)[0])
        self._device_action[0] = a
        _, rew, done = self._vec.step(self._device_action, auto_reset=False)
        return rew, done

    def close(self):
        if self._vec is not None:
            self._vec.close()

    def __deepcopy__(self, memo):
        """utils/test.py:1058-1059 deep-copies the env for evaluation: clone = new handle + copied state words."""
        clone = self._clone_blank()
        for f in self._copy_fields:
            clone._vec.set_field(f, self._vec.get_field(f))
        clone._vec._t_all, clone._vec._t_lanes = self._vec._t_all, None if self._vec._t_lanes is None else self._vec._t_lanes.copy()
        clone._vec._was_reset = self._vec._was_reset
        for k, v in self.__dict__.items():
            if k not in ("_vec", "_device_action", "observation_space", "action_space", "np_random"):
                clone.__dict__[k] = v
        memo[id(self)] = clone
        return clone
