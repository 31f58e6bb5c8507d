"""The slice of the gym 0.18 API the reference's hot path touches, so that `gym.make(id, **overrides)`,
`PreprocessEnv(env)` and `train.py`-style drivers keep working where `gym` itself is absent (it is absent in
the build image and on the GPU box; there is no package index).

Covers: Env / Wrapper (attribute forwarding of non-underscore names), spaces.Box (bounds cast to the space
dtype) and spaces.Discrete, envs.registration.register/make/registry.env_specs, wrappers.TimeLimit, and
utils.seeding.np_random (integer seed -> SHA-512 -> 64 bit -> MT19937 init_by_array, which decides the x0 / r
stream of the pH env: /root/reference/gym_control/envs/ph.py:123-125,420,424).

Use:  from pime_amd import gym_compat as gym
"""
import copy
import hashlib
import importlib
import os
import struct
import types

import numpy as np


class Error(Exception):
    pass


error = types.SimpleNamespace(Error=Error)


class _Logger:
    level = 30

    def set_level(self, level):
        self.level = level

    def warn(self, *a, **k):
        pass


logger = _Logger()


# ------------------------------------------------------------------------------------------------ core
class Env:
    metadata = {"render.modes": []}
    reward_range = (-float("inf"), float("inf"))
    spec = None
    action_space = None
    observation_space = None

    def step(self, action):
        raise NotImplementedError

    def reset(self):
        raise NotImplementedError

    def render(self, mode="human"):
        raise NotImplementedError

    def close(self):
        pass

    def seed(self, seed=None):
        return None

    @property
    def unwrapped(self):
        return self


class Wrapper(Env):
    def __init__(self, env):
        self.env = env
        self.action_space = env.action_space
        self.observation_space = env.observation_space
        self.reward_range = env.reward_range
        self.metadata = env.metadata

    def __getattr__(self, name):
        # only reached for attributes the wrapper itself lacks; private names never leak through
        if name.startswith("_"):
            raise AttributeError(f"attempted to get missing private attribute '{name}'")
        return getattr(self.env, name)

    @property
    def spec(self):
        return self.env.spec

    @property
    def unwrapped(self):
        return self.env.unwrapped

    def step(self, action):
        return self.env.step(action)

    def reset(self, **kwargs):
        return self.env.reset(**kwargs)

    def seed(self, seed=None):
        return self.env.seed(seed)

    def close(self):
        return self.env.close()


# ------------------------------------------------------------------------------------------------ spaces
class Box:
    def __init__(self, low, high, shape=None, dtype=np.float32):
        self.dtype = np.dtype(dtype)
        low, high = np.asarray(low), np.asarray(high)
        if shape is not None:
            low, high = np.full(shape, low), np.full(shape, high)
        self.shape = tuple(low.shape)
        self.low = low.astype(self.dtype)    # float64 bounds are cast: -np.ones(m)*0 becomes float32 -0.0
        self.high = high.astype(self.dtype)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

    def __repr__(self):
        return f"Box({self.low.min()}, {self.high.max()}, {self.shape}, {self.dtype})"


class Discrete:
    def __init__(self, n):
        self.n = int(n)
        self.shape = ()
        self.dtype = np.dtype(np.int64)


spaces = types.SimpleNamespace(Box=Box, Discrete=Discrete)


# ------------------------------------------------------------------------------------------------ seeding
def _bigint_from_bytes(data):
    data = data + b"\0" * (4 - len(data) % 4)
    words = struct.unpack(f"{len(data) // 4}I", data)
    return sum(w << (32 * i) for i, w in enumerate(words))


def _hash_seed(seed, max_bytes=8):
    return _bigint_from_bytes(hashlib.sha512(str(seed).encode("utf8")).digest()[:max_bytes])


def _create_seed(a=None, max_bytes=8):
    if a is None:
        return _bigint_from_bytes(os.urandom(max_bytes))
    if isinstance(a, str):
        raw = a.encode("utf8")
        return _bigint_from_bytes((raw + hashlib.sha512(raw).digest())[:max_bytes])
    if isinstance(a, (int, np.integer)):
        return int(a) % 2 ** (8 * max_bytes)
    raise Error(f"Invalid type for seed: {type(a)} ({a})")


def mt19937_key_from_seed(seed):
    """The init_by_array key gym 0.18 derives from an integer seed."""
    big = _hash_seed(_create_seed(seed))
    key = []
    while big > 0:
        big, word = divmod(big, 2 ** 32)
        key.append(word)
    return key or [0]


def np_random(seed=None):
    if seed is not None and not (isinstance(seed, (int, np.integer)) and seed >= 0):
        raise Error(f"Seed must be a non-negative integer or omitted, not {seed}")
    seed = _create_seed(seed)
    rng = np.random.RandomState()
    rng.seed(mt19937_key_from_seed(seed))
    return rng, seed


seeding = types.SimpleNamespace(np_random=np_random, create_seed=_create_seed, hash_seed=_hash_seed)
utils = types.SimpleNamespace(seeding=seeding)


# ------------------------------------------------------------------------------------------------ TimeLimit
class TimeLimit(Wrapper):
    def __init__(self, env, max_episode_steps=None):
        super().__init__(env)
        if max_episode_steps is None and env.spec is not None:
            max_episode_steps = env.spec.max_episode_steps
        self._max_episode_steps = max_episode_steps
        self._elapsed_steps = None

    def step(self, action):
        assert self._elapsed_steps is not None, "Cannot call env.step() before calling reset()"
        obs, reward, done, info = self.env.step(action)
        self._elapsed_steps += 1
        if self._elapsed_steps >= self._max_episode_steps:
            info["TimeLimit.truncated"] = not done
            done = True
        return obs, reward, done, info

    def reset(self, **kwargs):
        self._elapsed_steps = 0
        return self.env.reset(**kwargs)


wrappers = types.SimpleNamespace(TimeLimit=TimeLimit)


# ------------------------------------------------------------------------------------------------ registry
class EnvSpec:
    def __init__(self, id, entry_point=None, reward_threshold=None, max_episode_steps=None, kwargs=None,
                 nondeterministic=False):
        self.id = id
        self.entry_point = entry_point
        self.reward_threshold = reward_threshold
        self.max_episode_steps = max_episode_steps
        self.nondeterministic = nondeterministic
        self._kwargs = dict(kwargs or {})

    def make(self, **overrides):
        kwargs = dict(self._kwargs)
        kwargs.update(overrides)
        ctor = self.entry_point
        if not callable(ctor):
            mod, attr = ctor.split(":")
            ctor = getattr(importlib.import_module(mod), attr)
        env = ctor(**kwargs)
        spec = copy.copy(self)
        spec._kwargs = kwargs
        env.unwrapped.spec = spec
        return env


class _Registry:
    def __init__(self):
        self.env_specs = {}

    def register(self, id, **kwargs):
        if id in self.env_specs:
            raise Error(f"Cannot re-register id: {id}")
        self.env_specs[id] = EnvSpec(id, **kwargs)

    def spec(self, id):
        if id not in self.env_specs:
            raise Error(f"No registered env with id: {id}")
        return self.env_specs[id]

    def make(self, id, **kwargs):
        env = self.spec(id).make(**kwargs)
        if env.spec.max_episode_steps is not None:  # what puts TimeLimit(50) around the pH env
            env = TimeLimit(env, max_episode_steps=env.spec.max_episode_steps)
        return env

    def all(self):
        return self.env_specs.values()


registry = _Registry()
register = registry.register
make = registry.make
spec = registry.spec
envs = types.SimpleNamespace(registry=registry, registration=types.SimpleNamespace(
    register=register, make=make, registry=registry, EnvSpec=EnvSpec))
