import copy
import importlib

from .. import error


def _load(name):
    mod_name, attr_name = name.split(":")
    mod = importlib.import_module(mod_name)
    return getattr(mod, attr_name)


class EnvSpec(object):
    def __init__(self, id, entry_point=None, reward_threshold=None, nondeterministic=False,
                 max_episode_steps=None, kwargs=None):
        self.id = id
        self.entry_point = entry_point
        self.reward_threshold = reward_threshold
        self.nondeterministic = nondeterministic
        self.max_episode_steps = max_episode_steps
        self._kwargs = {} if kwargs is None else kwargs

    def make(self, **kwargs):
        if self.entry_point is None:
            raise error.Error("Attempting to make deprecated env {}".format(self.id))
        _kwargs = self._kwargs.copy()
        _kwargs.update(kwargs)
        cls = self.entry_point if callable(self.entry_point) else _load(self.entry_point)
        env = cls(**_kwargs)
        spec = copy.deepcopy(self)
        spec._kwargs = _kwargs
        env.unwrapped.spec = spec
        return env


class EnvRegistry(object):
    def __init__(self):
        self.env_specs = {}

    def make(self, path, **kwargs):
        spec = self.spec(path)
        env = spec.make(**kwargs)
        if env.spec.max_episode_steps is not None:
            from ..wrappers.time_limit import TimeLimit
            env = TimeLimit(env, max_episode_steps=env.spec.max_episode_steps)
        return env

    def all(self):
        return self.env_specs.values()

    def spec(self, path):
        try:
            return self.env_specs[path]
        except KeyError:
            raise error.Error("No registered env with id: {}".format(path))

    def register(self, id, **kwargs):
        if id in self.env_specs:
            raise error.Error("Cannot re-register id: {}".format(id))
        self.env_specs[id] = EnvSpec(id, **kwargs)


registry = EnvRegistry()


def register(id, **kwargs):
    return registry.register(id, **kwargs)


def make(id, **kwargs):
    return registry.make(id, **kwargs)


def spec(id):
    return registry.spec(id)
