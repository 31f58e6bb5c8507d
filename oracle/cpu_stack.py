"""CPU stand-ins built on the oracle, for (a) the host-logic tests that must run without a GPU (PPO update against
the reference golden, gloo data-parallel test) and (b) bench.py's `cpu_baseline` leg.

TEST INFRASTRUCTURE ONLY -- never imported by the product package.  `OracleBackend` and `OracleVecEnv` plug into
the product agents through their public constructor/explore arguments (agents accept any backend / env object)."""
import numpy as np
import torch

from . import binding as B


class OracleBackend:
    """GAE via the C oracle; no packed MLPs (plain torch forwards on CPU tensors)."""
    name = "oracle"

    def check_device(self, device):
        pass

    def gae(self, reward, mask, value, lam, use_gae):
        r, a = B.gae(reward.detach().cpu().numpy(), mask.detach().cpu().numpy(), value.detach().cpu().numpy(), lam, use_gae)
        return torch.from_numpy(r).to(reward.device), torch.from_numpy(a).to(reward.device)

    def packed(self, module):
        return None


class OracleVecEnv:
    """The vectorised-env duck type of pime_amd.vec_env on top of OraclePH / OracleWT (CPU tensors, Philox draws)."""
    action_dim = 1
    if_discrete = False

    def __init__(self, kind, num_envs, seed=0, env_offset=0, table=None, **kw):
        self._ctor = dict(kind=kind, num_envs=num_envs, seed=seed, env_offset=env_offset, table=table, **kw)
        self.reset_count = 0
        self.kind = kind
        if kind == "ph":
            self.table = B.ph_table() if table is None else table
            self.core = B.OraclePH(num_envs, self.table, seed=seed, env_offset=env_offset, **kw)
            self.K = np.array([-0.02, 0.02, 0.035])
            self.max_step = kw.get("max_steps", 50)
            self.n_integrator = 1
        else:
            self.core = B.OracleWT(num_envs, seed=seed, env_offset=env_offset, **kw)
            S = kw.get("num_stack", 0)
            self.K = np.array([0., 0.4, -0.4, 0.]) if S == 0 else np.concatenate([np.zeros(3 * S - 3), [0., 0.4, -0.4]])
            self.max_step = kw.get("max_steps", 200)
            if S == 0:
                self.n_integrator = 1
        self.num_envs = num_envs
        self.state_dim = self.obs_dim = self.core.obs_dim
        self.device = torch.device("cpu")
        self.target_return = 2 ** 16
        self._t = 0
        self._was_reset = False
        self._last_obs = None

    @property
    def fresh(self):
        return self._was_reset and self._t == 0

    def clone(self, **overrides):
        env = OracleVecEnv(**{**self._ctor, **overrides})
        for k in ("env_name", "target_return"):
            if hasattr(self, k):
                setattr(env, k, getattr(self, k))
        return env

    def reset(self, mask=None, out=None):
        obs = torch.from_numpy(self.core.reset(mask=mask))
        self._t, self._was_reset = 0, True
        self.reset_count += 1
        self._last_obs = obs
        if out is not None:
            out.copy_(obs)
            return out
        return obs

    def observe(self, out=None):
        if out is not None:
            out.copy_(self._last_obs)
            return out
        return self._last_obs

    def _finish(self, res, out_obs, out_reward, out_done):
        obs, _, rew, done = res
        self._t = 0 if done.all() else self._t + 1
        obs, rew, done = torch.from_numpy(obs), torch.from_numpy(rew.astype(np.float32)), torch.from_numpy(done.astype(np.uint8))
        self._last_obs = obs
        if out_obs is not None:
            out_obs.copy_(obs); obs = out_obs
        if out_reward is not None:
            out_reward.copy_(rew); rew = out_reward
        if out_done is not None:
            out_done.copy_(done); done = out_done
        return obs, rew, done

    def step(self, action, auto_reset=True, out_obs=None, out_reward=None, out_done=None):
        a = action.detach().reshape(-1).double().numpy()
        return self._finish(self.core.step(a, auto_reset=auto_reset), out_obs, out_reward, out_done)

    def step_residual(self, a_pre, obs_in, priorK=None, auto_reset=True, out_obs=None, out_reward=None, out_done=None):
        k = -self.K if priorK is None else np.asarray(priorK).reshape(-1)
        a = B.residual_action(a_pre.detach().reshape(-1).numpy(), obs_in.detach().numpy(), k)
        return self._finish(self.core.step(a, auto_reset=auto_reset), out_obs, out_reward, out_done)


class _TwinCore:
    """OraclePH / OracleWT's call shape on top of the CPU twin (oracle/twin.py: libpime_cpu.so)."""

    def __init__(self, tw):
        self.tw, self.obs_dim = tw, tw.obs_dim

    def reset(self, mask=None):
        return self.tw.reset(mask=mask)

    def step(self, a, auto_reset=True):
        obs, rew, done = self.tw.step(a, auto_reset=auto_reset)
        return obs, None, rew, done

    def get(self, name):
        return self.tw.get(name)


class TwinVecEnv(OracleVecEnv):
    """OracleVecEnv with the env arithmetic done by libpime_cpu.so -- the product's own lane functions compiled for the host
    (include/pime_cpu.h) -- instead of the C oracle: what bench.py's cpu_baseline times."""

    def __init__(self, kind, num_envs, seed=0, env_offset=0, table=None, threads=1, **kw):
        super().__init__(kind, num_envs, seed=seed, env_offset=env_offset, table=table, **kw)
        from .twin import TwinEnv
        over = {}
        if kind != "ph":
            import pime_amd.native as nt
            over = dict(reward_type=nt.REWARD[kw.get("reward_type", "square_distance")], max_steps=kw.get("max_steps", 200))
        self.twin = TwinEnv(kind, num_envs, table=self.table if kind == "ph" else None, seed=seed, env_offset=env_offset,
                            threads=threads, **over)
        self.core = _TwinCore(self.twin)

    def step_residual(self, a_pre, obs_in, priorK=None, auto_reset=True, out_obs=None, out_reward=None, out_done=None):
        k = -self.K if priorK is None else np.asarray(priorK).reshape(-1)
        obs, rew, done = self.twin.step_residual(a_pre.detach().reshape(-1).numpy(), obs_in.detach().numpy(), k, auto_reset=auto_reset)
        return self._finish((obs, None, rew, done), out_obs, out_reward, out_done)
