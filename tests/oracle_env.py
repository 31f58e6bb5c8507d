"""Test helper: a one-instance gym-style env on the CPU oracle with injected per-episode draws, used to exercise the
agents' one-instance rollout loop without a GPU.  Test infrastructure only."""
import numpy as np

import oracle


class OracleSinglePH:
    """Duck type of PreprocessEnv(TimeLimit(PH env)): reset() -> float32 obs, step(a) -> (obs32, reward, done, {})."""
    if_discrete = False
    state_dim, action_dim, max_step = 3, 1, 50
    n_integrator = 1
    K = np.array([-0.02, 0.02, 0.035])

    def __init__(self, table, draws):
        self.core = oracle.OraclePH(1, table)
        self.draws = list(draws)
        self.k = 0

    def reset(self):
        d = np.asarray(self.draws[self.k % len(self.draws)], dtype=np.float64)[None]
        self.k += 1
        return self.core.reset(draws=d)[0].astype(np.float32)

    def step(self, action):
        obs32, obs64, rew, done = self.core.step(np.asarray(action, dtype=np.float64).reshape(-1)[:1])
        return obs64[0].astype(np.float32), float(rew[0]), bool(done[0]), {}


class OracleSingleWT:
    if_discrete = False
    action_dim, max_step = 1, 200

    def __init__(self, draws, noise, num_stack=1, reward_type="distance"):
        self.core = oracle.OracleWT(1, reward_type=reward_type, num_stack=num_stack)
        self.state_dim = self.core.obs_dim
        self.K = np.concatenate([np.zeros(3 * num_stack - 3), [0., 0.4, -0.4]]) if num_stack else np.array([0., 0.4, -0.4, 0.])
        self.draws, self.noise = list(draws), np.asarray(noise)
        self.k = self.t = 0

    def reset(self):
        d = np.asarray(self.draws[self.k % len(self.draws)], dtype=np.float64)[None]
        self.k += 1
        return self.core.reset(draws=d)[0].astype(np.float32)

    def step(self, action):
        nz = self.noise[self.t % len(self.noise)][None]
        self.t += 1
        obs32, obs64, rew, done = self.core.step(np.asarray(action, dtype=np.float64).reshape(-1)[:1], noise=nz)
        return obs64[0].astype(np.float32), float(rew[0]), bool(done[0]), {}
