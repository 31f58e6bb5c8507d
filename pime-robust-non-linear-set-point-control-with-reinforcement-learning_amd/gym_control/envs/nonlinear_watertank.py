"""One-instance two-tank envs with the reference's class names and methods
(/root/reference/gym_control/envs/nonlinear_watertank.py:828-939 Integrator, :942-1053 plain goal, :1056-1208
Stacking), computed by the HIP kernels (float64 state mode).  Every draw -- ensemble params, initial levels, goal
and the two per-step process-noise normals -- is taken from the process-global `np.random`, in the reference's
order (:891-893, :912-913, :810-811)."""
import numpy as np

from ...vec_env import VecWaterTank
from ._facade import SingleEnvFacade, device_property


class NonLinearWaterTankChangingParamUniformGoalIntegrator(SingleEnvFacade):
    n_integrator = 1
    integral_max = 25.
    _num_stack = 0
    _copy_fields = ("h1", "h2", "r", "I", "a1", "a2", "Kp", "t", "episode")

    def __init__(self, a1=(1, 2), a2=(1, 2), A1=2, A2=2, Kp=(1, 2), G=9.8, z1=1, z2=0.1, max_step=500, noise_scale=0.01,
                 gamma=0.99, seed=None, r=9.0, N=100, overflow_cost=-10, n_discrete=1, sample_t=0.02,
                 reward_type="distance", controller_type="P", distance_threshold=0.05, linearize_r=9.0,
                 reset_from_last_state=True, P_control_K=np.array([0., 0.4]), P_control_L=np.array([-0.4]),
                 P_max_action=10.0, num_stack=None, device="cuda"):
        if controller_type != "P":
            raise NotImplementedError("only controller_type='P' is used by the registered ids (SURVEY.md §2 row 2)")
        if reset_from_last_state:
            raise NotImplementedError("reset_from_last_state=True is dead in every registered config (SURVEY.md A.2)")
        if num_stack is not None:
            self._num_stack = num_stack
        self.num_stack = self._num_stack
        self._ctor = dict(a1=tuple(a1), a2=tuple(a2), Kp=tuple(Kp), A1=A1, A2=A2, G=G, z1=z1, max_step=max_step,
                          noise_scale=noise_scale, n_discrete=n_discrete, sample_t=sample_t, reward_type=reward_type,
                          distance_threshold=distance_threshold, P_control_K=P_control_K, P_max_action=P_max_action,
                          device=device)
        self.a1_range, self.a2_range, self.Kp_range = tuple(a1), tuple(a2), tuple(Kp)
        self.reward_type, self.noise_scale, self.gamma = reward_type, noise_scale, gamma
        self.distance_threshold = distance_threshold
        self._max_step = max_step
        self._integral_punish = 0.0
        self.if_reset_all = True
        self.K, self.L = P_control_K, None
        self.P_max_action = P_max_action
        self.sample_parameters()  # the constructor's own sample (:864)
        vec = self._make_vec()
        D = vec.obs_dim
        low, high = -np.ones(D) * 0, np.ones(D) * np.inf
        if self._num_stack == 0:
            low[-1], high[-1] = -self.integral_max, self.integral_max
        self.m = D if self._num_stack else 3
        self._finish_init(vec, low, high, seed)
        self.r = r

    def _make_vec(self):
        c = self._ctor
        return VecWaterTank(1, device=c["device"], state_mode="f64", draws=self._draws(self._episode_draws, self._noise_draws),
                            reward_type=c["reward_type"], max_step=c["max_step"], num_stack=self._num_stack, a1=c["a1"],
                            a2=c["a2"], Kp=c["Kp"], A1=c["A1"], A2=c["A2"], G=c["G"], sample_t=c["sample_t"],
                            n_discrete=c["n_discrete"], noise_scale=c["noise_scale"], z1=c["z1"],
                            P_max_action=c["P_max_action"], P_control_K=c["P_control_K"],
                            distance_threshold=c["distance_threshold"])

    def _clone_blank(self):
        clone = object.__new__(type(self))
        clone.__dict__.update({k: v for k, v in self.__dict__.items() if k not in ("_vec", "_device_action")})
        vec = clone._make_vec()
        clone._finish_init(vec, self.observation_space.low, self.observation_space.high, None)
        clone.max_step = self._max_step
        clone.integral_punish = self._integral_punish
        return clone

    # ---- draws ---------------------------------------------------------------------------------------------
    def sample_parameters(self):
        a1 = np.random.uniform(self.a1_range[0], self.a1_range[1])
        a2 = np.random.uniform(self.a2_range[0], self.a2_range[1])
        Kp = np.random.uniform(self.Kp_range[0], self.Kp_range[1])
        return a1, a2, Kp

    def _episode_draws(self):
        a1, a2, kp = self.sample_parameters() if self.if_reset_all else (0.0, 0.0, 0.0)
        h1, h2 = tuple(np.random.uniform(0., 10., 2))   # :912
        r = np.random.uniform(0., 10.)                   # :913
        return a1, a2, kp, h1, h2, r

    def get_noise(self):
        return np.random.normal(loc=0., scale=self.noise_scale)   # :271-272

    def _noise_draws(self):
        return self.get_noise(), self.get_noise()         # h1 then h2 (:810-811)

    # ---- gym API -------------------------------------------------------------------------------------------
    def reset(self):
        self._vec.set_reset_all(self.if_reset_all)
        self._vec.reset()
        return self._get_observe()

    reset_all = reset_r = reset

    def step(self, action):
        rew, done = self._step_device(action)
        return self._get_observe(), float(rew[0].item()), bool(done[0].item()), {}

    def _get_observe(self):
        if self._num_stack:
            return self._frames64()
        return np.array([self.h1, self.h2, self.r, self.integrator])

    def _frames64(self):
        # float32 observation rows are what the policy sees; the float64 newest frame is patched in from state
        obs = self._vec.observe().cpu().numpy()[0].astype(np.float64)
        obs[-3:] = [self.h1, self.h2, self.r]
        return obs

    @property
    def state(self):
        return self._get_observe()

    # ---- attributes / harness hooks ------------------------------------------------------------------------
    h1 = device_property("h1")
    h2 = device_property("h2")
    r = device_property("r")
    a1 = device_property("a1")
    a2 = device_property("a2")
    Kp = device_property("Kp")
    _episode_steps = device_property("t", int)

    @property
    def integrator(self):
        return float(self._vec.get_field("I")[0])

    @integrator.setter
    def integrator(self, v):
        self._vec.set_field("I", float(v))

    @property
    def max_step(self):
        return self._max_step

    @max_step.setter
    def max_step(self, n):
        self._max_step = int(n)
        if self._vec is not None:
            self._vec.set_max_step(int(n))

    @property
    def integral_punish(self):
        return self._integral_punish

    @integral_punish.setter
    def integral_punish(self, v):
        self._integral_punish = float(v)
        if self._vec is not None:
            from ... import native
            native.check(self._vec._lib.pime_env_set_punish(self._vec._h, float(v), 0.0, 0.0))

    def set_state(self, h1, h2):
        self.h1, self.h2 = h1, h2
        return self._get_observe()

    def set_r(self, r):
        self.r = r
        return self._get_observe()

    def set_reward_type(self, tp):
        raise NotImplementedError("reward_type is fixed at construction (gym.make(id, reward_type=...))")

    def get_changable_parameters(self):
        return self.a1, self.a2, self.Kp

    def reset_changable_parameters(self, a1, a2, Kp):
        self.a1, self.a2, self.Kp = a1, a2, Kp

    def get_P_action(self, state):
        state = np.asarray(state)[:len(self.K)]
        return np.clip(-state @ np.asarray(self.K).T, self.action_space.low, self.action_space.high)

    get_linear_action = get_P_action


class NonLinearWaterTankChangingParamUniformGoalStacking(NonLinearWaterTankChangingParamUniformGoalIntegrator):
    """Observation = the last num_stack frames [h1, h2, r], oldest first; no integrator (:1056-1208)."""
    n_integrator = 0
    _num_stack = 4
    _copy_fields = ("h1", "h2", "r", "a1", "a2", "Kp", "t", "episode")

    @property
    def integrator(self):
        raise AttributeError("the Stacking variant has no integrator")


class NonLinearWaterTankChangingParamUniformGoal(NonLinearWaterTankChangingParamUniformGoalStacking):
    """Plain goal-conditioned variant, observation [h1, h2, r] (:942-1053) == one stacked frame."""
    _num_stack = 1
