"""One-instance pH-neutralisation envs with the reference's class names and methods
(/root/reference/gym_control/envs/ph.py:351-478), computed by the HIP kernels (float64 state mode).

Random draws come from where the reference takes them: the ensemble parameters from the process-global
`np.random` (ph.py:410), x0 and the goal r from the env's own gym-seeded `np_random` (ph.py:420,424) -- so
`env.seed(s); np.random.seed(s)` reproduces the reference trajectory seed-for-seed."""
import numpy as np

from ...vec_env import VecPH
from ._facade import SingleEnvFacade, device_property


class PH1DChangingParamUniformGoalIntegrator(SingleEnvFacade):
    n_integrator = 1
    integral_max = 25.
    dim = 1
    _integral_bound = True
    _copy_fields = ("x", "I", "r", "qww_V", "qc_V", "t", "episode")

    def __init__(self, qww_V=(0.005, 0.015), qc_V=(0.0015, 0.0025), kw=1e-14, kchem=5.6e-10, ka=0.5e-5, MNaOH=0.01,
                 MHA=0.005, MNH3=0.01, MHCl=None, r=7.0, n_discrete=200, sample_t=20, reset_from_last_state=False,
                 reward_type="distance", distance_threshold=0.05, P_control_K=np.array([1, 1]),
                 P_control_L=np.array([-0.4]), action_punishment=0., action_change_punishment=0.,
                 max_episode_steps=200, seed=None, device="cuda"):
        if reset_from_last_state:
            raise NotImplementedError("reset_from_last_state=True is dead in every registered config (SURVEY.md A.2)")
        MHCl = np.arange(0., 0.2, step=0.00001) if MHCl is None else np.asarray(MHCl)
        self._ctor = dict(qww_V=tuple(qww_V), qc_V=tuple(qc_V), reward_type=reward_type, P_control_K=P_control_K,
                          MHCl_step=float(MHCl[1] - MHCl[0]), MHCl_len=len(MHCl), sample_t=float(sample_t),
                          chem=dict(kw=kw, kchem=kchem, ka=ka, MNaOH=MNaOH, MHA=MHA, MNH3=MNH3),
                          action_punishment=action_punishment, action_change_punishment=action_change_punishment,
                          distance_threshold=distance_threshold, device=device)
        self.qww_Vrange, self.qc_Vrange = tuple(qww_V), tuple(qc_V)
        self.reward_type = reward_type
        self.distance_threshold = distance_threshold
        self.max_episode_steps = max_episode_steps
        self.action_punishment, self.action_change_punishment = action_punishment, action_change_punishment
        self._integral_punish = 0.0
        self.K, self.L = P_control_K, P_control_L
        self.if_reset_all = True
        self.low, self.high = 0., 1.5
        self.MHCl = MHCl
        np.random.uniform(*self.qww_Vrange), np.random.uniform(*self.qc_Vrange)  # the constructor's own sample (ph.py:384)
        vec = self._make_vec()
        self.pH = vec.table
        self._finish_init(vec, -np.ones(3) * np.inf, np.ones(3) * np.inf, seed)

    def _make_vec(self):
        c = self._ctor
        # the env itself never reports done (ph.py:348); the 50-step limit is gym's TimeLimit around it
        return VecPH(1, device=c["device"], state_mode="f64", draws=self._draws(self._episode_draws),
                     reward_type=c["reward_type"], max_episode_steps=2 ** 30, integral_bound=self._integral_bound,
                     qww_V=c["qww_V"], qc_V=c["qc_V"], P_control_K=c["P_control_K"], MHCl_step=c["MHCl_step"],
                     MHCl_len=c["MHCl_len"], chem=c["chem"], action_punishment=c["action_punishment"],
                     action_change_punishment=c["action_change_punishment"], sample_t=c["sample_t"],
                     distance_threshold=c["distance_threshold"])

    def _clone_blank(self):
        clone = object.__new__(type(self))
        clone.__dict__.update({k: v for k, v in self.__dict__.items() if k not in ("_vec", "_device_action")})
        clone._finish_init(clone._make_vec(), -np.ones(3) * np.inf, np.ones(3) * np.inf, None)
        clone._vec.set_reset_all(self.if_reset_all)
        clone.integral_punish = self._integral_punish
        return clone

    # ---- draws: exactly the reference's call sites ---------------------------------------------------------
    def sample_parameters(self):
        return np.random.uniform(*self.qww_Vrange), np.random.uniform(*self.qc_Vrange)   # ph.py:409-410

    def _episode_draws(self):
        qww, qc = self.sample_parameters() if self.if_reset_all else (0.0, 0.0)           # reset_all vs reset_r
        x0 = self.np_random.uniform(low=0, high=50)                                        # ph.py:420
        r = self.np_random.uniform(3., 11.)                                                # ph.py:424
        return qww, qc, x0, r

    # ---- gym API -------------------------------------------------------------------------------------------
    def reset(self):
        self._vec.set_reset_all(self.if_reset_all)
        self._vec.reset()
        return self._get_observe()

    reset_all = reset_r = reset

    def step(self, action):
        # the kernel computes the reward in float64 (state mode f64) and stores it as float32 -- the value the
        # reference's buffer ends up holding after its own float32 cast (replay.py:272-276)
        rew, _ = self._step_device(action)
        return self._get_observe(), float(rew[0].item()), False, {}

    def _get_observe(self):
        return np.array([self.y, self.r, self.integrator])

    # ---- attributes / harness hooks ------------------------------------------------------------------------
    state = device_property("x")
    y = device_property("y")
    r = device_property("r")
    integrator = device_property("I")
    qww_V = device_property("qww_V")
    qc_V = device_property("qc_V")
    _episode_steps = device_property("t", int)

    @property
    def integral_punish(self):
        return self._integral_punish

    @integral_punish.setter
    def integral_punish(self, v):
        self._integral_punish = float(v)
        if self._vec is not None:
            from ... import native
            native.check(self._vec._lib.pime_env_set_punish(self._vec._h, float(v), self.action_punishment,
                                                            self.action_change_punishment))

    def set_reset_all(self, if_reset_all):
        self.if_reset_all = if_reset_all

    def set_state(self, state):
        self.state = state
        return self._get_observe()

    def set_r(self, r):
        self.r = r
        return self._get_observe()

    def set_params(self, qww_V, qc_V):
        """The reference only stores the two numbers and forgets update_system (ph.py:263-265), so its pH robust
        sweep never changes the plant.  Here the plant IS rebuilt; pass through
        `set_params_reference_quirk` to get the reference's no-op."""
        self._vec.set_params(qww_V, qc_V)

    def set_params_reference_quirk(self, qww_V, qc_V):
        pass

    def set_qww_V(self, v):
        self._vec.set_field("qww_V", v)

    def set_qc_V(self, v):
        self._vec.set_field("qc_V", v)

    def get_changable_parameters(self):
        return self.qww_V, self.qc_V

    def get_linear_action(self, state=None):
        s = self._get_observe() if state is None else state
        return -s @ np.asarray(self.K).T


class PH1DChangingParamUniformGoalIntegrator_NoBound(PH1DChangingParamUniformGoalIntegrator):
    _integral_bound = False
