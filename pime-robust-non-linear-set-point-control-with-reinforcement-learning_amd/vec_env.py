"""Vectorised pH-process and water-tank envs: N independent instances advanced by one HIP launch per step.

This is the build's extension of the reference's one-instance envs (SURVEY.md §8b "vectorised extension"); the
per-instance semantics are the reference's (/root/reference/gym_control/envs/ph.py,
nonlinear_watertank.py) and live in csrc/env_kernels.hip.  Everything here is host-side plumbing: handle
ownership, tensor allocation, and the *draw sources* that decide where an episode's random numbers come from:

  PhiloxDraws      in-kernel Philox4x32-10 keyed by (seed, global env id, episode, slot)   -- throughput mode
  Mt19937Draws     per-env emulation of the reference's two MT19937 streams, env i seeded with base+i,
                   generated on the host and injected into the kernels                    -- seed-for-seed mode
  (the 1-instance facades in gym_control/ inject draws taken from the *global* np.random exactly where the
   reference takes them)

There is no CPU implementation: constructing an env without a gfx950 device raises `native.PimeError`.
"""
import ctypes as C

import numpy as np
import torch

from . import gym_compat, native

_STATE_MODES = {"f64": native.STATE_F64, "mixed": native.STATE_MIXED, "mixed16": native.STATE_MIXED16}


# ------------------------------------------------------------------------------------------------ draw sources
class PhiloxDraws:
    """Counter-based in-kernel draws; nothing to do on the host."""
    injects = False

    def reset_draws(self, env, lanes):
        return None

    def step_noise(self, env):
        return None

    def reseed(self, base):
        raise native.PimeError("Philox mode is keyed at construction: pass seed= to the env")


class Mt19937Draws:
    """Reference-compatible streams: env i behaves like the reference env after
    `env.seed(base+i); np.random.seed(base+i)` (train.py:105-106).

    pH  : ensemble params from the global stream (ph.py:410), x0 then r from gym's np_random (ph.py:420,424).
    WT  : everything from the global stream (nonlinear_watertank.py:891-893,912-913,271-272), including the two
          per-step normals, in call order.
    """
    injects = True

    def __init__(self, base_seed, num_envs, env_offset=0):
        self.num_envs = num_envs
        self.env_offset = env_offset
        self.reseed(base_seed)

    def reseed(self, base):
        ids = [int(base) + self.env_offset + i for i in range(self.num_envs)]
        self.global_rs = [np.random.RandomState(s) for s in ids]
        self.env_rs = [gym_compat.np_random(s)[0] for s in ids]

    def reset_draws(self, env, lanes):
        out = np.zeros((self.num_envs, env.draw_width), dtype=np.float64)
        for i in lanes:
            out[i] = env._draw_episode(self.global_rs[i], self.env_rs[i])
        return out

    def step_noise(self, env):
        if not env.has_step_noise:
            return None
        out = np.empty((self.num_envs, 2), dtype=np.float64)
        s = env.noise_scale
        for i, rs in enumerate(self.global_rs):
            out[i, 0] = rs.normal(loc=0., scale=s)  # get_noise(): nonlinear_watertank.py:271-272, h1 then h2
            out[i, 1] = rs.normal(loc=0., scale=s)
        return out


class CallbackDraws:
    """Draws supplied by callables (used by the 1-instance facades to read the process-global np.random)."""
    injects = True

    def __init__(self, episode_fn, noise_fn=None):
        self.episode_fn, self.noise_fn = episode_fn, noise_fn

    def reset_draws(self, env, lanes):
        out = np.zeros((env.num_envs, env.draw_width), dtype=np.float64)
        for i in lanes:
            out[i] = self.episode_fn(i)
        return out

    def step_noise(self, env):
        if not env.has_step_noise or self.noise_fn is None:
            return None
        return np.asarray([self.noise_fn(i) for i in range(env.num_envs)], dtype=np.float64).reshape(-1, 2)

    def reseed(self, base):
        pass


# ------------------------------------------------------------------------------------------------ base class
class VecControlEnv:
    """Common host logic of the two env families.  Tensors returned by reset/step live on `device`."""
    kind = None
    field_prefix = ""
    draw_width = 0
    has_step_noise = False
    action_dim = 1
    if_discrete = False

    def __init__(self, cfg, device, draws, K):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise native.PimeError(f"pime_amd envs run on a gfx950 GPU only (got device '{device}'); there is no "
                                   "CPU fallback")
        if self.device.index is None:  # "cuda" -> the concrete ordinal, so tensor.device comparisons are exact
            self.device = torch.device("cuda", torch.cuda.current_device())
        cfg.device_id = self.device.index
        self._lib = native.lib()
        with torch.cuda.device(self.device):
            self._h = C.c_void_p(self._lib.pime_env_create(C.byref(cfg)))
        if not self._h:
            raise native.PimeError(f"pime_env_create failed: {native.last_error()}")
        self.cfg = cfg
        self.num_envs = int(cfg.n_envs)
        self.state_dim = self.obs_dim = int(self._lib.pime_env_obs_dim(self._h))
        self.max_step = int(cfg.max_steps)
        self.draws = draws
        self.K = np.asarray(K, dtype=np.float64)
        self.action_max = 1.0
        self.target_return = 2 ** 16
        N, D = self.num_envs, self.obs_dim
        self.obs = torch.zeros((N, D), dtype=torch.float32, device=self.device)
        self.reward = torch.zeros((N,), dtype=torch.float32, device=self.device)
        self.done = torch.zeros((N,), dtype=torch.uint8, device=self.device)
        # Host mirror of the per-lane step counter (episodes are fixed length, so the host can tell which lanes end
        # without reading the device).  While every lane is at the same step -- the normal case -- it is ONE integer;
        # a masked reset or a write to the `t` field switches to a per-lane array.
        self._t_all = 0
        self._t_lanes = None
        self._was_reset = False
        self.reset_count = 0   # host-initiated resets so far: lets a caller that caches observations across calls (the
        #                        off-policy agents) notice that someone else -- the evaluator -- reset the env in between

    def clone(self, **overrides):
        """A second, independent env of the same configuration (own handle, own state slab): `copy.deepcopy` for an env whose
        state lives in HBM.  Used to give an off-policy run its own evaluation env (run.py: the evaluator resets the env it is
        given, which must not be the one whose running episodes the agent continues)."""
        env = type(self)(**{**self._ctor, **overrides})
        for k in ("env_name", "target_return"):
            if hasattr(self, k):
                setattr(env, k, getattr(self, k))
        return env

    # -- lifetime ------------------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            native.destroy_handle(self._lib.pime_env_destroy, self._h)   # parked, not freed, while a stream capture is open
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # interpreter shutdown
            pass

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _dev64(self, host):
        return None if host is None else torch.as_tensor(np.ascontiguousarray(host), dtype=torch.float64).to(
            self.device, non_blocking=False)

    # -- gym-like API --------------------------------------------------------------------------------------
    def seed(self, seed=None):
        """Env i uses seed+i.  Only meaningful for injected (MT19937) draws; Philox is keyed at construction."""
        if seed is not None and self.draws.injects:
            self.draws.reseed(seed)
        return [seed]

    def reset(self, mask=None, out=None):
        """Reset the lanes selected by `mask` (bool/uint8 [N], host or device; None = all).  Returns obs [N, D]."""
        obs = self.obs if out is None else out
        if mask is None:
            lanes, mask_dev = range(self.num_envs), None
            self._t_all, self._t_lanes = 0, None
        else:
            m = torch.as_tensor(mask).to(torch.uint8)
            lanes = np.nonzero(m.cpu().numpy())[0]
            mask_dev = m.to(self.device)
            self._lane_steps()[lanes] = 0
        draws = self._dev64(self.draws.reset_draws(self, lanes)) if self.draws.injects else None
        native.check(self._lib.pime_env_reset(self._h, native.ptr(mask_dev), native.ptr(draws), native.ptr(obs),
                                              self._stream()), "pime_env_reset")
        self._was_reset = True
        self.reset_count += 1
        return obs

    def _lane_steps(self):
        """Per-lane view of the step mirror (materialised on first need)."""
        if self._t_lanes is None:
            self._t_lanes = np.full(self.num_envs, self._t_all, dtype=np.int64)
        return self._t_lanes

    def _pre_step(self, auto_reset):
        noise = self._dev64(self.draws.step_noise(self)) if self.draws.injects else None
        if self._t_lanes is None:                      # lock-step fast path: O(1) host work per launch
            self._t_all += 1
            ending = range(self.num_envs) if self._t_all >= self.max_step else ()
            if auto_reset and len(ending):
                self._t_all = 0
        else:
            self._t_lanes += 1
            ending = np.nonzero(self._t_lanes >= self.max_step)[0]
            if auto_reset and len(ending):
                self._t_lanes[ending] = 0
                if not self._t_lanes.any():
                    self._t_all, self._t_lanes = 0, None   # back in lock-step
        reset_draws = None
        if auto_reset and len(ending) and self.draws.injects:
            reset_draws = self._dev64(self.draws.reset_draws(self, ending))
        return noise, reset_draws

    def step(self, action, auto_reset=True, out_obs=None, out_reward=None, out_done=None):
        """action: [N] or [N,1] tensor (float32 or float64) of env actions.  Returns (obs, reward, done)."""
        a = action.reshape(-1)
        if a.dtype not in (torch.float32, torch.float64):
            a = a.to(torch.float32)
        a = a.contiguous()
        assert a.numel() == self.num_envs and a.device == self.device, "action must be an [N] tensor on the env device"
        obs = self.obs if out_obs is None else out_obs
        rew = self.reward if out_reward is None else out_reward
        done = self.done if out_done is None else out_done
        noise, reset_draws = self._pre_step(auto_reset)
        native.check(self._lib.pime_env_step(self._h, native.ptr(a), native.F32 if a.dtype == torch.float32 else native.F64,
                                             native.ptr(noise), int(auto_reset), native.ptr(reset_draws), native.ptr(obs),
                                             native.ptr(rew), native.ptr(done), self._stream()), "pime_env_step")
        return obs, rew, done

    def step_residual(self, a_pre, obs_in, priorK=None, auto_reset=True, out_obs=None, out_reward=None, out_done=None):
        """env.step(tanh(a_pre) + obs_in @ priorK) with the composition fused into the kernel
        (elegantrl/agent_residual.py:61).  a_pre [N] float32, obs_in [N, D] float32 (may alias nothing written)."""
        a = a_pre.reshape(-1).to(torch.float32).contiguous()
        obs_in = obs_in.contiguous()
        assert obs_in.dtype == torch.float32 and tuple(obs_in.shape) == (self.num_envs, self.obs_dim)
        k = np.ascontiguousarray(-self.K if priorK is None else np.asarray(priorK, dtype=np.float64).reshape(-1))
        assert k.size == self.obs_dim
        obs = self.obs if out_obs is None else out_obs
        assert obs.data_ptr() != obs_in.data_ptr(), "obs_in and the output obs must not alias"
        rew = self.reward if out_reward is None else out_reward
        done = self.done if out_done is None else out_done
        noise, reset_draws = self._pre_step(auto_reset)
        native.check(self._lib.pime_env_step_residual(self._h, native.ptr(a), native.ptr(obs_in), native.ptr(k),
                                                      native.ptr(noise), int(auto_reset), native.ptr(reset_draws),
                                                      native.ptr(obs), native.ptr(rew), native.ptr(done), self._stream()),
                     "pime_env_step_residual")
        return obs, rew, done

    # -- binary16 observation / reward buffers (state_mode "mixed16": BASELINE.json config 5) -----------------------------
    def _half_buffers(self):
        if getattr(self, "_obs_h", None) is None:
            if self.cfg.state_mode != native.STATE_MIXED16:
                raise native.PimeError("binary16 observations need state_mode='mixed16'")
            N, D = self.num_envs, self.obs_dim
            self._obs_h = torch.zeros((N, D), dtype=torch.float16, device=self.device)
            self._reward_h = torch.zeros((N,), dtype=torch.float16, device=self.device)
        return self._obs_h, self._reward_h

    def reset_h(self, mask=None, out=None):
        """`reset` writing float16 observations (pime_env_reset_h)."""
        obs = self._half_buffers()[0] if out is None else out
        assert obs.dtype == torch.float16
        if mask is None:
            lanes, mask_dev = range(self.num_envs), None
            self._t_all, self._t_lanes = 0, None
        else:
            m = torch.as_tensor(mask).to(torch.uint8)
            lanes = np.nonzero(m.cpu().numpy())[0]
            mask_dev = m.to(self.device)
            self._lane_steps()[lanes] = 0
        draws = self._dev64(self.draws.reset_draws(self, lanes)) if self.draws.injects else None
        native.check(self._lib.pime_env_reset_h(self._h, native.ptr(mask_dev), native.ptr(draws), native.ptr(obs),
                                                self._stream()), "pime_env_reset_h")
        self._was_reset = True
        self.reset_count += 1
        return obs

    def step_h(self, action, auto_reset=True, out_obs=None, out_reward=None, out_done=None):
        """`step` with float16 observation / reward buffers (pime_env_step_h); action float32 [N]."""
        a = action.reshape(-1).to(torch.float32).contiguous()
        obs_h, rew_h = self._half_buffers()
        obs = obs_h if out_obs is None else out_obs
        rew = rew_h if out_reward is None else out_reward
        done = self.done if out_done is None else out_done
        noise, reset_draws = self._pre_step(auto_reset)
        native.check(self._lib.pime_env_step_h(self._h, native.ptr(a), native.ptr(noise), int(auto_reset), native.ptr(reset_draws),
                                               native.ptr(obs), native.ptr(rew), native.ptr(done), self._stream()),
                     "pime_env_step_h")
        return obs, rew, done

    def step_residual_h(self, a_pre, obs_in, priorK=None, auto_reset=True, out_obs=None, out_reward=None, out_done=None):
        """`step_residual` with float16 buffers: obs_in is the float16 observation the policy saw (pime_env_step_residual_h)."""
        a = a_pre.reshape(-1).to(torch.float32).contiguous()
        obs_in = obs_in.contiguous()
        assert obs_in.dtype == torch.float16 and tuple(obs_in.shape) == (self.num_envs, self.obs_dim)
        k = np.ascontiguousarray(-self.K if priorK is None else np.asarray(priorK, dtype=np.float64).reshape(-1))
        obs_h, rew_h = self._half_buffers()
        obs = obs_h if out_obs is None else out_obs
        assert obs.data_ptr() != obs_in.data_ptr(), "obs_in and the output obs must not alias"
        rew = rew_h if out_reward is None else out_reward
        done = self.done if out_done is None else out_done
        noise, reset_draws = self._pre_step(auto_reset)
        native.check(self._lib.pime_env_step_residual_h(self._h, native.ptr(a), native.ptr(obs_in), native.ptr(k),
                                                        native.ptr(noise), int(auto_reset), native.ptr(reset_draws),
                                                        native.ptr(obs), native.ptr(rew), native.ptr(done), self._stream()),
                     "pime_env_step_residual_h")
        return obs, rew, done

    @property
    def fresh(self):
        """True when every lane is at step 0 of an episode (just reset, or auto-reset by the last step)."""
        return self._was_reset and (self._t_all == 0 if self._t_lanes is None else not self._t_lanes.any())

    def observe(self, out=None):
        obs = self.obs if out is None else out
        native.check(self._lib.pime_env_observe(self._h, native.ptr(obs), self._stream()), "pime_env_observe")
        return obs

    @property
    def supports_fused_rollout(self):
        """The one-launch-per-episode rollout kernel needs mixed-precision state (float32 words, or binary16 storage: "mixed16")
        and in-kernel (Philox) draws; which observation / actor shapes it serves is the library's answer (`rollout_supported`)."""
        return self.cfg.state_mode in (native.STATE_MIXED, native.STATE_MIXED16) and not self.draws.injects

    @property
    def trajectory_dtype(self):
        """dtype of the observation / reward rows this env hands to a trajectory buffer: float16 in state_mode "mixed16"
        (BASELINE.json config 5 "fp16 state"), else float32."""
        return torch.float16 if self.cfg.state_mode == native.STATE_MIXED16 else torch.float32

    def rollout_supported(self, packed_actor):
        """Does the fused rollout kernel serve this env with this packed actor (kind / width)?  (pime_rollout_supported)"""
        kind = native.MLP_MODULAR_ACTOR if packed_actor.kind == "modular_actor" else native.MLP_PLAIN_ACTOR
        return self.supports_fused_rollout and bool(self._lib.pime_rollout_supported(self._h, kind, int(packed_actor.md)))

    def rollout(self, packed_actor, a_std_log, priorK, n_steps, noise_seed, noise_epoch, state, action, noise, reward, done):
        """Advance every lane `n_steps` steps under the packed residual policy in ONE launch (csrc/rollout.hip):
        state [n_steps+1, N, D] (slot 0 = current observation), action / noise / reward [n_steps, N], done uint8.
        Requires whole episodes from a fresh env (every lane at step 0), like AgentResidual*.explore_env collects."""
        assert self.fresh and n_steps % self.max_step == 0, "fused rollout advances whole episodes from a fresh env"
        for t_ in (state, action, noise, reward, done):
            assert t_.is_contiguous() and t_.device == self.device
        k = np.ascontiguousarray(np.asarray(priorK, dtype=np.float64).reshape(-1))
        assert k.size == self.obs_dim and packed_actor.D == self.obs_dim
        assert state.dtype == reward.dtype == self.trajectory_dtype and action.dtype == noise.dtype == torch.float32
        fn = self._lib.pime_rollout_h if state.dtype == torch.float16 else self._lib.pime_rollout
        native.check(fn(
            self._h, native.MLP_MODULAR_ACTOR if packed_actor.kind == "modular_actor" else native.MLP_PLAIN_ACTOR,
            packed_actor.md, native.ptr(packed_actor.packed), native.ptr(a_std_log), native.ptr(k), int(n_steps),
            C.c_uint64(noise_seed), C.c_uint32(noise_epoch), native.ptr(state), native.ptr(action), native.ptr(noise),
            native.ptr(reward), native.ptr(done), self._stream()), "pime_rollout")
        # whole episodes with in-kernel auto-reset: every lane is back at step 0 of a new episode

    def offpolicy_rollout_supported(self, packed_actor):
        """Does the fused off-policy exploration kernel serve this env with this packed deterministic actor?"""
        return (self.cfg.state_mode == native.STATE_MIXED and not self.draws.injects and packed_actor.kind == "critic"
                and packed_actor.D == self.obs_dim and bool(self._lib.pime_rollout_offpolicy_supported(self._h, int(packed_actor.md))))

    def rollout_offpolicy(self, packed_actor, priorK, explore_noise, gamma, reward_scale, n_steps, noise_seed, noise_epoch, obs,
                          ring_state, ring_other, slot0):
        """`n_steps` lock-steps of every lane under the deterministic actor + clipped exploration noise in ONE launch, the
        transitions written into the device ring from slot `slot0` on (csrc/rollout_offpolicy.hip); `obs` [N, D] is read (the
        lanes' current observation) and overwritten with the observation after the last step.  The running episodes continue."""
        for t_ in (obs, ring_state, ring_other):
            assert t_.is_contiguous() and t_.device == self.device and t_.dtype == torch.float32
        slots = ring_state.shape[0]
        assert ring_state.shape == (slots, self.num_envs, self.obs_dim) and ring_other.shape == (slots, self.num_envs, 3)
        k = np.ascontiguousarray(np.asarray(priorK, dtype=np.float64).reshape(-1))
        assert k.size == self.obs_dim and self._was_reset
        native.check(self._lib.pime_rollout_offpolicy(
            self._h, int(packed_actor.md), native.ptr(packed_actor.packed), native.ptr(k), C.c_float(explore_noise), C.c_float(gamma),
            C.c_float(reward_scale), int(n_steps), C.c_uint64(noise_seed), C.c_uint32(noise_epoch), native.ptr(obs),
            native.ptr(ring_state), native.ptr(ring_other), int(slot0), int(slots), self._stream()), "pime_rollout_offpolicy")
        if self._t_lanes is None:     # host mirror of the step counters: auto-reset wraps them at max_step
            self._t_all = (self._t_all + n_steps) % self.max_step
        else:
            self._t_lanes = (self._t_lanes + n_steps) % self.max_step

    def eval_supported(self, packed_actor=None, trace=False, schedule=False):
        """Does the fused evaluation kernel serve this env (with this packed actor, or the prior controller alone)?
        (pime_rollout_eval_supported: 1 = everything, 2 = returns and trace but no set-point schedule -- a Stacking observation
        at width 256; float64 or mixed state, in-kernel draws).  `schedule`: the caller wants a set-point schedule."""
        if self.draws.injects:
            return False
        if packed_actor is None:
            level = self._lib.pime_rollout_eval_supported(self._h, -1, 0)
        else:
            if packed_actor.kind not in ("modular_actor", "plain_actor") or packed_actor.D != self.obs_dim:
                return False
            kind = native.MLP_MODULAR_ACTOR if packed_actor.kind == "modular_actor" else native.MLP_PLAIN_ACTOR
            level = self._lib.pime_rollout_eval_supported(self._h, kind, int(packed_actor.md))
        return level == 1 or (level == 2 and not schedule)

    def rollout_eval(self, packed_actor, priorK, n_steps, setpoints=None, seg_len=0, want_trace=False, ret=None):
        """`n_steps` steps of every lane under the deterministic residual policy in ONE launch (csrc/rollout_eval.hip; no
        exploration noise, no auto-reset): returns (ret float64 [N] = per-lane sum of rewards, accumulated into `ret` if given,
        trace float64 [n_steps, 6, N] or None).  packed_actor None = the prior controller alone.  setpoints / seg_len: a
        step-response schedule (see include/pime_hip.h).  The lanes are left mid-episode: reset before the next rollout."""
        k = np.ascontiguousarray(np.asarray(priorK, dtype=np.float64).reshape(-1))
        assert k.size == self.obs_dim
        if ret is None:
            ret = torch.zeros(self.num_envs, dtype=torch.float64, device=self.device)
        trace = torch.empty((n_steps, 6, self.num_envs), dtype=torch.float64, device=self.device) if want_trace else None
        sp = np.ascontiguousarray(np.asarray(setpoints if setpoints is not None else [], dtype=np.float64))
        if packed_actor is None:
            kind, md, img = -1, 0, None
        else:
            kind = native.MLP_MODULAR_ACTOR if packed_actor.kind == "modular_actor" else native.MLP_PLAIN_ACTOR
            md, img = int(packed_actor.md), packed_actor.packed
        native.check(self._lib.pime_rollout_eval(self._h, kind, md, native.ptr(img), native.ptr(k), int(n_steps), int(seg_len),
                                                 native.ptr(sp) if sp.size else None, int(sp.size), native.ptr(ret),
                                                 native.ptr(trace), self._stream()), "pime_rollout_eval")
        self._was_reset = False   # the lanes sit somewhere inside an episode: the next rollout must reset first
        if self._t_lanes is None:
            self._t_all += n_steps
        else:
            self._t_lanes += n_steps
        return ret, trace

    # -- state access (float64 numpy on the host; synchronous) -------------------------------------------------
    def get_field(self, name):
        out = np.empty(self.num_envs, dtype=np.float64)
        native.check(self._lib.pime_env_read_field(self._h, native.FIELD[self.field_prefix + name], native.ptr(out),
                                                   self._stream()), f"read_field({name})")
        return out

    def set_field(self, name, values, mask=None):
        v = np.ascontiguousarray(np.broadcast_to(np.asarray(values, dtype=np.float64), (self.num_envs,)))
        m = None if mask is None else np.ascontiguousarray(np.asarray(mask, dtype=np.uint8))
        native.check(self._lib.pime_env_write_field(self._h, native.FIELD[self.field_prefix + name], native.ptr(v),
                                                    native.ptr(m), self._stream()), f"write_field({name})")
        if name == "t":
            t = self._lane_steps()
            t[:] = v.astype(np.int64) if mask is None else np.where(m.astype(bool), v, t)

    def set_reset_all(self, if_reset_all, every=1):
        """if_reset_all False keeps the ensemble params across resets (ph.py:111-112; attribute
        nonlinear_watertank.py:861).  `every` = n of "resample every n episodes" (README step 4)."""
        self.if_reset_all = bool(if_reset_all)
        native.check(self._lib.pime_env_set_resample_every(self._h, int(every) if if_reset_all else 0))

    def set_max_step(self, n):
        self.max_step = int(n)
        native.check(self._lib.pime_env_set_max_steps(self._h, int(n)))

    def get_linear_action(self, state):
        """Prior controller -state @ K (ph.py:227-231; nonlinear_watertank.py:755-759 clips to the action box)."""
        k = torch.as_tensor(self.K, dtype=state.dtype, device=state.device)
        return -(state @ k)


# ------------------------------------------------------------------------------------------------ pH
_PH_TABLE_CACHE = {}


def ph_table(n=100000, step=1e-5, chem=None):
    """Titration LUT (cached per configuration); built by the native library in ~10 ms (reference: 13 s)."""
    key = (n, step, None if chem is None else tuple(sorted(chem.items())))
    if key not in _PH_TABLE_CACHE:
        _PH_TABLE_CACHE[key] = native.ph_table_build(n, step, chem)
    return _PH_TABLE_CACHE[key]


class VecPH(VecControlEnv):
    """N x PH1DChangingParamUniformGoalIntegrator[_NoBound] behind gym's TimeLimit (registered id ...-v35)."""
    kind = native.ENV_PH
    field_prefix = "ph_"
    draw_width = 4
    n_integrator = 1
    dim = 1

    def __init__(self, num_envs, device="cuda", state_mode="mixed", seed=0, env_offset=0, draws="philox",
                 reward_type="square_distance", max_episode_steps=50, integral_bound=True, resample_every=1,
                 qww_V=(0.005, 0.015), qc_V=(0.0015, 0.0025), P_control_K=(-0.02, 0.02, 0.035),
                 MHCl_step=1e-5, MHCl_len=100000, chem=None, action_punishment=0., action_change_punishment=0.,
                 integral_punish=0., sample_t=20.0, distance_threshold=0.05):
        self._ctor = {k: v for k, v in locals().items() if k not in ("self", "__class__")}
        cfg = native.EnvCfg()
        native.check(native.lib().pime_env_cfg_default(native.ENV_PH, C.byref(cfg)))
        self.table = ph_table(MHCl_len, MHCl_step, chem)
        cfg.n_envs = num_envs
        cfg.state_mode = _STATE_MODES[state_mode]
        cfg.max_steps = max_episode_steps
        cfg.reward_type = native.REWARD[reward_type]
        cfg.integral_bound = int(bool(integral_bound))
        cfg.resample_every = resample_every
        cfg.env_offset = env_offset
        cfg.seed = seed
        cfg.integral_punish, cfg.action_punish, cfg.action_change_punish = integral_punish, action_punishment, \
            action_change_punishment
        cfg.distance_threshold = distance_threshold
        cfg.range_lo[0], cfg.range_hi[0] = qww_V
        cfg.range_lo[1], cfg.range_hi[1] = qc_V
        cfg.ph_sample_t = sample_t
        cfg.ph_table_scale = 1.0 / MHCl_step
        cfg.ph_table = self.table.ctypes.data_as(C.POINTER(C.c_double))
        cfg.ph_table_len = len(self.table)
        self.qww_Vrange, self.qc_Vrange = tuple(qww_V), tuple(qc_V)
        self.reward_type = reward_type
        self.if_reset_all = resample_every > 0
        if draws == "philox":
            draws = PhiloxDraws()
        elif draws == "mt19937":
            draws = Mt19937Draws(seed, num_envs, env_offset)
        super().__init__(cfg, device, draws, P_control_K)

    def _draw_episode(self, global_rs, env_rs):
        # reset_all: sample_parameters (2 global uniforms, ph.py:410,413) then x0, r from np_random (:420,:424).
        # With if_reset_all False the reference does not touch the global stream (reset_r, :428-439).
        if self.if_reset_all:
            qww, qc = global_rs.uniform(*self.qww_Vrange), global_rs.uniform(*self.qc_Vrange)
        else:
            qww = qc = 0.0
        x0 = env_rs.uniform(low=0, high=50)
        r = env_rs.uniform(3., 11.)
        return qww, qc, x0, r

    def get_changable_parameters(self):
        return self.get_field("qww_V"), self.get_field("qc_V")

    def set_params(self, qww_V, qc_V, mask=None):
        """Unlike the reference's set_params (ph.py:263-265, which forgets update_system -- SURVEY.md App. C.3),
        this rebuilds the discretised plant."""
        self.set_field("qww_V", qww_V, mask)
        self.set_field("qc_V", qc_V, mask)

    reset_changable_parameters = set_params


# ------------------------------------------------------------------------------------------------ water tank
class VecWaterTank(VecControlEnv):
    """N x NonLinearWaterTankChangingParamUniformGoalIntegrator (num_stack=0) or ...GoalStacking (num_stack=S)."""
    kind = native.ENV_WT
    field_prefix = "wt_"
    draw_width = 6
    has_step_noise = True

    def __init__(self, num_envs, device="cuda", state_mode="mixed", seed=0, env_offset=0, draws="philox",
                 reward_type="square_distance", max_step=200, num_stack=0, resample_every=1,
                 a1=(0.0015, 0.0024), a2=(0.0015, 0.0024), Kp=(0.07, 0.17), A1=1, A2=1, G=980, sample_t=2, n_discrete=20,
                 noise_scale=0.01, z1=1, P_max_action=10.0, P_control_K=None, integral_punish=0.,
                 distance_threshold=0.05):
        self._ctor = {k: v for k, v in locals().items() if k not in ("self", "__class__")}
        cfg = native.EnvCfg()
        native.check(native.lib().pime_env_cfg_default(native.ENV_WT, C.byref(cfg)))
        cfg.n_envs = num_envs
        cfg.state_mode = _STATE_MODES[state_mode]
        cfg.max_steps = max_step
        cfg.reward_type = native.REWARD[reward_type]
        cfg.num_stack = num_stack
        cfg.resample_every = resample_every
        cfg.env_offset = env_offset
        cfg.seed = seed
        cfg.integral_punish = integral_punish
        cfg.distance_threshold = distance_threshold
        for j, rng in enumerate((a1, a2, Kp)):
            cfg.range_lo[j], cfg.range_hi[j] = rng
        cfg.wt_A1, cfg.wt_A2, cfg.wt_G = A1, A2, G
        cfg.wt_n_discrete = n_discrete
        cfg.wt_dt = sample_t / n_discrete
        cfg.wt_noise_scale = noise_scale
        cfg.wt_z1 = z1
        cfg.wt_pmax = P_max_action
        self.a1_range, self.a2_range, self.Kp_range = tuple(a1), tuple(a2), tuple(Kp)
        self.noise_scale = noise_scale
        self.num_stack = num_stack
        self.reward_type = reward_type
        self.if_reset_all = resample_every > 0
        if num_stack == 0:
            self.n_integrator = 1
        if P_control_K is None:
            if num_stack == 0:
                P_control_K = [0., 0.4, -0.4, 0.]            # gym_control/__init__.py:67
            else:
                P_control_K = np.zeros(3 * num_stack)       # :71-73
                P_control_K[-3:] = [0., 0.4, -0.4]
        if draws == "philox":
            draws = PhiloxDraws()
        elif draws == "mt19937":
            draws = Mt19937Draws(seed, num_envs, env_offset)
        super().__init__(cfg, device, draws, P_control_K)

    def _draw_episode(self, global_rs, env_rs):
        # reset_all: sample_parameters (:890-894) then uniform(0,10,2) and uniform(0,10) (:912-913), one stream
        if self.if_reset_all:
            a1 = global_rs.uniform(self.a1_range[0], self.a1_range[1])
            a2 = global_rs.uniform(self.a2_range[0], self.a2_range[1])
            kp = global_rs.uniform(self.Kp_range[0], self.Kp_range[1])
        else:
            a1 = a2 = kp = 0.0
        h1, h2 = tuple(global_rs.uniform(0., 10., 2))
        r = global_rs.uniform(0., 10.)
        return a1, a2, kp, h1, h2, r

    def get_changable_parameters(self):
        return self.get_field("a1"), self.get_field("a2"), self.get_field("Kp")

    def reset_changable_parameters(self, a1, a2, Kp, mask=None):
        self.set_field("a1", a1, mask)
        self.set_field("a2", a2, mask)
        self.set_field("Kp", Kp, mask)
