"""Base agents: AgentBase, AgentPPO (on-policy, the residual agents' parent) and AgentTD3
(interface of /root/reference/elegantrl/agent.py: AgentBase :15-124, AgentTD3 :276-394, AgentPPO :543-712).

What differs from the reference, by design:
  * `explore_env` understands vectorised envs (`env.num_envs`): all N lanes advance in lock-step, the policy
    runs once per step on [N, D], and transitions go straight into a time-major `TrajectoryBuffer` in HBM.
    A one-instance env (the gym-style facade) still takes the reference's per-step loop.
  * the value pass and the rollout policy mean run on the fused f32-MFMA forward, GAE on the scan kernel
    (via the injected backend; the product backend is HIP-only).
  * loss scalars are accumulated on the device and read back once per `update_net`, not four `.item()` syncs
    per minibatch (agent.py:644-653).
  * under torch.distributed every optimizer step all-reduces ONE flat gradient buffer, and the buffer-global
    advantage normalisation (agent.py:707) all-reduces three moments.
"""
import os

import numpy as np
import torch

from . import logger
from .. import dist as pdist
from ..backend import HipBackend
from .net import Actor, ActorPPO, CriticAdv, CriticTwin
from .replay import TrajectoryBuffer, VecReplayBuffer


def _no_gc():
    """Every HIP-graph capture of the agents runs inside native.capture_guard: no cyclic garbage collection while the capture is
    open, and the library parks -- instead of hipFree-ing -- any device memory a handle releases meanwhile (a hipFree under stream
    capture aborts the process: seen once in tests/test_gpu_td3.py, "Garbage-collecting" in the fatal error's stack, when a cyclic
    collection finalised an earlier env handle inside torch.cuda.graph).  The guard also covers refcount-driven finalisation,
    which disabling the collector alone does not."""
    from .. import native
    return native.capture_guard()


class AgentBase:
    def __init__(self, backend=None, device=None):
        self.learning_rate = 1e-4
        self.soft_update_tau = 2 ** -8
        self.state = None
        self.device = torch.device(device) if device is not None else None
        self.backend = backend if backend is not None else HipBackend()
        self.act = self.act_target = None
        self.cri = self.cri_target = None
        self.act_optimizer = self.cri_optimizer = None
        self.criterion = None
        self.get_obj_critic = None
        self.if_on_policy = False
        self._n_updates = 0
        self.dp = None          # pime_amd.dist.DataParallel when training sharded
        self.index_hook = None  # tests: callable(step, buf_len, batch_size) -> LongTensor of minibatch indices
        self.index_table_hook = None  # tests: callable(n_steps, buf_len, batch_size) -> LongTensor [n_steps, batch_size], the
        #                               whole update's minibatches at once (keeps the one-graph-per-step path, unlike index_hook)

    def _pick_device(self):
        if self.device is None:
            self.device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
        if self.device.type == "cuda" and self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.backend.check_device(self.device)
        return self.device

    def select_action(self, state):
        states = torch.as_tensor(np.asarray(state)[None], dtype=torch.float32, device=self.device)
        with torch.no_grad():
            return self.act(states)[0].cpu().numpy()

    def explore_env(self, env, buffer, target_step, reward_scale, gamma):
        """Off-policy default: `target_step` transitions continuing from self.state (agent.py:54-70)."""
        for _ in range(target_step):
            action = self.select_action(self.state)
            next_s, reward, done, _ = env.step(action)
            buffer.append_buffer(self.state, (reward * reward_scale, 0.0 if done else gamma, *action))
            self.state = env.reset() if done else next_s
        return target_step

    def update_net(self, buffer, target_step, batch_size, repeat_times):
        raise NotImplementedError

    def save_load_model(self, cwd, if_save):
        """actor.pth / critic.pth state_dicts, the reference's checkpoint layout (agent.py:86-114).  Loading uses
        weights_only=True: nothing in the file is executed."""
        paths = {"act": os.path.join(cwd, "actor.pth"), "cri": os.path.join(cwd, "critic.pth")}
        for name, path in paths.items():
            net = getattr(self, name)
            if net is None:
                continue
            if if_save:
                torch.save(net.state_dict(), path)
            elif os.path.exists(path):
                net.load_state_dict(torch.load(path, map_location="cpu", weights_only=True))
                print(f"Loaded {name}:", cwd)
            else:
                print(f"FileNotFound when load {name}: {cwd}")
        if not if_save:
            self.weights_changed()

    def weights_changed(self):
        """Invalidate packed (kernel-layout) copies of the weights."""
        fused = getattr(self, "_packed", {}).get("fused")
        self._packed = {}
        if fused and fused.params_are(self):   # (False = "no fused kernel for these nets", remembered below)
            fused.repack()
            self._packed["fused"] = fused

    @staticmethod
    def soft_update(target_net, current_net, tau):
        """target <- (1 - tau) target + tau current (agent.py:116-124), as two multi-tensor launches per net on the GPU (the
        per-parameter loop is ~40 small launches per delayed update of a TD3 agent)."""
        with torch.no_grad():
            tar, cur = list(target_net.parameters()), list(current_net.parameters())
            if tar and tar[0].is_cuda:
                torch._foreach_mul_(tar, 1 - tau)
                torch._foreach_add_(tar, cur, alpha=tau)
                return
            for t, c in zip(tar, cur):
                t.mul_(1 - tau).add_(c, alpha=tau)


# ================================================================================================= PPO
class AgentPPO(AgentBase):
    def __init__(self, backend=None, device=None):
        super().__init__(backend, device)
        self.ratio_clip = 0.2
        self.lambda_entropy = 0.02
        self.lambda_gae_adv = 0.97
        self.if_use_gae = True
        self.if_on_policy = True
        self.if_use_dn = False
        self.noise = None
        self.optimizer = None
        self.compute_reward = None
        self._packed = {}
        self.noise_hook = None  # tests: callable(t, shape) -> exploration noise tensor (else torch.randn)
        self.use_fused_update = True
        self.use_hip_graphs = True
        self.use_single_graph = True     # one graph per optimizer step when nothing has to happen between its launches
        self.use_graph_collective = True  # data parallel: capture the RCCL all-reduce INSIDE that one graph
        self.use_update_graph = True      # ... and, once that graph exists, all n_steps optimizer steps of an update as ONE graph
        self.use_fused_rollout = True
        self.launch_timer = None  # optional callable(name, thunk) that brackets the thunk with HIP events

    # ---- construction ------------------------------------------------------------------------------------
    def _build_nets(self, net_dim, state_dim, action_dim):
        self.cri = CriticAdv(state_dim, net_dim, self.if_use_dn).to(self.device)
        self.act = ActorPPO(net_dim, state_dim, action_dim, self.if_use_dn).to(self.device)

    def init(self, net_dim, state_dim, action_dim, if_per=False):
        assert if_per is False, "on-policy agents do not use prioritised replay"
        self._pick_device()
        self.compute_reward = self.compute_reward_gae if self.if_use_gae else self.compute_reward_adv
        self._build_nets(net_dim, state_dim, action_dim)
        self._make_optimizer()
        self.criterion = torch.nn.SmoothL1Loss()

    def _make_optimizer(self):
        # ONE Adam over both nets (agent.py:565-566); rebuilt whenever the reference rebuilds it
        fused = self._packed.get("fused")
        if fused and fused.params_are(self):
            # the parameters already live in the fused path's flat tensor: a fresh optimizer there (fresh moments and step
            # count, as a rebuilt torch Adam has), and the captured graphs -- which replay the OLD optimizer's buffers -- go
            self.optimizer = fused.make_optimizer(self.learning_rate)
            fused.static = None
            self.weights_changed()
            return
        groups = [{"params": self.act.parameters(), "lr": self.learning_rate},
                  {"params": self.cri.parameters(), "lr": self.learning_rate}]
        # fused=True: one multi-tensor kernel per step on the GPU instead of ~10 foreach launches
        # capturable=True: the step counter lives on the device, so the step can be replayed from a HIP graph
        self.optimizer = torch.optim.Adam(groups, fused=True, capturable=True) if self.device.type == "cuda" \
            else torch.optim.Adam(groups)
        self.weights_changed()

    def init_actor_zero(self):
        """Zero the policy's output layer so the initial policy is the prior controller alone (agent.py:569-574)."""
        with torch.no_grad():
            self.act.net[-1].bias.fill_(0.)
            self.act.net[-1].weight.fill_(0.)
        self._make_optimizer()

    def frozen_transfer(self):
        self.cri.frozen_transfer()
        self.act.frozen_transfer()

    # ---- acting ------------------------------------------------------------------------------------------
    def select_action(self, state, if_deterministic=False):
        states = torch.as_tensor(np.asarray(state)[None], dtype=torch.float32, device=self.device)
        with torch.no_grad():
            if if_deterministic:
                return self.act(states)[0].cpu().numpy(), None
            actions, noises = self.act.get_action_noise(states)
        return actions[0].cpu().numpy(), noises[0].cpu().numpy()

    def _env_action(self, state, action):
        """What is sent to a one-instance env for the sampled pre-tanh `action` (agent.py:599)."""
        return np.tanh(action)

    def _packed_for(self, name):
        if name not in self._packed:
            self._packed[name] = self.backend.packed(getattr(self, name))
        return self._packed[name]

    def policy_mean(self, states):
        """a_avg for a [M, D] batch without autograd: fused MFMA forward when the shape is supported."""
        pk = self._packed_for("act")
        if pk is not None:
            return pk(states).unsqueeze(1)
        with torch.no_grad():
            return self.act.mean(states)

    def state_value(self, states):
        pk = self._packed_for("cri")
        if pk is not None:
            return pk(states)
        with torch.no_grad():
            out = [self.cri(states[i:i + 2 ** 16])[:, 0] for i in range(0, states.shape[0], 2 ** 16)]
        return torch.cat(out)

    def explore_env(self, env, buffer, target_step, reward_scale, gamma):
        if hasattr(env, "num_envs"):
            return self.explore_vec_env(env, buffer, target_step, reward_scale, gamma)
        # one-instance env: whole episodes until >= target_step transitions (agent.py:591-609)
        buffer.empty_buffer_before_explore()
        actual_step = 0
        while actual_step < target_step:
            state = env.reset()
            for _ in range(env.max_step):
                action, noise = self.select_action(state)
                next_state, reward, done, _ = env.step(self._env_action(state, action))
                actual_step += 1
                buffer.append_buffer(state, (reward * reward_scale, 0.0 if done else gamma, *action, *noise))
                if done:
                    break
                state = next_state
        return actual_step

    def _rollout_priorK(self):
        """Prior-controller gain of the fused rollout: none for plain PPO (the env sees tanh(a_pre), agent.py:599)."""
        return np.zeros(self.act.state_dim)

    def _fused_rollout_ok(self, env):
        if not (self.use_fused_rollout and getattr(env, "supports_fused_rollout", False)) or self.noise_hook is not None:
            return False
        if not hasattr(self, "_rollout_seed"):
            self._rollout_seed = int(torch.initial_seed()) & (2 ** 63 - 1)   # exploration stream follows torch's seed
            self._rollout_epoch = 0
        pk = self._packed_for("act")
        return pk is not None and self.act.state_dim == env.obs_dim and env.rollout_supported(pk)

    def fused_eval_policy(self, env):
        """(packed actor, priorK) if the fused evaluation kernel can run this agent's deterministic policy on `env`
        (run.py:600-619 as one launch, csrc/rollout_eval.hip), else None -> the evaluator steps the env launch by launch."""
        if not self.use_fused_rollout or not hasattr(env, "eval_supported"):
            return None
        pk = self._packed_for("act")
        if pk is None or not env.eval_supported(pk):
            return None
        return pk, self._rollout_priorK()

    def _vec_env_step(self, env, a_pre, obs, out_obs, out_reward, out_done):
        """Plain PPO: the env sees tanh(a_pre) (agent.py:599)."""
        step = env.step_h if out_obs.dtype == torch.float16 else env.step
        return step(torch.tanh(a_pre), auto_reset=True, out_obs=out_obs, out_reward=out_reward, out_done=out_done)

    def explore_vec_env(self, env, buffer, target_step, reward_scale, gamma):
        """Lock-step rollout of all lanes for whole episodes until >= target_step transitions are stored.
        Every tensor stays in HBM; per step: policy mean (fused forward) + noise + ONE env launch that also
        applies tanh + prior and writes obs/reward/done into the trajectory slots."""
        assert isinstance(buffer, TrajectoryBuffer) and buffer.num_envs == env.num_envs
        buffer.empty_buffer_before_explore()
        N, T_max = env.num_envs, buffer.horizon
        episodes = max(1, -(-target_step // (N * env.max_step)))
        assert episodes * env.max_step <= T_max, "TrajectoryBuffer horizon too short for target_step"
        std = None
        t = 0
        fused = self._fused_rollout_ok(env)
        # Every lane sits at the start of an episode either because nothing ran yet (-> reset) or because the last
        # step of the previous rollout auto-reset it inside the kernel (-> just read the observation back).
        half = buffer.state.dtype == torch.float16   # env in state_mode "mixed16": binary16 observation / reward rows
        assert buffer.state.dtype == getattr(env, "trajectory_dtype", torch.float32), "buffer / env row dtype mismatch"
        if half:
            if env.fresh:
                buffer.state[0].copy_(env.observe())   # float32 -> binary16: the rounding the *_h kernels apply
            else:
                env.reset_h(out=buffer.state[0])
        elif env.fresh:
            env.observe(out=buffer.state[0])
        else:
            env.reset(out=buffer.state[0])
        for ep in range(episodes):
            if fused:  # one launch per episode: policy forward + noise + env step + buffer writes (csrc/rollout.hip)
                n = env.max_step
                self._rollout_epoch += 1
                env.rollout(self._packed_for("act"), self.act.a_std_log.detach(), self._rollout_priorK(), n,
                            self._rollout_seed, self._rollout_epoch, buffer.state[t:t + n + 1], buffer.action[t:t + n],
                            buffer.noise[t:t + n], buffer.reward[t:t + n], buffer.done[t:t + n])
                t += n
                continue
            for _ in range(env.max_step):
                obs = buffer.state[t]
                with torch.no_grad():
                    if std is None:
                        std = self.act.a_std_log.detach().exp()
                    a_avg = self.policy_mean(obs.float() if half else obs)
                    noise = torch.randn_like(a_avg) if self.noise_hook is None else self.noise_hook(t, a_avg.shape)
                    a_pre = a_avg + noise * std
                    buffer.action[t] = a_pre
                    buffer.noise[t] = noise
                self._vec_env_step(env, a_pre, obs, buffer.state[t + 1], buffer.reward[t], buffer.done[t])
                t += 1
        with torch.no_grad():
            if reward_scale != 1.0:
                buffer.reward[:t] *= reward_scale
            buffer.mask[:t] = (1.0 - buffer.done[:t].to(torch.float32)) * gamma  # 0.0 if done else gamma
        buffer.length = t
        return t * N

    # ---- learning ----------------------------------------------------------------------------------------
    def _trajectory_views(self, buffer):
        """(reward, mask, action, noise, state) flattened in storage order plus the [T, N] shape for the scan."""
        buffer.update_now_len_before_sample()
        if isinstance(buffer, TrajectoryBuffer):
            T, N = buffer.length, buffer.num_envs
        else:
            T, N = buffer.now_len, 1  # flat time-ordered ring: one lane
        rew, mask, action, noise, state = buffer.sample_all()
        return T, N, rew, mask, action, noise, state

    def update_net(self, buffer, _target_step, batch_size, repeat_times=4):
        T, N, buf_reward, buf_mask, buf_action, buf_noise, buf_state = self._trajectory_views(buffer)
        buf_len = T * N
        dev = buf_state.device
        with torch.no_grad():
            buf_value = self.state_value(buf_state)                                # agent.py:619-620
            buf_logprob = self.act.old_logprob(buf_noise)                          # :621
            buf_r_sum, buf_advantage = self.compute_reward(buf_len, buf_reward, buf_mask, buf_value, shape=(T, N))

        n_steps = int(repeat_times * buf_len / batch_size)                         # :629
        fused = self._fused_grad(batch_size)
        if fused is not None:
            return self._update_fused(fused, n_steps, buf_len, batch_size, repeat_times, buf_state, buf_action,
                                      buf_r_sum, buf_logprob, buf_advantage)
        sums = torch.zeros(4, device=dev)  # united, actor, critic, entropy
        obj_actor = obj_critic = torch.zeros((), device=dev)
        params = [p for g in self.optimizer.param_groups for p in g["params"]]
        for step in range(n_steps):
            indices = self._minibatch_indices(step, buf_len, batch_size, dev)
            state = buf_state[indices]
            action = buf_action[indices]
            r_sum = buf_r_sum[indices]
            logprob = buf_logprob[indices]
            advantage = buf_advantage[indices]

            new_logprob = self.act.compute_logprob(state, action)
            ratio = (new_logprob - logprob).exp()
            surrogate = torch.min(advantage * ratio,
                                  advantage * ratio.clamp(1 - self.ratio_clip, 1 + self.ratio_clip))
            obj_entropy = (new_logprob.exp() * new_logprob).mean()                 # ElegantRL's entropy proxy (:643)
            obj_actor = -surrogate.mean() + obj_entropy * self.lambda_entropy
            value = self.cri(state).squeeze(1)
            obj_critic = self.criterion(value, r_sum)
            if self.dp is None:
                obj_united = obj_actor + obj_critic / (r_sum.std() + 1e-5)         # :652
                self.optimizer.zero_grad(set_to_none=False)
                obj_united.backward()
            else:
                # Data parallel: the minibatch of :652 is the UNION of the ranks' minibatches.  Actor and critic parameters are
                # disjoint and the united loss is linear in the critic's factor, so: back-propagate actor + UNSCALED critic, let
                # the one flat all-reduce of the step also carry (sum r, sum r^2, count), then scale the averaged critic gradient
                # by 1 / (std of the union + 1e-5) -- the same weights as one rank stepping on the concatenated minibatch.
                self.optimizer.zero_grad(set_to_none=False)
                (obj_actor + obj_critic).backward()
                r64 = r_sum.detach().double()
                mom = self.dp.average_gradients(params, extra=torch.stack([r64.sum(), (r64 * r64).sum(),
                                                                           torch.tensor(float(r64.numel()), dtype=torch.float64)]).float())
                G = float(self.dp.world)
                n_tot, s1, s2 = G * mom[2].double(), G * mom[0].double(), G * mom[1].double()
                var = ((s2 - s1 * s1 / n_tot) / (n_tot - 1.0)).clamp_min(0.0)
                scale = (1.0 / (var.sqrt().float() + 1e-5))
                obj_united = obj_actor + obj_critic * scale                        # (logged: this rank's terms, the union's scale)
                cri_params = {id(p) for p in self.cri.parameters()}
                with torch.no_grad():
                    for p in params:
                        if id(p) in cri_params and p.grad is not None:
                            p.grad.mul_(scale)
            self.optimizer.step()
            sums += torch.stack([obj_united.detach(), obj_actor.detach(), obj_critic.detach(), obj_entropy.detach()])
        self.weights_changed()
        self._n_updates += int(repeat_times)
        if n_steps:
            mean = (sums / n_steps).tolist()                                       # the only host sync of the update
            self._log_losses(*mean)
        return float(obj_actor.detach()), float(obj_critic.detach())

    def _minibatch_indices(self, step, buf_len, batch_size, dev, out=None):
        if self.index_hook is not None:
            return self.index_hook(step, buf_len, batch_size).to(dev)
        if out is not None:   # same draws, written where the captured graph reads them (saves a copy launch per step)
            return torch.randint(buf_len, size=(batch_size,), device=dev, out=out)
        return torch.randint(buf_len, size=(batch_size,), device=dev)              # agent.py:630

    @staticmethod
    def _log_losses(united, actor, critic, entropy):
        logger.record("train/united_loss", united)
        logger.record("train/actor_loss", actor)
        logger.record("train/critic_loss", critic)
        logger.record("train/entropy_losses", entropy)

    def _fused_grad(self, batch_size):
        """The fused HIP gradient path when the backend offers it for these nets (width 64/128, action_dim 1, GPU);
        otherwise None and the update runs through torch autograd on the same device."""
        if not self.use_fused_update or not hasattr(self.backend, "fused_ppo"):
            return None
        f = self._packed.get("fused")
        if f is None or f.max_batch < batch_size:
            f = self.backend.fused_ppo(self.act, self.cri, batch_size)
            self._packed["fused"] = f
            if f:  # parameters now live in one flat tensor: give Adam that tensor (fresh state, as after init)
                self.optimizer = f.make_optimizer(self.learning_rate)
        return f if f else None

    def _update_fused(self, fused, n_steps, buf_len, batch_size, repeat_times, buf_state, buf_action, buf_r_sum,
                      buf_logprob, buf_advantage):
        """Per optimizer step: indices -> minibatch r_sum scale -> three HIP launches (critic, actor, slab reduction) that leave
        d(obj_united)/d(theta) in the flat gradient buffer.  On one GPU the reduction also applies Adam and writes the new
        parameter values into the packed weight images (pime_ppo_minibatch_step + image map): nothing else is launched.  Under
        data parallelism ONE all-reduce of the flat buffer follows, then the Adam launch (which keeps the images current as well).

        The launch sequence of a step is identical every time, so after one eager step (which also creates Adam's state) it is
        captured into HIP graphs and replayed (the update is otherwise bound by ~200 us/step of host work): ONE graph per step
        with torch's own index draw (all minibatches of the update drawn at once into a table the kernels walk with a device-side
        cursor; the RCCL all-reduce is captured inside it), and from the second update on ONE graph for all n_steps steps of the
        update; two graphs per step -- [gradients] and [Adam] -- only where something host-side sits in between (an injected index
        tensor, the bench's launch timer, a collective that refuses capture)."""
        dev = buf_state.device
        fused.loss_sums.zero_()
        st = self._fused_static(fused, buf_len, batch_size, buf_state.shape[1], dev)
        st.r_sum.copy_(buf_r_sum); st.logprob.copy_(buf_logprob); st.adv.copy_(buf_advantage)
        # states / actions: the trajectory buffer's storage is already contiguous and address-stable; a flat ring
        # buffer hands out strided column views, which are copied once per update
        action = buf_action.reshape(-1)
        if not (action.is_contiguous() and buf_state.is_contiguous()):
            st.action.copy_(action); st.state.copy_(buf_state)
            action, buf_state = st.action, st.state
        # everything a captured graph bakes in besides the static tensors: data pointers, the loss scalars (launch arguments)
        # and the optimizer object whose buffers and learning rate the Adam launch reads
        key = (buf_state.data_ptr(), action.data_ptr(), float(self.ratio_clip), float(self.lambda_entropy),
               id(self.optimizer), float(getattr(self.optimizer, "lr", self.learning_rate)))
        if st.key != key:
            st.key, st.graph_a, st.graph_b, st.graph_full = key, None, None, None
            st.graph_update, st.graph_update_steps = None, None

        # Minibatch indices: with torch's own draw, all n_steps minibatches are drawn at once (agent.py:630 draws them one
        # torch.randint per step) into a table the kernels walk with a device-side row cursor, so a captured graph needs
        # no per-step input.  An index hook (parity tests) hands over one tensor per step instead.
        use_table = self.index_hook is None
        if use_table:
            if st.table is None or st.table.shape[0] < n_steps:
                st.table = torch.empty((n_steps, batch_size), dtype=torch.int64, device=dev)
                st.graph_a = st.graph_b = st.graph_full = None
                st.graph_update, st.graph_update_steps = None, None
            if self.index_table_hook is not None:
                st.table[:n_steps].copy_(self.index_table_hook(n_steps, buf_len, batch_size).to(dev))
            else:
                torch.randint(buf_len, size=(n_steps, batch_size), device=dev, out=st.table[:n_steps])
            st.row.zero_()

        # Single GPU: the Adam step rides in the gradient call's last launch (the slab reduction; pime_ppo_minibatch_step).
        # Data parallel (the all-reduce sits between gradients and Adam), the bench's gradient-only event bracket, a torch
        # optimizer, or nets on the split pipeline keep the separate Adam launch.
        from ..ops import FlatAdam
        fuse_adam = (self.dp is None and self.launch_timer is None and isinstance(self.optimizer, FlatAdam)
                     and getattr(fused, "adam_fusable", True))

        # Data parallel: the critic's gradient leaves the kernels UNSCALED with the minibatch's target moments behind it; the one
        # all-reduce of an optimizer step carries both, and the Adam launch applies 1 / (std of the UNION minibatch + 1e-5)
        # (agent.py:652 on the minibatch the ranks hold together).  Nets whose critic takes the split pipeline (a modular actor
        # on a stacked observation) keep the rank-local scale.
        dp_union = self.dp is not None and isinstance(self.optimizer, FlatAdam) and getattr(fused, "dp_union_ok", True)

        def grads():   # overwrite: no zeroing launch; the running sum of the critic scale lands in loss_sums[3]
            fused(buf_state, action, st.logprob, st.adv, st.r_sum, st.table if use_table else st.idx, self.ratio_clip,
                  self.lambda_entropy, st.scale, overwrite=True, index_row=st.row if use_table else None,
                  adam=self.optimizer if fuse_adam else None, defer_critic_scale=dp_union)

        def all_reduce():
            self.dp.all_reduce_mean(fused.flat_grad_dp if dp_union else fused.flat_grad)

        # With the image map the launch that applies Adam writes every new parameter value into the packed images as well (the
        # fused step on one GPU, pime_adam_step_images behind the all-reduce under data parallelism): no re-pack launch.
        images_follow = isinstance(self.optimizer, FlatAdam) and fused.images_follow_step

        def apply():
            if not fuse_adam:
                if dp_union:
                    self.optimizer.step(images=fused if images_follow else None, dp=(fused, self.dp.world))
                elif images_follow:
                    self.optimizer.step(images=fused)
                else:
                    self.optimizer.step()
            if not images_follow:
                fused.repack()

        if dp_union and not getattr(fused, "dp_union_probed", False):   # does the library defer the scale for these nets?
            from ..native import PimeError
            fused.dp_union_probed = True
            snap = [t.clone() for t in (fused.loss_sums, st.row)]
            try:
                grads()
            except PimeError as exc:
                print(f"| critic scale over the union minibatch unavailable for these nets ({exc}); using the rank-local scale")
                fused.dp_union_ok, dp_union = False, False
            for dst, src in zip((fused.loss_sums, st.row), snap):
                dst.copy_(src)

        if fuse_adam and not getattr(fused, "adam_probed", False):   # does the library fuse the step for these nets?
            from ..native import PimeError
            fused.adam_probed = True
            snap = [t.clone() for t in (fused.flat_param, self.optimizer.exp_avg, self.optimizer.exp_avg_sq,
                                        self.optimizer.step_count, fused.loss_sums, st.row)]
            try:
                grads()
            except PimeError:
                fused.adam_fusable, fuse_adam = False, False
            for dst, src in zip((fused.flat_param, self.optimizer.exp_avg, self.optimizer.exp_avg_sq,
                                 self.optimizer.step_count, fused.loss_sums, st.row), snap):
                dst.copy_(src)   # the probe must leave no trace
            if images_follow:
                fused.repack()   # ... nor in the packed images it may have updated

        def capture(*thunks):
            torch.cuda.synchronize(dev)
            g = torch.cuda.CUDAGraph()
            # thread_local: the RCCL watchdog thread of a data-parallel run may touch the HIP API meanwhile
            with _no_gc(), torch.cuda.graph(g, capture_error_mode="thread_local"):
                for thunk in thunks:
                    thunk()
            return g

        # Data parallel: the flat-gradient all-reduce is an RCCL kernel on the compute stream, so it is captured between the
        # gradient launches and Adam like any other launch: ONE graph per optimizer step there too (the two-graph sequence with
        # an eager all-reduce in between cost 14 % on one rank before any communication).  If the capture fails (a torch / RCCL
        # build that refuses collectives under capture) the agent falls back to the two-graph sequence for good.
        in_graph_dp = self.dp is not None and self.use_graph_collective and getattr(self.dp, "graph_capturable", False)
        one_graph = self.use_single_graph and (self.dp is None or in_graph_dp) and use_table and self.launch_timer is None
        if st.mode != (use_table, one_graph, fuse_adam, dp_union):   # the captured graphs bake in the index source and the step form
            st.mode, st.graph_a, st.graph_b, st.graph_full = (use_table, one_graph, fuse_adam, dp_union), None, None, None
            st.graph_update, st.graph_update_steps = None, None
        last = None
        # With the index table every optimizer step is the same launch sequence (the row cursor lives on the device), so once
        # the per-step graph exists the WHOLE update -- n_steps x (critic, actor, reduction [, all-reduce, Adam]) and the copy of
        # the loss sums in front of the last step -- is captured as one graph: one replay per update_net instead of n_steps, and
        # the ~5 us between two replays become a node boundary.
        if (self.use_update_graph and one_graph and self.use_hip_graphs and st.graph_full is not None and n_steps > 1
                and getattr(st, "graph_update_steps", None) != n_steps):
            st.graph_update, st.graph_update_steps = None, n_steps
            try:
                def whole_update():
                    for k in range(n_steps):
                        if k == n_steps - 1:
                            st.last.copy_(fused.loss_sums)
                        grads()
                        if self.dp is not None:
                            all_reduce()
                        apply()
                st.graph_update = capture(whole_update)
            except RuntimeError as exc:
                print(f"| capture of the whole update refused ({exc}); replaying one graph per optimizer step")
                self.use_update_graph = False
                torch.cuda.synchronize(dev)
        if st.graph_update is not None and st.graph_update_steps == n_steps and one_graph and self.use_update_graph:
            st.graph_update.replay()
            last = st.last.clone()
            n_loop = 0
        else:
            n_loop = n_steps
        for step in range(n_loop):
            if not use_table:
                st.idx.copy_(self.index_hook(step, buf_len, batch_size).to(dev))
            if step == n_steps - 1:
                last = fused.loss_sums.clone()
            if self.use_hip_graphs and st.warm and st.graph_a is None and st.graph_full is None:
                try:
                    if one_graph and self.dp is not None:
                        refused = None
                        try:
                            st.graph_full = capture(grads, all_reduce, apply)
                        except RuntimeError as exc:
                            refused = exc
                        # the ranks must use the SAME launch form from here on (a rank replaying the collective from its graph
                        # while another issues it eagerly between two graphs would still match up, but a rank-local failure
                        # must not go unnoticed): one MAX over the ranks decides for all of them
                        if self.dp.max_over_ranks(1.0 if refused is not None else 0.0) > 0.5:
                            print(f"| all-reduce inside the HIP graph refused on a rank ({refused}); every rank uses the "
                                  "two-graph step sequence")
                            self.use_graph_collective, one_graph = False, False
                            st.mode = (use_table, one_graph, fuse_adam, dp_union)
                            st.graph_full = None
                            torch.cuda.synchronize(dev)
                            st.graph_a, st.graph_b = capture(grads), capture(apply)
                    elif one_graph:
                        st.graph_full = capture(grads, apply)
                    else:
                        st.graph_a = capture(grads)
                        st.graph_b = None if (images_follow and fuse_adam) else capture(apply)   # nothing left to launch after a fused step
                except RuntimeError as exc:  # keep training on the eager launch sequence
                    print(f"| HIP graph capture failed ({exc}); continuing with eager launches")
                    self.use_hip_graphs = False
                    torch.cuda.synchronize(dev)
            if st.graph_full is not None:
                st.graph_full.replay()
                continue
            run = st.graph_a.replay if st.graph_a is not None else grads
            if self.launch_timer is not None:   # bench.py: HIP events around the gradient launches only
                self.launch_timer("ppo_minibatch_grad", run)
            else:
                run()
            if self.dp is not None:
                all_reduce()
            if st.graph_b is not None:
                st.graph_b.replay()
            else:
                apply()
                st.warm = True
        self._packed = {"fused": fused}  # packed forward images of the value pass / rollout are stale now
        self._n_updates += int(repeat_times)
        if not n_steps:
            return 0.0, 0.0
        tot = fused.loss_sums.tolist()                                             # the only host sync of the update
        if self.dp is not None:
            self.dp.check()   # a timed-out one-shot all-reduce left gradients un-averaged: fatal, here where the stream is drained
        lst = last.tolist()
        B = float(batch_size)
        ent, cri = tot[1] / (n_steps * B), tot[2] / (n_steps * B)
        act = tot[0] / (n_steps * B) + self.lambda_entropy * ent
        self._log_losses(act + tot[4] / (n_steps * B), act, cri, ent)   # mean over steps of (actor + critic * scale), agent.py:652
        obj_a = (tot[0] - lst[0]) / B + self.lambda_entropy * (tot[1] - lst[1]) / B
        obj_c = (tot[2] - lst[2]) / B
        return obj_a, obj_c

    def _fused_static(self, fused, buf_len, batch_size, state_dim, dev):
        """Tensors with stable addresses that the captured graphs read (the per-update r_sum / log-prob / advantage
        buffers are fresh allocations, so they are copied in)."""
        st = getattr(fused, "static", None)
        if st is None or st.buf_len != buf_len or st.batch != batch_size:
            import types
            f32 = dict(dtype=torch.float32, device=dev)
            st = types.SimpleNamespace(buf_len=buf_len, batch=batch_size, key=None, graph_a=None, graph_b=None, graph_full=None, table=None, mode=None, warm=False,
                                       graph_update=None, graph_update_steps=None, last=torch.zeros(6, **f32),
                                       row=torch.zeros(1, dtype=torch.int64, device=dev),
                                       r_sum=torch.empty(buf_len, **f32), logprob=torch.empty(buf_len, **f32),
                                       adv=torch.empty(buf_len, **f32), scale=torch.ones(1, **f32),
                                       action=torch.empty(buf_len, **f32), state=torch.empty((buf_len, state_dim), **f32),
                                       scale_sum=torch.zeros(1, **f32),
                                       idx=torch.zeros(batch_size, dtype=torch.int64, device=dev))
            fused.static = st
        return st

    def _normalise_advantage(self, adv):
        """(adv - mean) / (std + 1e-5) over the WHOLE buffer with torch's unbiased std (agent.py:707); under data
        parallelism the buffer is the union of all ranks' slices -> all-reduce (count, sum, sum of squares)."""
        if self.dp is None:
            return (adv - adv.mean()) / (adv.std() + 1e-5)
        a64 = adv.double()
        m = torch.stack([torch.tensor(float(adv.numel()), dtype=torch.float64, device=adv.device), a64.sum(),
                         (a64 * a64).sum()])
        self.dp.all_reduce_sum(m)
        n, s, ss = m[0], m[1], m[2]
        mean = s / n
        var = (ss - n * mean * mean) / (n - 1)
        return ((a64 - mean) / (var.clamp_min(0).sqrt() + 1e-5)).float()

    def compute_reward_gae(self, buf_len, buf_reward, buf_mask, buf_value, shape=None):
        """r_sum and GAE advantage, ElegantRL's recursion (agent.py:685-708), as one reverse scan per env lane."""
        T, N = shape if shape is not None else (buf_len, 1)
        value = buf_value.reshape(-1)
        r_sum, adv = self.backend.gae(buf_reward.reshape(T, N), buf_mask.reshape(T, N), value.reshape(T, N),
                                      self.lambda_gae_adv, True)
        return r_sum.reshape(-1), self._normalise_advantage(adv.reshape(-1))

    def compute_reward_adv(self, buf_len, buf_reward, buf_mask, buf_value, shape=None):
        T, N = shape if shape is not None else (buf_len, 1)
        value = buf_value.reshape(-1)
        r_sum, adv = self.backend.gae(buf_reward.reshape(T, N), buf_mask.reshape(T, N), value.reshape(T, N), 0.0, False)
        return r_sum.reshape(-1), self._normalise_advantage(adv.reshape(-1))


# ================================================================================================= TD3
class AgentTD3(AgentBase):
    """Twin-delayed DDPG (agent.py:276-394): twin critics, target policy smoothing, delayed soft target updates.

    One-instance env + flat ring buffer: the reference's loop, op for op (pinned against the reference's weights by
    tests/test_td3_golden_cpu.py).  Vectorised env (`env.num_envs`) + `VecReplayBuffer`: all lanes step in lock-step through
    the HIP env kernel, transitions stay in HBM, and `update_net` runs target_step / num_envs * repeat_times optimizer steps
    (the reference's "one gradient step per env step" counted per LOCK-STEP, not per lane).  On the GPU an optimizer step is
    four hand-written launches (`pime_td3_step`, csrc/td3_fused.hip: critic gradients, slab reduction + Adam + delayed soft
    update, actor gradients through the target critic, the same for the actor), a whole update_net one HIP graph; shapes the
    kernels do not serve (state_dim > 7, widths other than 64 / 128, data parallel) and CPU tensors run the same arithmetic as
    PyTorch modules (`_one_update`)."""

    def __init__(self, backend=None, device=None):
        super().__init__(backend, device)
        self.explore_noise = 0.1
        self.policy_noise = 0.2
        self.update_freq = 2
        self.use_hip_graphs = True
        self.use_fused_rollout = True   # vectorised env: the whole explore call as ONE launch (csrc/rollout_offpolicy.hip)
        # target Q of the critic objective through pime_mlp_forward (_target_packs).  OFF by default: at the TD3 batch of 4 096 rows a
        # forward is 16 workgroups on three serial 256-MFMA chains (~25 us each), slower than rocBLAS's small GEMMs -- 165 vs 157 ms
        # per bench step.  PIME_TD3_FUSED_TARGETS=1 switches it on (parity: tests/test_gpu_td3.py).
        self.use_fused_targets = os.environ.get("PIME_TD3_FUSED_TARGETS", "0") == "1"
        self.use_fused_update = os.environ.get("PIME_TD3_FUSED", "1") == "1"   # the optimizer step on the hand-written kernels
        # critic / actor chains as parallel graph branches (rows without a soft update): bit-identical, measured NEUTRAL (58.1 vs 58.4 M:
        # each launch already fills the chip, overlapped launches only stretch each other -- DESIGN.md section 4c), so off by default
        self.use_two_streams = os.environ.get("PIME_TD3_TWO_STREAMS", "0") == "1"
        self.draw_hook = None      # tests: callable(n_steps, batch) -> (idx, nxt, noise) tables of a whole update (injected draws)
        self.launch_timer = None   # bench.py: callable(name, fn) timing one update's launches with HIP events
        self._fused_td3 = None
        self._graphs = None
        self._obs = None
        self._packed_act = None
        self._tpacks = None

    def init(self, net_dim, state_dim, action_dim, if_per=False):
        assert not if_per, "prioritised replay is not on the residual-control path"
        self._pick_device()
        from copy import deepcopy
        self.cri = CriticTwin(net_dim, state_dim, action_dim).to(self.device)
        self.cri_target = deepcopy(self.cri)
        self.act = Actor(net_dim, state_dim, action_dim).to(self.device)
        self.act_target = deepcopy(self.act)
        self._make_optimizers()
        self.criterion = torch.nn.SmoothL1Loss()
        self.get_obj_critic = self.get_obj_critic_raw

    def _make_optimizers(self):
        kw = dict(fused=True, capturable=True) if self.device.type == "cuda" else {}
        self.cri_optimizer = torch.optim.Adam(self.cri.parameters(), lr=self.learning_rate, **kw)
        self.act_optimizer = torch.optim.Adam(self.act.parameters(), lr=self.learning_rate, **kw)
        self._graphs = None
        self._packed_act = None
        self._tpacks = None
        self._fused_td3 = None   # its Adam moments belong to the optimizers just replaced

    def weights_changed(self):
        super().weights_changed()
        self._tpacks = None   # the target nets' packed images are stale (checkpoint load, rebuilt optimizer)
        self._graphs = None   # (the fused step reads the parameters where they live: nothing of it goes stale)

    def _fused_step(self, batch_size):
        """ops.FusedTD3 serving the current nets, or None -> the PyTorch modules (_one_update)."""
        if not self.use_fused_update or self.device.type != "cuda" or not hasattr(self.backend, "fused_td3"):
            return None
        f = self._fused_td3
        if f is False:
            return None
        if f is None or not f.wraps(self):
            f = self._fused_td3 = self.backend.fused_td3(self, batch_size)
            if f is False:
                return None
        f.ensure_batch(batch_size)
        return f

    def _target_packs(self):
        """(actor_target, q1 head, q2 head of cri_target) as packed images of the hand-written forward kernel, or None -> the torch
        modules.  The no-grad half of the critic objective (agent.py:363-367: next_a = act_target.get_action(next_s),
        next_q = min(cri_target.get_q1_q2(next_s, next_a))) is then three pime_mlp_forward launches on the f32 matrix cores (the twin
        heads share the trunk through an identity layer: backend.packed_twin_heads) instead of ~15 rocBLAS / elementwise ones; the
        images are re-packed right behind every soft update (_one_update), inside the same captured graph."""
        if self._tpacks is None:
            self._tpacks = False
            ok = (self.use_fused_targets and self.device.type == "cuda" and hasattr(self.backend, "packed_twin_heads")
                  and getattr(self.act_target, "action_dim", 1) == 1)
            if ok:
                a = self.backend.packed(self.act_target)
                q = self.backend.packed_twin_heads(self.cri_target) if a is not None else None
                if a is not None and q is not None:
                    self._tpacks = (a, q[0], q[1])
        return self._tpacks or None

    def _prior_term(self, states):
        """Prior-controller part of the env action (none for plain TD3)."""
        return None

    def _rollout_priorK(self):
        """float64 prior gain of the fused exploration kernel's composition a_env = a + s @ priorK (zeros: plain TD3)."""
        return np.zeros(self.act.state_dim)

    def _fused_explore(self, env):
        """Packed deterministic actor for the fused exploration kernel, or None -> lock-step by lock-step launches.  The image is
        re-packed on every call: the actor's weights change with every update_net."""
        if not (self.use_fused_rollout and hasattr(env, "offpolicy_rollout_supported") and hasattr(self.backend, "packed")):
            return None
        if getattr(self.act, "action_dim", 1) != 1:
            return None
        if self._packed_act is None or self._packed_act is False:
            if self._packed_act is False:
                return None
            self._packed_act = self.backend.packed(self.act) or False
            if self._packed_act is False:
                return None
        pk = self._packed_act
        if not env.offpolicy_rollout_supported(pk):
            return None
        if not hasattr(self, "_rollout_seed"):
            self._rollout_seed = int(torch.initial_seed()) & (2 ** 63 - 1)   # exploration stream follows torch's seed
            self._rollout_epoch = 0
        pk.repack()
        return pk

    def select_action(self, state, if_deterministic=False):
        states = torch.as_tensor(np.asarray(state)[None], dtype=torch.float32, device=self.device)
        with torch.no_grad():
            action = self.act(states)[0]
            if not if_deterministic:
                action = (action + torch.randn_like(action) * self.explore_noise).clamp(-1, 1)
        return action.cpu().numpy()

    def _env_action(self, state, action):
        """What a one-instance env receives for the stored `action` (the residual agents add the prior term)."""
        return action

    def explore_env(self, env, buffer, target_step, reward_scale, gamma):
        if hasattr(env, "num_envs"):
            return self.explore_vec_env(env, buffer, target_step, reward_scale, gamma)
        for _ in range(target_step):   # agent.py:54-70, continuing from self.state
            action = self.select_action(self.state)
            next_s, reward, done, _ = env.step(self._env_action(self.state, action))
            buffer.append_buffer(self.state, (reward * reward_scale, 0.0 if done else gamma, *action))
            self.state = env.reset() if done else next_s
        return target_step

    def explore_vec_env(self, env, buffer, target_step, reward_scale, gamma):
        """target_step transitions = target_step / N lock-steps of all N lanes, continuing the running episodes; finished
        lanes are reset inside the env kernel and their next row holds the new episode's first observation."""
        assert isinstance(buffer, VecReplayBuffer) and buffer.num_envs == env.num_envs
        N = env.num_envs
        steps = max(1, target_step // N)
        if buffer.stored_slots + steps < 2:
            steps = 2   # sampling needs one stored lock-step WITH a successor (replay.py: row i and row i + N)
        if self._obs is not None and getattr(self, "_obs_epoch", None) != (id(env), env.reset_count):
            # someone else reset this env since the last call (the evaluator, when it shares the training env): the cached
            # observation is stale and the lanes sit in a post-evaluation state.  Start new episodes, and cut the newest
            # stored lock-step off from what follows it (its successor slot will hold a reset observation).
            buffer.cut_last_step()
            self._obs = None
        if self._obs is None:
            self._obs = env.reset().clone()
            self._next_obs = torch.empty_like(self._obs)
            self._obs_epoch = (id(env), env.reset_count)
        pk = self._fused_explore(env)
        if pk is not None:   # ONE launch for the whole call: actor forward, noise, composition, env step, ring writes
            done_steps = 0
            while done_steps < steps:
                n = min(steps - done_steps, buffer.slots)
                self._rollout_epoch += 1
                env.rollout_offpolicy(pk, self._rollout_priorK(), self.explore_noise, gamma, reward_scale, n, self._rollout_seed,
                                      self._rollout_epoch, self._obs, buffer.state, buffer.other, buffer.next_slot)
                buffer.advance(n)
                done_steps += n
            return steps * N
        for _ in range(steps):
            obs = self._obs
            with torch.no_grad():
                a = self.act(obs)
                a = (a + torch.randn_like(a) * self.explore_noise).clamp(-1, 1)   # agent.py:305
                prior = self._prior_term(obs)
                a_env = a if prior is None else a + prior
            _, rew, done = env.step(a_env, auto_reset=True, out_obs=self._next_obs)
            with torch.no_grad():
                mask = (1.0 - done.to(torch.float32)) * gamma
                buffer.append_step(obs, rew * reward_scale if reward_scale != 1.0 else rew, mask, a)
            self._obs, self._next_obs = self._next_obs, self._obs
        return steps * N

    def get_obj_critic_raw(self, buffer, batch_size):
        with torch.no_grad():
            reward, mask, action, state, next_s = buffer.sample_batch(batch_size)
            tp = self._target_packs() if next_s.is_cuda else None
            if tp is not None:   # the same arithmetic, the forwards on the hand-written kernel (net.py: Actor.get_action)
                a = tp[0](next_s).tanh().unsqueeze(1)
                noise = (torch.randn_like(a) * self.policy_noise).clamp(-0.5, 0.5)
                next_a = (a + noise).clamp(-1.0, 1.0)
                sa = torch.cat((next_s, next_a), dim=1)
                next_q = torch.min(tp[1](sa), tp[2](sa)).unsqueeze(1)
            else:
                next_a = self.act_target.get_action(next_s, self.policy_noise)
                next_q = torch.min(*self.cri_target.get_q1_q2(next_s, next_a))
            q_label = reward + mask * next_q
        q1, q2 = self.cri.get_q1_q2(state, action)
        return self.criterion(q1, q_label) + self.criterion(q2, q_label), state

    def _one_update(self, buffer, batch_size, soft):
        """One iteration of the reference's loop (agent.py:314-331).  Data parallel (the reference has no collective): every rank
        samples its own minibatch from its own lanes and the gradients of BOTH backward passes are averaged before their optimizer
        step -- the two means over batch_size samples become the means over the union of the ranks' minibatches, so G ranks make the
        step of one rank on a G x batch_size minibatch (tests/test_dist_gloo.py) and the replicas stay identical.  Two all-reduces per
        step: the actor's objective needs the critic's step applied."""
        obj_critic, state = self.get_obj_critic(buffer, batch_size)
        self.cri_optimizer.zero_grad(set_to_none=False)
        obj_critic.backward()
        if self.dp is not None:
            self.dp.average_gradients([p for p in self.cri.parameters() if p.grad is not None])
        self.cri_optimizer.step()
        tp = self._tpacks or None
        if soft:
            self.soft_update(self.cri_target, self.cri, self.soft_update_tau)
            if tp is not None:
                tp[1].repack(); tp[2].repack()
        obj_actor = -self.cri_target(state, self.act(state)).mean()
        self.act_optimizer.zero_grad(set_to_none=False)
        obj_actor.backward()
        if self.dp is not None:
            self.dp.average_gradients([p for p in self.act.parameters() if p.grad is not None])
        self.act_optimizer.step()
        if soft:
            self.soft_update(self.act_target, self.act, self.soft_update_tau)
            if tp is not None:
                tp[0].repack()
        return obj_actor.detach(), obj_critic.detach()

    def update_net(self, buffer, target_step, batch_size, repeat_times):
        buffer.update_now_len_before_sample()
        dev = self.device
        vec = isinstance(buffer, VecReplayBuffer)
        n_steps = int(target_step * repeat_times) if not vec else max(1, int(target_step // buffer.num_envs * repeat_times))
        fused = self._fused_step(batch_size) if n_steps else None
        if fused is not None:
            return self._update_fused(fused, buffer, n_steps, batch_size, int(target_step if not vec else n_steps))
        sums = torch.zeros(2, device=dev)
        obj_actor = obj_critic = torch.zeros((), device=dev)
        use_graphs = vec and self.use_hip_graphs and dev.type == "cuda" and self.dp is None   # (the all-reduces run eagerly)
        graphs = self._graphs if use_graphs else None
        key = (id(buffer), batch_size)
        for i in range(n_steps):
            soft = i % self.update_freq == 0
            if use_graphs and (i >= 2 or (graphs and graphs.get("key") == key)):
                # the step's launch sequence is fixed once Adam's state exists (two eager steps): capture it twice (with /
                # without the delayed soft update) and replay.  The sampler reads its index bounds from the device
                # (VecReplayBuffer._bounds), so the two graphs serve every later call as well.
                if graphs is None or graphs.get("key") != key:
                    graphs = self._capture_updates(buffer, batch_size, key)
                    self._graphs = graphs
                if graphs:
                    g = graphs[soft]
                    g["graph"].replay()
                    sums += g["out"]
                    obj_actor, obj_critic = g["out"][0], g["out"][1]
                    continue
            obj_actor, obj_critic = self._one_update(buffer, batch_size, soft)
            sums += torch.stack([obj_actor, obj_critic])
        self._n_updates += int(target_step if not vec else n_steps)
        if n_steps:
            mean = (sums / n_steps).tolist()
            logger.record("train/n_updates", self._n_updates, exclude="tensorboard")
            logger.record("train/actor_loss", mean[0])
            logger.record("train/critic_loss", mean[1])
        return float(obj_actor), float(obj_critic) / 2

    def _update_fused(self, f, buffer, n_steps, batch_size, n_updates):
        """update_net on the fused step: the sampled rows of ALL n_steps optimizer steps are drawn at once into an index table that
        the kernels read by row (a launch argument), the smoothing noise is drawn inside
        the critic kernel (Philox stream 3; `draw_hook` injects tables instead), and from the second call on the whole update --
        n_steps x 4 launches -- is ONE HIP graph.  The only host synchronisation is the read of the four loss words at the end."""
        dev = self.device
        vec = isinstance(buffer, VecReplayBuffer)
        st = getattr(f, "tables", None)
        if st is None or st["shape"] != (n_steps, batch_size):
            i64 = dict(dtype=torch.int64, device=dev)
            st = f.tables = {"shape": (n_steps, batch_size), "idx": torch.zeros((n_steps, batch_size), **i64),
                             "nxt": torch.zeros((n_steps, batch_size), **i64), "noise": None, "graph": None, "key": None, "warm": False}
        idx, nxt = st["idx"], st["nxt"]
        if self.draw_hook is not None:
            h_idx, h_nxt, h_noise = self.draw_hook(n_steps, batch_size)
            idx.copy_(torch.as_tensor(h_idx).to(dev)); nxt.copy_(torch.as_tensor(h_nxt).to(dev))
            if st["noise"] is None:
                st["noise"] = torch.zeros((n_steps, batch_size), dtype=torch.float32, device=dev)
            st["noise"].copy_(torch.as_tensor(h_noise).to(dev).reshape(n_steps, batch_size))
        elif vec:   # VecReplayBuffer.sample_indices for the whole table: uniform over the rows that have a successor
            assert buffer.stored_slots >= 2, "need two stored steps before sampling"
            N = buffer.num_envs
            u = torch.randint(2 ** 62, (n_steps, batch_size), device=dev) % buffer._bounds[0]   # bounds live on the device (replay.py)
            lane = u % N
            slot = (u // N + buffer._bounds[1]) % buffer.slots                   # slots in age order start at the oldest
            torch.add(slot * N, lane, out=idx)
            torch.add(((slot + 1) % buffer.slots) * N, lane, out=nxt)            # successor: same lane, next slot
        else:       # ReplayBuffer.sample_batch (replay.py:344-351): rows [0, now_len - 1), successor = the next row
            torch.randint(buffer.now_len - 1, (n_steps, batch_size), device=dev, out=idx)
            torch.add(idx, 1, out=nxt)
        noise = st["noise"] if self.draw_hook is not None else None
        if not hasattr(self, "_smooth_seed"):
            self._smooth_seed = (int(torch.initial_seed()) ^ 0x5DEECE66D) & (2 ** 63 - 1)   # smoothing noise follows torch's seed
        f.loss.zero_()
        f.begin_update()   # table row 0; the noise epoch advances (a captured graph draws fresh noise in every replay)

        side = st.get("side")
        if side is None and self.use_two_streams and self.dp is None:
            side = st["side"] = torch.cuda.Stream(device=dev)
        # data parallel: the two all-reduces of every step are captured inside the update's graph where the communicator allows it
        # (RCCL: yes; gloo and a refused capture: eager launches, decided for all ranks together)
        in_graph_dp = (self.dp is not None and getattr(self, "use_graph_collective", True)
                       and getattr(self.dp, "graph_capturable", False))
        can_graph = self.use_hip_graphs and (self.dp is None or in_graph_dp)

        def one(k, phases):   # the row is a launch argument: every node of the captured graph carries its own
            f.step(buffer.buf_state, buffer.buf_other, idx, nxt, noise, self.soft_update_tau, self.update_freq, self.policy_noise,
                   noise_seed=self._smooth_seed, row=k, phases=phases)

        def run():
            """The critic chain (gradients, apply) on the current stream, the actor chain on a side stream.  Only rows with the
            delayed soft update tie the two completely: the actor's gradients read the critic TARGET (written by the critic's apply
            on soft rows only) and the state rows its critic launch gathered; the next row's critic gradients read the actor TARGET
            (written by the actor's apply on soft rows only).  On the other rows the critic's apply runs beside the actor's
            gradients and the actor's apply beside the next row's critic gradients -- inside the captured graph these are parallel
            branches.  Same arithmetic, same bits as the one-stream order (tests/test_gpu_td3_fused.py)."""
            if self.dp is not None:
                # data parallel: a net's slab reduction leaves THIS rank's gradient, the ranks average it, Adam (+ the delayed soft
                # update) is applied from the averaged tensor -- five launches and two all-reduces per step (ops.FusedTD3.step_dp); G
                # ranks with their own minibatches make the step of one rank on the union minibatch
                for k in range(n_steps):
                    one(k, 1 | 16)
                    self.dp.all_reduce_mean(f.cri_grad)
                    one(k, 32 | 4 | 64)
                    self.dp.all_reduce_mean(f.act_grad)
                    one(k, 128)
                return
            if side is None:
                for k in range(n_steps):
                    one(k, 15)
                return
            main = torch.cuda.current_stream(dev)
            side.wait_stream(main)
            for k in range(n_steps):
                soft = k % self.update_freq == 0
                one(k, 1)                       # critic gradients (gathers the row's states)
                if soft:
                    one(k, 2)                   # critic apply writes the critic target: the actor's gradients must see it
                    side.wait_stream(main)
                else:
                    side.wait_stream(main)      # the actor's gradients need the gathered rows only
                    one(k, 2)
                with torch.cuda.stream(side):
                    one(k, 4)
                    one(k, 8)
                if soft or k == n_steps - 1:
                    main.wait_stream(side)      # the next critic gradients read the actor target this row's actor apply wrote

        key = (buffer.buf_state.data_ptr(), buffer.buf_other.data_ptr(), noise is None, self.soft_update_tau, self.update_freq,
               self.policy_noise, side is not None, self.dp is not None)
        if can_graph and st["warm"] and (st["graph"] is None or st["key"] != key):
            refused = None
            try:
                torch.cuda.synchronize(dev)
                g = torch.cuda.CUDAGraph()
                with _no_gc(), torch.cuda.graph(g, capture_error_mode="thread_local"):
                    run()
                st["graph"], st["key"] = g, key
            except RuntimeError as exc:
                refused = exc
            if self.dp is not None:
                # the ranks must agree on the launch form: one MAX over the ranks decides for all of them
                if self.dp.max_over_ranks(1.0 if refused is not None else 0.0) > 0.5:
                    print(f"| all-reduce inside the TD3 update's HIP graph refused on a rank ({refused}); every rank launches eagerly")
                    self.use_graph_collective, can_graph = False, False
                    torch.cuda.synchronize(dev)
                    st["graph"] = None
            elif refused is not None:
                print(f"| HIP graph capture of the TD3 update failed ({refused}); continuing with eager launches")
                self.use_hip_graphs = False
                torch.cuda.synchronize(dev)
                st["graph"] = None
        go = st["graph"].replay if (can_graph and st["graph"] is not None and st["key"] == key) else run
        if self.launch_timer is not None:
            self.launch_timer("td3_update", go)
        else:
            go()
        st["warm"] = True
        f.row = n_steps        # (begin_update of the next call moves them into the optimizers' step base)
        self._n_updates += n_updates
        tot = f.loss.tolist()   # the update's only host synchronisation
        if self.dp is not None:
            self.dp.check()     # a timed-out one-shot all-reduce left gradients un-averaged: fatal, here where the stream is drained
        logger.record("train/n_updates", self._n_updates, exclude="tensorboard")
        logger.record("train/actor_loss", tot[0] / n_steps)
        logger.record("train/critic_loss", tot[1] / n_steps)
        return tot[2], tot[3] / 2

    def _capture_updates(self, buffer, batch_size, key):
        out = {"key": key}
        try:
            torch.cuda.synchronize(self.device)
            for soft in (True, False):
                res = torch.zeros(2, device=self.device)
                g = torch.cuda.CUDAGraph()
                with _no_gc(), torch.cuda.graph(g, capture_error_mode="thread_local"):
                    oa, oc = self._one_update(buffer, batch_size, soft)
                    res.copy_(torch.stack([oa, oc]))
                out[soft] = {"graph": g, "out": res}
            return out
        except RuntimeError as exc:   # keep training on eager launches
            print(f"| HIP graph capture of the TD3 update failed ({exc}); continuing with eager launches")
            self.use_hip_graphs = False
            torch.cuda.synchronize(self.device)
            return {}
