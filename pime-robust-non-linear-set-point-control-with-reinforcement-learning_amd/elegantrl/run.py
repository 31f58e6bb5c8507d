"""Arguments, the single-process train/evaluate loop and the Evaluator
(interface of /root/reference/elegantrl/run.py: Arguments :14-90, train_and_evaluate :96-225, Evaluator :483-598,
get_episode_return :600-619).  The multiprocessing trainer (:228-477) is superseded by on-GPU vectorisation.

The loop is the reference's: explore -> update -> evaluate -> log, until break_step / target_return / a `stop`
file in cwd.  With a vectorised env one explore call advances all lanes for whole episodes, and the evaluator
averages one deterministic episode over the lanes of `env_eval`."""
import os
import time
from copy import deepcopy

import numpy as np
import torch

from . import logger
from .env import PreprocessEnv
from .replay import ReplayBuffer, TrajectoryBuffer, VecReplayBuffer


class Arguments:
    def __init__(self, agent=None, env=None, gpu_id=None, if_on_policy=False):
        self.agent = agent
        self.cwd = None
        self.env = env
        self.env_eval = None
        self.gpu_id = gpu_id
        self.net_dim = 2 ** 8
        self.batch_size = 2 ** 8
        self.repeat_times = 2 ** 0
        self.target_step = 2 ** 10
        self.learning_start = 0
        self.max_memo = 2 ** 17
        if if_on_policy:
            self.net_dim = 2 ** 9
            self.batch_size = 2 ** 9
            self.repeat_times = 2 ** 4
            self.target_step = 2 ** 12
            self.max_memo = self.target_step
        self.gamma = 0.99          # NB train.py never forwards --gamma / --learning_rate: 0.99 / 1e-4 rule
        self.reward_scale = 2 ** 0
        self.if_per = False
        self.rollout_num = 2
        self.num_threads = 8
        self.break_step = 2 ** 20
        self.if_remove = True
        self.if_allow_break = True
        self.eval_gap = 5
        self.eval_times1 = 2 ** 2
        self.eval_times2 = 2 ** 4
        self.random_seed = 0
        # attributes the reference's train.py attaches ad hoc and run.py then reads unguarded (run.py:136-160,174-175)
        self.SCN_kwargs = {}
        self.residual_kwargs = {}
        self.Modular_kwargs = {}
        self.Q_kwargs = {}
        self.if_residual = False
        self.fix_K = False
        self.frozen_modular_integrator = False
        self.frozen_transfer = False
        self.test_render = None
        self.test_render_times = 10000
        self.load = "None"

    def init_before_training(self, if_main=True):
        if self.agent is None:
            raise RuntimeError("\n| Why agent=None? Assignment args.agent = AgentXXX please.")
        if not hasattr(self.agent, "init"):
            raise RuntimeError("\n| There should be agent=AgentXXX() instead of agent=AgentXXX")
        if self.env is None:
            raise RuntimeError("\n| Why env=None? Assignment args.env = XxxEnv() please.")
        if isinstance(self.env, str) or not (hasattr(self.env, "env_name") or hasattr(self.env, "num_envs")):
            raise RuntimeError("\n| What is env.env_name? use env=PreprocessEnv(env). It is a Wrapper.")
        self.gpu_id = "0" if self.gpu_id is None or not str(self.gpu_id).isdigit() else str(self.gpu_id)
        if self.cwd is None:
            name = getattr(self.env, "env_name", type(self.env).__name__)
            self.cwd = f"./{self.agent.__class__.__name__}/{name}_{self.gpu_id}"
        if if_main:
            print(f"| GPU id: {self.gpu_id}, cwd: {self.cwd}")
            if self.if_remove:
                import shutil
                shutil.rmtree(self.cwd, ignore_errors=True)
                print("| Remove history")
            os.makedirs(self.cwd, exist_ok=True)
        torch.set_num_threads(self.num_threads)
        torch.set_default_dtype(torch.float32)
        torch.manual_seed(self.random_seed)
        np.random.seed(self.random_seed)


def make_buffer(agent, env, max_memo, if_per=False):
    """TrajectoryBuffer for a vectorised env under an on-policy agent, else the flat ring (run.py:168-169)."""
    on_policy = getattr(agent, "if_on_policy", False)
    if hasattr(env, "num_envs") and on_policy:
        per_episode = env.num_envs * env.max_step
        episodes = max(1, -(-max_memo // per_episode))
        return TrajectoryBuffer(episodes * env.max_step, env.num_envs, env.state_dim, env.action_dim, agent.device,
                                dtype=getattr(env, "trajectory_dtype", torch.float32))
    if hasattr(env, "num_envs"):   # off-policy agent on a vectorised env: the per-lane device ring
        return VecReplayBuffer(max_memo, env.num_envs, env.state_dim, env.action_dim, agent.device)
    return ReplayBuffer(max_len=max_memo + env.max_step, state_dim=env.state_dim,
                        action_dim=1 if env.if_discrete else env.action_dim, if_on_policy=on_policy, if_per=if_per,
                        if_gpu=True, device=agent.device)


def _agree(dp, flag):
    """The loop-exit decision must be the SAME on every data-parallel rank (a rank that leaves alone strands the others in
    their next all-reduce): any rank's reason to stop -- goal reached on its lanes, its view of the `stop` file -- stops all."""
    if dp is None:
        return bool(flag)
    return dp.max_over_ranks(1.0 if flag else 0.0) > 0.5


def train_and_evaluate(args):
    dp = getattr(args.agent, "dp", None)
    is_main = dp is None or dp.rank == 0          # under data parallelism only rank 0 owns cwd: checkpoints, logs, plots
    args.init_before_training(if_main=is_main)
    cwd, env, agent = args.cwd, args.env, args.agent
    env_eval = args.env_eval if args.env_eval is not None else (env if hasattr(env, "num_envs") else deepcopy(env))
    if hasattr(env, "num_envs") and env_eval is env and not getattr(args.agent, "if_on_policy", False):
        # An off-policy agent CONTINUES the running episodes of its env from one explore call to the next, and the evaluator
        # resets the env it is given: on a shared vectorised env every evaluation would cut the episodes (the reference shares
        # the env too, train.py:227-228, and stores that broken transition once per evaluation).  Evaluate on a clone.
        env_eval = env.clone()

    if "integrator_dim" in args.Modular_kwargs:
        agent.init(args.net_dim, env.state_dim, env.action_dim, args.Modular_kwargs["integrator_dim"], args.if_per)
    else:
        agent.init(args.net_dim, env.state_dim, env.action_dim, args.if_per)
    if len(args.SCN_kwargs) > 0:
        agent.init_SCN(args.SCN_kwargs)
    if len(args.residual_kwargs) > 0:
        agent.init_residual(args.residual_kwargs)
    if len(args.Q_kwargs) > 0:
        agent.init_Q(args.Q_kwargs)
    if args.if_residual:
        agent.init_actor_zero()      # the reference calls it a second time here (run.py:148-150)
    if args.fix_K:
        agent.fix_K()
    if args.frozen_modular_integrator:
        agent.frozen_integrator()
        print("frozen_modular_integrator!=================")
    if args.frozen_transfer:
        agent.frozen_transfer()
        print("frozen_transfer!===========================")
    if args.load != "None":
        agent.save_load_model(args.load, if_save=False)
    if dp is not None:
        dp.broadcast_module(agent.act, agent.cri)
        # identical replicas, DIFFERENT exploration / minibatch streams: init_before_training seeded torch alike on every rank
        torch.manual_seed(args.random_seed + dp.rank)
        if not is_main:
            logger.configure(folder=None)

    if_on_policy = getattr(agent, "if_on_policy", False)
    buffer = make_buffer(agent, env, args.max_memo, args.if_per)
    evaluator = Evaluator(cwd=cwd, agent_id=args.gpu_id, device=agent.device, env=env_eval, eval_gap=args.eval_gap,
                          eval_times1=args.eval_times1, eval_times2=args.eval_times2, is_main=is_main)
    if_reach_goal = evaluator.evaluate_act(agent)
    logger.dump(step=0)

    agent.state = None if hasattr(env, "num_envs") else env.reset()
    total_step = 0
    if args.test_render is not None and is_main:
        save_path = os.path.join(cwd, "step_0")
        os.makedirs(save_path, exist_ok=True)
        args.test_render(agent, save_path)

    if not if_on_policy:
        while args.learning_start > 0 and total_step < args.learning_start:
            total_step += agent.explore_env(env, buffer, args.target_step, args.reward_scale, args.gamma)
            logger.record("training/total_step", total_step)
            logger.dump(step=total_step)

    while not _agree(dp, (args.if_allow_break and if_reach_goal) or total_step >= args.break_step
                     or os.path.exists(f"{cwd}/stop")):
        t0 = time.time()
        steps = agent.explore_env(env, buffer, args.target_step, args.reward_scale, args.gamma)
        total_step += steps
        obj_a, obj_c = agent.update_net(buffer, args.target_step, args.batch_size, args.repeat_times)
        logger.record("perf/env_steps_per_s", steps / max(time.time() - t0, 1e-9))
        if_reach_goal = evaluator.evaluate_save(agent, steps, obj_a, obj_c)
        if args.test_render is not None and is_main and total_step % args.test_render_times == 0:
            save_path = os.path.join(cwd, f"step_{total_step}")
            os.makedirs(save_path, exist_ok=True)
            args.test_render(agent, save_path)
        logger.record("training/total_step", total_step)
        logger.dump(step=total_step)
    if is_main:
        print(f"| SavedDir: {cwd}\n| UsedTime: {time.time() - evaluator.start_time:.0f}")
    return agent, buffer


class Evaluator:
    def __init__(self, cwd, agent_id, eval_times1, eval_times2, eval_gap, env, device, is_main=True):
        self.is_main = is_main   # False on data-parallel ranks > 0: evaluate (same call sequence on the shared env) but write nothing
        self.recorder = [(0., -np.inf, 0., 0., 0.)]  # total_step, r_avg, r_std, obj_a, obj_c
        self.r_max = -np.inf
        self.total_step = 0
        self.cwd, self.device, self.agent_id = cwd, device, agent_id
        self.eval_gap, self.eval_times1, self.eval_times2 = eval_gap, eval_times1, eval_times2
        self.env = env
        self.target_return = env.target_return
        self.used_time = None
        self.start_time = time.time()
        self.eval_func_time = 1
        if is_main:
            print(f"{'ID':>2}  {'Step':>8}  {'MaxR':>8} |{'avgR':>8}  {'stdR':>8}   {'objA':>8}  {'objC':>8} |")

    @staticmethod
    def _policy(agent):
        """What maps an observation to the deterministic env action: the actor module itself (its forward adds the prior term
        for the residual PPO actors), or the agent's `eval_policy` where the actor's forward is the residual action alone."""
        return getattr(agent, "eval_policy", None) or agent.act

    def _returns(self, act, times, agent=None):
        if hasattr(self.env, "num_envs"):  # one launch evaluates num_envs episodes at once
            fused = agent.fused_eval_policy(self.env) if hasattr(agent, "fused_eval_policy") else None
            r = []
            while len(r) < times:
                r.extend(get_episode_return_vec(self.env, act, fused=fused).tolist())
            return np.asarray(r[:max(times, 1)])
        return np.asarray([get_episode_return(self.env, act, self.device)[0] for _ in range(times)])

    def evaluate_act(self, agent):
        if self.eval_times1 == 0:
            return False
        r = self._returns(self._policy(agent), self.eval_times1, agent)
        r_avg, r_std = float(r.mean()), float(r.std())
        if r_avg > self.r_max:
            self.r_max = r_avg
            if self.is_main:
                agent.save_load_model(self.cwd, if_save=True)
        logger.record("rollout/ep_rew_mean", r_avg)
        logger.record("rollout/ep_rew_std", r_std)
        logger.record("rollout/log_rew_max", self.r_max)
        if self.is_main:
            os.makedirs(os.path.join(self.cwd, "init"), exist_ok=True)
            agent.save_load_model(os.path.join(self.cwd, "init"), if_save=True)
            print(f"{self.agent_id:<2}  {self.total_step:8.2e}  {self.r_max:8.2f} |{r_avg:8.2f}  {r_std:8.2f}")
        self.recorder.append((self.total_step, r_avg, r_std, 0., 0.))
        return bool(self.r_max > self.target_return)

    def evaluate_save(self, agent, steps, obj_a, obj_c):
        if self.eval_times1 == 0:
            return False
        self.total_step += steps
        if_reach_goal = False
        if self.eval_func_time % self.eval_gap == 0:
            r = self._returns(self._policy(agent), self.eval_times1, agent)
            if r.mean() > self.r_max:  # confirm a new best with more episodes before saving
                r = np.concatenate([r, self._returns(self._policy(agent), max(self.eval_times2 - self.eval_times1, 0), agent)]) \
                    if self.eval_times2 > self.eval_times1 else r
            r_avg, r_std = float(r.mean()), float(r.std())
            if r_avg > self.r_max:
                self.r_max = r_avg
                if self.is_main:
                    agent.save_load_model(self.cwd, if_save=True)
                    print(f"{self.agent_id:<2}  {self.total_step:8.2e}  {self.r_max:8.2f} |")
            logger.record("rollout/ep_rew_mean", r_avg)
            logger.record("rollout/ep_rew_std", r_std)
            logger.record("rollout/log_rew_max", self.r_max)
            self.recorder.append((self.total_step, r_avg, r_std, obj_a, obj_c))
            if_reach_goal = bool(self.r_max > self.target_return)
            if if_reach_goal and self.used_time is None:
                self.used_time = int(time.time() - self.start_time)
                if self.is_main:
                    print(f"{self.agent_id:<2}  {self.total_step:8.2e}  {self.target_return:8.2f} |"
                      f"{r_avg:8.2f}  {r_std:8.2f}   {self.used_time:>8}  ########")
        self.eval_func_time += 1
        return if_reach_goal

    def draw_plot(self):
        np.save(f"{self.cwd}/recorder.npy", self.recorder)


def get_episode_return(env, act, device):
    """One deterministic episode on a one-instance env (run.py:600-619)."""
    episode_return = 0.0
    state = env.reset()
    step = 0
    for step in range(env.max_step):
        s = torch.as_tensor(np.asarray(state)[None], device=device)
        with torch.no_grad():
            a = act(s)
        if env.if_discrete:
            a = a.argmax(dim=1)
        state, reward, done, _ = env.step(a.cpu().numpy()[0])
        episode_return += reward
        if done:
            break
    return getattr(env, "episode_return", episode_return), step + 1


def get_episode_return_vec(env, act, fused=None):
    """One deterministic episode on every lane of a vectorised env; returns the per-lane undiscounted returns.
    fused = (packed actor, priorK) from agent.fused_eval_policy(env): reset + ONE launch for the whole episode
    (csrc/rollout_eval.hip) and one device-to-host copy, instead of max_step x [policy forward, env step] launches."""
    if fused is not None:
        env.reset()
        ret, _ = env.rollout_eval(fused[0], fused[1], env.max_step)
        return ret.cpu().numpy()
    obs = env.reset()
    ret = torch.zeros(env.num_envs, dtype=torch.float64, device=obs.device)
    for _ in range(env.max_step):
        with torch.no_grad():
            a = act(obs)
        obs, rew, done = env.step(a, auto_reset=False)
        ret += rew.double()
    return ret.cpu().numpy()
