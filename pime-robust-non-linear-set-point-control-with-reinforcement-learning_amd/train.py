"""Command-line driver with the reference's flags (/root/reference/train.py:23-78) plus the vectorisation ones.

    python -m pime_amd.train --algo ResidualIntegratorModularPPO --fix_K \\
        --env PH1DChangingParamUniformGoalIntegrator-SqaureDistance-v35 --net_dim 128 \\
        --num_envs 16384 --target_step 819200 --batch_size 65536 --repeat_times 8

With --num_envs 1 (default) it builds the one-instance gym-style env exactly as the reference's train.py does
(gym.make + env.seed + np.random.seed + PreprocessEnv); with --num_envs N > 1 it builds the vectorised env and, when
launched under torchrun, shards N lanes per rank with one RCCL gradient all-reduce per optimizer step.

Reference quirks kept by default: --gamma and --learning_rate are parsed but NOT forwarded
(effective 0.99 / 1e-4); water-tank envs are made with reward_type=args.reward_type and r=args.goal.
`--no_reference_quirks` forwards gamma / learning_rate."""
import argparse
import os
import time

import numpy as np
import torch

from . import dist as pdist
from . import gym_compat as gym
from . import gym_control
from .elegantrl.env import PreprocessEnv
from .elegantrl.run import Arguments, train_and_evaluate
from .elegantrl.utils import configure_logger
from .utils import IF_ONPOLICY, MODELS


def build_parser():
    p = argparse.ArgumentParser()
    p.add_argument("--algo", default="PPO", type=str)
    p.add_argument("--env", default=gym_control.WT_INTEGRATOR, type=str)
    p.add_argument("--reward_type", default="distance", type=str, choices=["distance", "sparse"])
    p.add_argument("--fix_K", action="store_true")
    p.add_argument("--seed", default=0, type=int)
    p.add_argument("--target_return", default=1e6, type=float)
    p.add_argument("--target_step", default=2 ** 3, type=int)
    p.add_argument("--reward_scale", default=1., type=float)
    p.add_argument("--break_step", default=2 ** 20, type=int)
    p.add_argument("--learning_start", default=0, type=int)
    p.add_argument("--gamma", default=0.995, type=float)
    p.add_argument("--batch_size", default=2 ** 6, type=int)
    p.add_argument("--learning_rate", default=3e-4, type=float)
    p.add_argument("--buffer_size", default=2 ** 20, type=int)
    p.add_argument("--net_dim", default=2 ** 4, type=int)
    p.add_argument("--verbose", default=0, type=int)
    p.add_argument("--tensorboard_log", default="tensorboard", type=str)
    p.add_argument("--env_zero_noise", action="store_true")
    p.add_argument("--eval_times1", default=2 ** 3, type=int)
    p.add_argument("--eval_times2", default=2 ** 4, type=int)
    p.add_argument("--eval_gap", default=5, type=int)
    p.add_argument("--robust_test", action="store_true")
    p.add_argument("--goal", default=4.0, type=float)
    p.add_argument("--repeat_times", default=2 ** 4, type=int)
    p.add_argument("--lambda_gae_adv", default=0.97, type=float)
    p.add_argument("--lambda_entropy", default=0.02, type=float)
    p.add_argument("--ratio_clip", default=0.2, type=float)
    p.add_argument("--test_render_times", default=10000, type=int)
    p.add_argument("--load", default="None", type=str)
    p.add_argument("--frozen_modular_integrator", action="store_true")
    p.add_argument("--frozen_transfer", action="store_true")
    # --- this build's additions ---
    p.add_argument("--num_envs", default=1, type=int, help="env lanes per GPU (1 = the reference's single instance)")
    p.add_argument("--resample_every", default=1, type=int, help="redraw ensemble params every n episodes (0 = never)")
    p.add_argument("--device", default="cuda", type=str)
    p.add_argument("--state_dtype", default="mixed", choices=["mixed", "f64"])
    p.add_argument("--draws", default="philox", choices=["philox", "mt19937"],
                   help="vectorised envs: in-kernel Philox or per-env MT19937 streams replayed seed-for-seed")
    p.add_argument("--no_reference_quirks", action="store_true")
    p.add_argument("--log_root", default=None, type=str)
    return p


def make_env(args, dp=None):
    wt = "NonLinearWaterTank" in args.env
    overrides = {}
    if wt:
        overrides.update(reward_type=args.reward_type, r=args.goal)   # train.py:98-101
    if args.env_zero_noise:
        overrides["noise_scale"] = 0.
    if args.num_envs > 1:
        overrides.pop("r", None)
        offset = dp.lane_offset(args.num_envs) if dp is not None else 0
        env = gym_control.make_vec(args.env, args.num_envs, device=args.device, state_mode=args.state_dtype,
                                   seed=args.seed, env_offset=offset, draws=args.draws,
                                   resample_every=args.resample_every, **overrides)
        env.env_name = args.env
        env.target_return = args.target_return
        return env
    env = gym.make(args.env, device=args.device, **overrides)
    env.seed(args.seed)
    np.random.seed(args.seed)
    env.target_return = args.target_return
    return env


def prepare_train(args, env):
    algo = args.algo.lower()
    kargs = Arguments(if_on_policy=IF_ONPOLICY[algo])
    kargs.repeat_times = args.repeat_times if IF_ONPOLICY[algo] else 1
    kargs.gpu_id = 0
    kargs.if_remove = False
    kargs.random_seed = args.seed
    kargs.env = PreprocessEnv(env=env)
    kargs.env_eval = PreprocessEnv(env=env)   # the SAME env object, as in the reference (train.py:227-228)
    for k in ("reward_scale", "net_dim", "batch_size", "break_step", "learning_start", "eval_times1", "eval_times2",
              "eval_gap", "fix_K", "frozen_modular_integrator", "frozen_transfer", "test_render_times", "target_step",
              "load"):
        setattr(kargs, k, getattr(args, k))
    if IF_ONPOLICY[algo]:
        kargs.max_memo = args.target_step
    kargs.agent = MODELS[algo](device=args.device)
    if "ppo" in algo:
        kargs.agent.lambda_entropy = args.lambda_entropy
        kargs.agent.ratio_clip = args.ratio_clip
        kargs.agent.lambda_gae_adv = args.lambda_gae_adv
    if args.no_reference_quirks:
        kargs.gamma = args.gamma
        kargs.agent.learning_rate = args.learning_rate
    root = args.log_root or f"log_{args.break_step}"
    tag = f"{args.algo}{'-sparse' if args.reward_type == 'sparse' else ''}-{args.net_dim}{'-fixK' if args.fix_K else ''}"
    zero = "-zero" if args.env_zero_noise else ""
    tb = os.path.join(root, f"{args.tensorboard_log}_{args.env}{zero}/")
    configure_logger(args.verbose, tb, tag, True)
    # data-parallel ranks must agree on the run directory (they all watch its `stop` file): rank 0's clock names it
    stamp = time.strftime("%Y-%m-%d-%H_%M_%S", time.localtime(pdist.broadcast_scalar(time.time())))
    kargs.cwd = os.path.join(root, f"{args.env}{zero}/{tag}/seed{args.seed}/{stamp}")
    return kargs, tb


def main(argv=None):
    args = build_parser().parse_args(argv)
    assert args.test_render_times % args.target_step == 0 or args.num_envs > 1, "Must be an integer multiple"
    dp = pdist.init_from_env(device=args.device) if args.num_envs > 1 else None
    if dp is not None and args.device.startswith("cuda"):
        args.device = f"cuda:{dp.local_rank}"
        torch.cuda.set_device(dp.local_rank)
    print(f"Agent: {args.algo}, Env: {args.env}, Seed: {args.seed}, lanes/GPU: {args.num_envs}")
    env = make_env(args, dp)
    kargs, _ = prepare_train(args, env)
    kargs.agent.dp = dp
    algo = args.algo.lower()
    K = getattr(env, "K", None)
    kargs.residual_kwargs = {"init_K": np.asarray(K).reshape(-1, 1)} if ("residual" in algo and K is not None) else {}
    kargs.Modular_kwargs = {"integrator_dim": env.n_integrator} if ("modular" in algo and hasattr(env, "n_integrator")) else {}
    kargs.if_residual = hasattr(kargs.agent, "init_actor_zero")   # the reference sets True and crashes for TD3
    agent, _ = train_and_evaluate(kargs)
    if dp is None or dp.rank == 0:   # the replicas are identical: rank 0 alone writes the run directory
        save_dir = os.path.join(kargs.cwd, "final_model")
        os.makedirs(save_dir, exist_ok=True)
        agent.save_load_model(save_dir, if_save=True)
        with open(os.path.join(kargs.cwd, "args.txt"), "w") as f:
            f.write(str(args))
        print(f"Finish Training and Saved in {kargs.cwd}")
    return agent


if __name__ == "__main__":
    main()
