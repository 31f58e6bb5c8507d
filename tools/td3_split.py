#!/usr/bin/env python3
"""Where bench.py --workload wt_td3 spends its step: explore (200 lock-steps) vs update (200 optimizer steps), wall clock with
synchronisation in between.  usage: python tools/td3_split.py [steps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from pime_amd import gym_control  # noqa: E402
from pime_amd.elegantrl.agent_residual import AgentResidualTD3  # noqa: E402
from pime_amd.elegantrl.run import make_buffer  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
lanes, T, batch = 4096, 200, 4096
env = gym_control.make_vec(gym_control.WT_INTEGRATOR, lanes, device="cuda:0", state_mode="mixed", seed=0, reward_type="distance")
torch.manual_seed(0)
agent = AgentResidualTD3(device="cuda:0")
agent.init(128, env.state_dim, 1)
agent.init_residual({"init_K": env.K.reshape(-1, 1)})
buf = make_buffer(agent, env, 2 ** 21)
for it in range(steps + 2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = agent.explore_env(env, buf, lanes * T, 1.0, 0.99)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    agent.update_net(buf, lanes * T, batch, 1)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    if it >= 2:
        print(f"iter {it}: explore {1e3 * (t1 - t0):.1f} ms, update {1e3 * (t2 - t1):.1f} ms -> {n / (t2 - t0) / 1e6:.2f} M env-steps/s")
