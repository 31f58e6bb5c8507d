"""BASELINE config 4's whole lane count (131 072 pH envs, sharded over 8 GPUs there) on ONE MI355X: one rollout + one update_net
(800 optimizer steps), finite losses, env-steps/s.  A robustness check of the sizes, not a headline number."""
import sys
import time

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from pime_amd import gym_control
from pime_amd.elegantrl.agent_residual import AgentResidualIntegratorModularPPO
from pime_amd.elegantrl.run import make_buffer

lanes = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
dev = "cuda:0"
env = gym_control.make_vec(gym_control.PH_V35, lanes, device=dev, state_mode="mixed", seed=0)
torch.manual_seed(0)
ag = AgentResidualIntegratorModularPPO(device=dev)
ag.init(128, env.state_dim, 1, env.n_integrator)
ag.init_residual({"init_K": env.K.reshape(-1, 1)})
ag.init_actor_zero()
ag.fix_K()
buf = make_buffer(ag, env, lanes * env.max_step)
for it in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = ag.explore_env(env, buf, lanes * env.max_step, 1.0, 0.99)
    oa, oc = ag.update_net(buf, n, 65536, 8)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert oa == oa and oc == oc, "non-finite loss"
    print(f"iter {it}: {n} env-steps + {int(8 * n / 65536)} optimizer steps in {dt * 1e3:.1f} ms = {n / dt / 1e6:.2f} M env-steps/s  "
          f"obj_a {oa:.4f} obj_c {oc:.3f}  peak memory {torch.cuda.max_memory_allocated() / 2**30:.2f} GiB")
