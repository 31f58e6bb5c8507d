"""Child process of tests/test_gpu_oneshot_allreduce.py::test_data_parallel_update_with_the_oneshot_allreduce: one data-parallel
rank (both ranks on cuda:0, gloo for the small collectives) running rollout + update_net of the bench's agent on its lane slice;
the flat-gradient all-reduce of every optimizer step is the one-shot kernel when PIME_ONESHOT_ALLREDUCE=1, gloo otherwise."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from pime_amd import dist as pdist  # noqa: E402
from pime_amd import gym_control  # noqa: E402
from pime_amd.elegantrl.agent_residual import AgentResidualIntegratorModularPPO  # noqa: E402
from pime_amd.elegantrl.run import make_buffer  # noqa: E402

out = sys.argv[1]
dev = "cuda:0"
torch.cuda.set_device(0)
dp = pdist.init_from_env(backend="gloo", device=dev)
N = 2048
env = gym_control.make_vec(gym_control.PH_V35, N, device=dev, state_mode="mixed", seed=3, env_offset=dp.lane_offset(N))
torch.manual_seed(0)
ag = AgentResidualIntegratorModularPPO(device=dev)
ag.init(128, env.state_dim, 1, env.n_integrator)
ag.init_residual({"init_K": env.K.reshape(-1, 1)})
with torch.no_grad():
    ag.act.net[-1].weight.normal_(0, 0.05)
ag.weights_changed()
ag.dp = dp
dp.broadcast_module(ag.act, ag.cri)
torch.manual_seed(100 + dp.rank)
buf = make_buffer(ag, env, N * 50)
objs = []
for it in range(3):      # eager step, per-step graph, whole-update graph
    n = ag.explore_env(env, buf, N * 50, 1.0, 0.99)
    objs.append(ag.update_net(buf, n, 8192, 2))
torch.cuda.synchronize()
fused = ag._packed.get("fused")
ar = next(iter(dp._oneshot.values())) if dp._oneshot else None
torch.save({"flat": fused.flat_param.cpu(), "objs": objs, "oneshot": ar is not None, "status": ar.status() if ar else 0,
            "in_graph": fused.static.graph_full is not None or fused.static.graph_update is not None},
           f"{out}.{dp.rank}.pt")
dp.barrier()
torch.distributed.destroy_process_group()
