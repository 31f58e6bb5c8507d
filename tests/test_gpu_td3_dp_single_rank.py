"""The fused TD3 step split for data parallelism (pime_td3_step phases 16 / 32 / 64 / 128: slab reduction only, all-reduce of the net's
gradient tensor, Adam + delayed soft update from it) on the GPU with ONE rank -- an RCCL communicator of size 1, whose mean is the
identity: the update must leave all four nets bit-identical to the single-GPU fused update (same kernels, same sums; only the launch
that applies Adam reads the gradient back from memory instead of holding it in registers).  World sizes > 1: the module path's
semantics are pinned on the CPU with gloo (tests/test_dist_gloo.py::test_td3_data_parallel_update_equals_one_rank_on_the_union_minibatch);
real multi-GPU runs are the driver's.  Child process: the process group must not leak into the rest of the suite."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

_CHILD = r'''
import os, sys, torch
sys.path.insert(0, os.environ["PIME_ROOT"])
from pime_amd import dist as pdist, gym_control
from pime_amd.elegantrl.agent_residual import AgentResidualTD3
from pime_amd.elegantrl.replay import VecReplayBuffer

def run(dp):
    N = 512
    env = gym_control.make_vec(gym_control.WT_INTEGRATOR, N, device="cuda:0", seed=5, reward_type="distance", max_step=40)
    torch.manual_seed(0)
    ag = AgentResidualTD3(device="cuda:0")
    ag.init(128, env.state_dim, 1)
    ag.init_residual({"init_K": env.K.reshape(-1, 1)})
    ag.dp = dp
    buf = VecReplayBuffer(64 * N, N, env.state_dim, 1, "cuda:0")
    ag.explore_env(env, buf, 40 * N, 1.0, 0.99)
    torch.manual_seed(1)                       # the same index tables in both runs (the smoothing noise follows torch's initial seed)
    for _ in range(2):                         # the second call replays the captured graph on the single-GPU path
        oa, oc = ag.update_net(buf, 20 * N, 1024, 1.0)
    torch.cuda.synchronize()
    assert ag._fused_td3, "update_net did not take the fused TD3 step"
    out = torch.cat([p.detach().reshape(-1) for m in (ag.act, ag.cri, ag.act_target, ag.cri_target) for p in m.parameters()]).clone()
    env.close()
    return out, (oa, oc)

single, obj1 = run(None)
dp = pdist.init_from_env(backend="nccl", device="cuda:0")
assert dp is not None and dp.world == 1
multi, obj2 = run(dp)
torch.distributed.destroy_process_group()
assert torch.isfinite(single).all() and float(single.abs().max()) > 0
assert torch.equal(single, multi), float((single - multi).abs().max())
assert abs(obj1[0] - obj2[0]) <= 1e-6 * max(1.0, abs(obj1[0])) and abs(obj1[1] - obj2[1]) <= 1e-6 * max(1.0, abs(obj1[1])), (obj1, obj2)
print("TD3_DP_SINGLE_RANK_OK")
'''


def test_fused_td3_dp_phases_equal_the_single_gpu_step():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ, PIME_ROOT=root, PIME_FORCE_DP="1", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", _CHILD], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "TD3_DP_SINGLE_RANK_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
