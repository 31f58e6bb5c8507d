"""The data-parallel update path on the GPU with ONE rank (RCCL communicator of size 1): the two-graph step sequence with
the flat-gradient all-reduce (RCCL AVG) in between must give the weights of the single-graph path (up to the float64
advantage normalisation the data-parallel path uses for its buffer-global moments: ulp-level differences).  Runs in a
child process so that the process group does not leak into the rest of the suite.  (World sizes > 1 are covered on the
CPU with gloo, tests/test_dist_gloo.py; real multi-GPU runs are the driver's.)"""
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
from pime_amd.elegantrl.run import make_buffer
from pime_amd.utils import MODELS

def run(dp):
    env = gym_control.make_vec(gym_control.PH_V35, 2048, device="cuda:0", seed=3)
    torch.manual_seed(0)
    agent = MODELS["residualintegratormodularppo"](device="cuda:0")
    agent.init(128, env.state_dim, 1, env.n_integrator)
    agent.init_residual({"init_K": env.K.reshape(-1, 1)})
    agent.init_actor_zero()
    agent.dp = dp
    buf = make_buffer(agent, env, 2048 * env.max_step)
    steps = agent.explore_env(env, buf, 2048 * env.max_step, 1.0, 0.99)
    torch.manual_seed(1)                      # same minibatch table in both runs
    agent.update_net(buf, steps, 8192, 2.0)   # 25 optimizer steps: eager, capture, replay
    torch.cuda.synchronize()
    out = torch.cat([p.detach().reshape(-1) for p in list(agent.act.parameters()) + list(agent.cri.parameters())]).clone()
    env.close()
    return out

single = run(None)
dp = pdist.init_from_env(backend="nccl", device="cuda:0")
assert dp is not None and dp.world == 1
multi = run(dp)
torch.distributed.destroy_process_group()
assert torch.isfinite(single).all() and not torch.equal(single, torch.zeros_like(single))
assert torch.allclose(single, multi, rtol=1e-4, atol=1e-6), float((single - multi).abs().max())
print("DP_SINGLE_RANK_OK")
'''


def test_dp_update_path_equals_single_gpu_path():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as sock:   # a free rendezvous port, not a fixed one another run may hold
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ, PIME_ROOT=root, PIME_FORCE_DP="1", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", _CHILD], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "DP_SINGLE_RANK_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
