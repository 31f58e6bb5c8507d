"""Phase marks of workgroup 0 of the two TD3 gradient kernels (PIME_TD3_TRACE=1 python tools/td3_trace.py [width] [batch]) and
event-timed optimizer steps on a synthetic replay buffer (no env): the tuning loop of csrc/td3_fused.hip."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pime_amd.elegantrl.agent import AgentTD3  # noqa: E402

md = int(sys.argv[1]) if len(sys.argv) > 1 else 128
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
dev, D, rows, steps = "cuda:0", 4, 2 ** 20, 200
torch.manual_seed(0)
ag = AgentTD3(device=dev)
ag.init(md, D, 1)
f = ag._fused_step(B)
state = torch.randn(rows, D, device=dev) * 3 + 5
other = torch.stack([-torch.rand(rows, device=dev) * 5, torch.full((rows,), 0.99, device=dev), torch.tanh(torch.randn(rows, device=dev))], 1).contiguous()
idx = torch.randint(rows - 1, (steps, B), device=dev)
nxt = idx + 1


def run(n):
    for k in range(n):
        f.step(state, other, idx, nxt, None, ag.soft_update_tau, ag.update_freq, ag.policy_noise, noise_seed=7, row=k)


run(3)
torch.cuda.synchronize()
if os.environ.get("PIME_TD3_TRACE"):
    sys.exit(0)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    run(steps)
for _ in range(3):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); g.replay(); e.record()
    torch.cuda.synchronize()
    print(f"width {md} batch {B}: {s.elapsed_time(e) / steps * 1e3:.1f} us per optimizer step (graph of {steps})")
