#!/usr/bin/env python3
"""One minibatch gradient (B = 65536, state_dim 3) timed with HIP events for a (critic, actor) pair:
   usage: grad_ab.py <actor: modular|resid> <width> [reps]
Run under PIME_MLP16=1 to route the critic / plain actor through the 16-tile family (csrc/mlp16.hip); use
tools/kstats.sh tools/grad_ab.py ... for the per-kernel split."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from pime_amd import ops  # noqa: E402
from pime_amd.elegantrl.net import CriticAdv  # noqa: E402
from pime_amd.elegantrl.net_residual import ActorResidualIntegratorModularPPO, ActorResidualPPO  # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else "modular"
md = int(sys.argv[2]) if len(sys.argv) > 2 else 128
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 50
D = int(os.environ.get("GRAD_AB_D", "3"))
DEV = "cuda:0"
torch.manual_seed(0)
cri = CriticAdv(D, md).to(DEV)
act = (ActorResidualIntegratorModularPPO(md, D, 1, 1) if kind == "modular" else ActorResidualPPO(md, D, 1)).to(DEV)
L, B = 819200, int(os.environ.get("GRAD_AB_B", "65536"))
state = torch.randn(L, D, device=DEV) * 3 + 5
action = torch.randn(L, device=DEV)
lp = torch.randn(L, device=DEV) * 0.1 - 1
adv = torch.randn(L, device=DEV)
rs = torch.randn(L, device=DEV) * 10
f = ops.FusedPPOGrad(act, cri, B)
scale = torch.ones(1, device=DEV)
idx = torch.randint(L, (B,), device=DEV)
for _ in range(5):
    f(state, action, lp, adv, rs, idx, 0.2, 0.02, scale, overwrite=True)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(reps):
    f(state, action, lp, adv, rs, idx, 0.2, 0.02, scale, overwrite=True)
e.record()
torch.cuda.synchronize()
us = s.elapsed_time(e) / reps * 1e3
flops = 3 * 2 * (sum(p.numel() for p in act.parameters() if p.dim() == 2) + sum(p.numel() for p in cri.parameters() if p.dim() == 2)) * B
print(f"actor={kind} width={md} D={D} MLP16={os.environ.get('PIME_MLP16', '0')}: {us:.1f} us per minibatch gradient "
      f"= {flops / us / 1e6:.1f} TFLOP/s ({flops / us / 1e6 / 157.3:.3f} of f32 MFMA peak)")
