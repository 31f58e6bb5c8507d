#!/usr/bin/env python3
"""Value-pass forward (819 200 rows, state_dim 3) timed with HIP events: usage fwd_ab.py <width> [reps]; PIME_MLP16=1 routes
widths 64 / 128 through the 16-tile family (csrc/mlp16.hip)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from pime_amd import ops  # noqa: E402
from pime_amd.elegantrl.net import CriticAdv  # noqa: E402

md = int(sys.argv[1]) if len(sys.argv) > 1 else 128
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
D = int(os.environ.get("GRAD_AB_D", "3"))
torch.manual_seed(0)
cri = CriticAdv(D, md).to("cuda:0")
M = 819200
x = torch.randn(M, D, device="cuda:0") * 3 + 5
pk = ops.PackedMLP.from_module(cri)
out = torch.empty(M, device="cuda:0")
for _ in range(3):
    pk(x, out=out)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(reps):
    pk(x, out=out)
e.record()
torch.cuda.synchronize()
us = s.elapsed_time(e) / reps * 1e3
flops = 2 * sum(p.numel() for p in cri.parameters() if p.dim() == 2) * M
print(f"value pass width={md} D={D} MLP16={os.environ.get('PIME_MLP16', '0')}: {us:.1f} us = {flops / us / 1e6:.1f} TFLOP/s "
      f"({flops / us / 1e6 / 157.3:.3f} of f32 MFMA peak)")
