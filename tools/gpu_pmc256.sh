#!/bin/bash
# matrix-pipe counters (tools/pmc_pipe.sh) of the width-256 gradient kernels, f32 path and bf16x3 variant
o=gpurun_out/${1:-r04m}; mkdir -p $o
export GRAD_AB_D=30
bash tools/pmc_pipe.sh $o/wt256_pmc_pipe_f32.json tools/grad_ab.py resid 256 5 > $o/pmc_f32.log 2>&1 || exit 1
PIME_GRAD_BF16X3=1 bash tools/pmc_pipe.sh $o/wt256_pmc_pipe_bf16x3.json tools/grad_ab.py resid 256 5 > $o/pmc_b3.log 2>&1 || exit 1
python3 - $o <<'PY'
import json, sys
for f in ("f32", "bf16x3"):
    d = json.load(open(f"{sys.argv[1]}/wt256_pmc_pipe_{f}.json"))
    for k, v in d.items():
        if "ppo16" in k:
            print(f, k[:48], {a: (round(b, 3) if isinstance(b, float) else b) for a, b in v.items()})
PY
