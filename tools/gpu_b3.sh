#!/bin/bash
# bf16x3 variant: its tests, then the headline bench three ways (default kernels, 16-tile family f32, 16-tile family bf16x3)
set -o pipefail
out=gpurun_out/${1:-r04k}
mkdir -p $out
python -m pytest tests/test_gpu_bf16x3.py tests/test_gpu_mlp16.py -m gpu -x -q > $out/b3_tests.log 2>&1; rc=$?
tail -15 $out/b3_tests.log
[ $rc -ne 0 ] && exit $rc
python bench.py > $out/bench_default.json 2> $out/bench_default.err &&
PIME_MLP16=1 python bench.py > $out/bench_mlp16.json 2> $out/bench_mlp16.err &&
PIME_MLP16=1 PIME_GRAD_BF16X3=1 python bench.py > $out/bench_b3.json 2> $out/bench_b3.err
for f in default mlp16 b3; do python3 -c "
import json
d=json.loads(open('$out/bench_$f.json').read().strip().splitlines()[-1]); print('$f', round(d['value']/1e6,2), round(d['ms_per_step'],2), round(d['roofline']['frac'],3), round(d['roofline']['avg_launch_ms']*1000,1))"; done
