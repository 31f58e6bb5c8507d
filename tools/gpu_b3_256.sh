#!/bin/bash
# width 256 (the reference's live water-tank script) with and without the bf16x3 gradient layers
set -o pipefail
o=gpurun_out/${1:-r04k}
mkdir -p $o
python bench.py --workload wt256 --no-cpu-baseline > $o/bench_wt256_f32.json 2> $o/bench_wt256_f32.err &&
PIME_GRAD_BF16X3=1 python bench.py --workload wt256 --no-cpu-baseline > $o/bench_wt256_b3.json 2> $o/bench_wt256_b3.err &&
python bench.py --workload wtmod256 --no-cpu-baseline > $o/bench_wtmod256_f32.json 2> $o/bench_wtmod256_f32.err &&
PIME_GRAD_BF16X3=1 python bench.py --workload wtmod256 --no-cpu-baseline > $o/bench_wtmod256_b3.json 2> $o/bench_wtmod256_b3.err
rc=$?
for f in wt256_f32 wt256_b3 wtmod256_f32 wtmod256_b3; do python3 -c "
import json
d=json.loads(open('$o/bench_$f.json').read().strip().splitlines()[-1]); r=d.get('roofline',{}); print('$f', round(d['value']/1e6,3), round(d['ms_per_step'],2), r.get('frac'), r.get('avg_launch_ms'))"; done
exit $rc
