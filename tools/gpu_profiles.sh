#!/bin/bash
# usage: tools/gpu_profiles.sh <tag>  -- the round's evidence: bench lines, rocprofv3 kernel stats, PMC traffic, env PMC sweep
TAG=$1; OUT=gpurun_out; mkdir -p $OUT
timeout -k 10 200 python bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err || exit 1
tail -c 200 $OUT/${TAG}_bench.json
PIME_FORCE_DP=1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29517 timeout -k 10 200 python bench.py --no-cpu-baseline > $OUT/${TAG}_bench_dp.json 2> $OUT/${TAG}_bench_dp.err || exit 1
python -c "import json,sys; d=json.loads(open('$OUT/${TAG}_bench_dp.json').read().strip().splitlines()[-1]); print('forced-DP (1 rank):', d['value'])"
grep -i "refused\|graph" $OUT/${TAG}_bench_dp.err | head -3
timeout -k 10 200 python bench.py --workload wt --no-cpu-baseline > $OUT/${TAG}_bench_water_tank.json 2>/dev/null || exit 1
for w in wt_td3 wt256 wtmod256 mixed16; do   # the other workloads (BASELINE configs 2 / 5, the reference script's width 256)
  timeout -k 10 300 python bench.py --workload $w --no-cpu-baseline > $OUT/${TAG}_bench_$w.json 2>/dev/null || exit 1
  python -c "import json; d=json.load(open('$OUT/${TAG}_bench_$w.json')); print('$w', round(d['value'] / 1e6, 2), 'M env-steps/s')"
done
timeout -k 10 300 bash tools/profile_bench.sh $TAG || exit 1
timeout -k 10 400 bash tools/pmc_traffic.sh $OUT/${TAG}_pmc_hbm_traffic.json bench.py --steps 2 --warmup 1 --no-cpu-baseline || exit 1
timeout -k 10 400 bash tools/env_pmc.sh $OUT/${TAG}_env_pmc.json || exit 1
# the TD3 step's kernels (BASELINE config 2): per-kernel stats and HBM traffic of the same bench command
timeout -k 10 300 bash tools/profile_bench.sh ${TAG}_td3 --workload wt_td3 || exit 1
timeout -k 10 400 bash tools/pmc_traffic.sh $OUT/${TAG}_td3_hbm_traffic_pmc.json bench.py --workload wt_td3 --steps 2 --warmup 1 --no-cpu-baseline || exit 1
timeout -k 10 300 bash tools/profile_bench.sh ${TAG}_wt256 --workload wt256 || exit 1
# the opt-in bf16x3 variant (PIME_GRAD_BF16X3=1): width 256 (this family by default), the headline shape through the 16-tile family
# with and without it, per-kernel stats of the width-256 run, and the two in-place layer / weight-gradient microbenchmarks
for w in wt256 wtmod256; do
  PIME_GRAD_BF16X3=1 timeout -k 10 300 python bench.py --workload $w --no-cpu-baseline > $OUT/${TAG}_bench_${w}_bf16x3.json 2>/dev/null || exit 1
  python -c "import json; d=json.load(open('$OUT/${TAG}_bench_${w}_bf16x3.json')); print('$w bf16x3', round(d['value'] / 1e6, 2), 'M env-steps/s, frac', round(d['roofline']['frac'], 3))"
done
PIME_MLP16=1 timeout -k 10 300 python bench.py --no-cpu-baseline > $OUT/${TAG}_bench_ph_mlp16.json 2>/dev/null || exit 1
PIME_MLP16=1 PIME_GRAD_BF16X3=1 timeout -k 10 300 python bench.py --no-cpu-baseline > $OUT/${TAG}_bench_bf16x3.json 2>/dev/null || exit 1
python -c "
import json
for f in ('ph_mlp16', 'bf16x3'):
    d = json.load(open('$OUT/${TAG}_bench_' + f + '.json')); print(f, round(d['value'] / 1e6, 2), 'M env-steps/s, minibatch gradient', round(d['roofline']['avg_launch_ms'] * 1e3, 1), 'us')"
PIME_GRAD_BF16X3=1 timeout -k 10 300 bash tools/profile_bench.sh ${TAG}_wt256_bf16x3 --workload wt256 || exit 1
if [ -x tools/bin/layer16_b3_bench ]; then
  timeout -k 10 120 ./tools/bin/layer16_b3_bench > $OUT/${TAG}_layer16_b3_bench.txt 2>&1 || exit 1
  timeout -k 10 120 ./tools/bin/dw16_b3_bench > $OUT/${TAG}_dw16_b3_bench.txt 2>&1 || exit 1
  tail -1 $OUT/${TAG}_layer16_b3_bench.txt; tail -1 $OUT/${TAG}_dw16_b3_bench.txt
fi
