#!/bin/bash
# TD3 iteration on the GPU box: parity of the fused optimizer step, then the config-2 bench and its kernel list
OUT=gpurun_out; TAG=${1:-td3}
mkdir -p $OUT
timeout -k 10 500 python -m pytest tests/test_gpu_td3_fused.py tests/test_gpu_td3.py -q -x > $OUT/${TAG}_pytest.log 2>&1
rc=$?; tail -15 $OUT/${TAG}_pytest.log
if [ $rc -ne 0 ]; then exit 1; fi
timeout -k 10 300 python bench.py --workload wt_td3 --steps 5 --warmup 2 > $OUT/${TAG}_bench_wt_td3.json 2> $OUT/${TAG}_bench_wt_td3.err || { tail -20 $OUT/${TAG}_bench_wt_td3.err; exit 1; }
cat $OUT/${TAG}_bench_wt_td3.json
timeout -k 10 300 bash tools/kstats.sh bench.py --workload wt_td3 --steps 3 --warmup 2 > $OUT/${TAG}_kstats.txt 2>&1 || { tail -20 $OUT/${TAG}_kstats.txt; exit 1; }
cat $OUT/${TAG}_kstats.txt
