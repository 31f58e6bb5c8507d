#!/bin/bash
# A/B of the TD3 gradient kernels' waves per workgroup (PIME_TD3_WAVES=4|8): parity with 8, phase traces, step timing, bench lines interleaved
OUT=gpurun_out; TAG=${1:-td3w}
mkdir -p $OUT
PIME_TD3_WAVES=8 timeout -k 10 400 python -m pytest tests/test_gpu_td3_fused.py tests/test_gpu_td3.py -q -x > $OUT/${TAG}_pytest_w8.log 2>&1
rc=$?; tail -3 $OUT/${TAG}_pytest_w8.log
if [ $rc -ne 0 ]; then tail -40 $OUT/${TAG}_pytest_w8.log; exit 1; fi
for w in 4 8; do
  PIME_TD3_WAVES=$w PIME_TD3_TRACE=1 timeout -k 10 120 python tools/td3_trace.py 2>&1 | grep "td3 trace" | tail -4 > $OUT/${TAG}_trace_w$w.txt || exit 1
  echo "waves $w"; cat $OUT/${TAG}_trace_w$w.txt
done
for rep in 1 2; do
  for w in 4 8; do
    PIME_TD3_WAVES=$w timeout -k 10 200 python bench.py --workload wt_td3 --steps 5 --warmup 2 --no-cpu-baseline > $OUT/${TAG}_bench_w${w}_$rep.json 2> $OUT/${TAG}_bench_w${w}_$rep.err || { tail -20 $OUT/${TAG}_bench_w${w}_$rep.err; exit 1; }
    python -c "import json,sys; d=json.load(open('$OUT/${TAG}_bench_w${w}_$rep.json')); print('waves $w rep $rep:', round(d['value']/1e6,2), 'M', round(d['roofline']['avg_launch_ms']*1e3,2), 'us/step')"
  done
done
for w in 4 8; do
  PIME_TD3_WAVES=$w timeout -k 10 200 bash tools/kstats.sh bench.py --workload wt_td3 --steps 3 --warmup 2 --no-cpu-baseline > $OUT/${TAG}_kstats_w$w.txt 2>&1 || { tail -20 $OUT/${TAG}_kstats_w$w.txt; exit 1; }
  echo "waves $w"; head -6 $OUT/${TAG}_kstats_w$w.txt
done
