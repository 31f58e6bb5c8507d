#!/bin/bash
# tools/ab_td3.sh <tag> v1 v2 ...: parity of the in-tree library, then bench.py --workload wt_td3 with each variants/<v>.so, interleaved, same box
TAG=$1; shift
OUT=gpurun_out; mkdir -p $OUT
timeout -k 10 400 python -m pytest tests/test_gpu_td3_fused.py tests/test_gpu_td3.py -q -x > $OUT/${TAG}_pytest.log 2>&1
rc=$?; tail -3 $OUT/${TAG}_pytest.log
if [ $rc -ne 0 ]; then tail -40 $OUT/${TAG}_pytest.log; exit 1; fi
PIME_TD3_TRACE=1 timeout -k 10 120 python tools/td3_trace.py 2>&1 | grep "td3 trace" | tail -2
rm -f $OUT/${TAG}.log
for rep in 1 2 3; do
  for v in "$@"; do
    r=$(PIME_ALLOW_LIB_OVERRIDE=1 PIME_LIB_PATH=$PWD/variants/$v.so timeout -k 10 200 python bench.py --workload wt_td3 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print(round(d['value']/1e6,2), 'M', round(d['roofline']['avg_launch_ms']*1e3,2), 'us/step')") || exit 1
    echo "$v: $r" >> $OUT/${TAG}.log
  done
done
sort $OUT/${TAG}.log
