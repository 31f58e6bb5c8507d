#!/bin/bash
# quick TD3 kernel iteration: parity, phase trace, step timing, kernel list
OUT=gpurun_out; TAG=${1:-td3q}
mkdir -p $OUT
timeout -k 10 300 python -m pytest tests/test_gpu_td3_fused.py -q -x > $OUT/${TAG}_pytest.log 2>&1
rc=$?; tail -3 $OUT/${TAG}_pytest.log
if [ $rc -ne 0 ]; then tail -40 $OUT/${TAG}_pytest.log; exit 1; fi
PIME_TD3_TRACE=1 timeout -k 10 120 python tools/td3_trace.py 2>&1 | grep "td3 trace" | tail -4 > $OUT/${TAG}_trace.txt || exit 1
cat $OUT/${TAG}_trace.txt
timeout -k 10 120 python tools/td3_trace.py > $OUT/${TAG}_steps.txt 2>&1 || { tail -20 $OUT/${TAG}_steps.txt; exit 1; }
cat $OUT/${TAG}_steps.txt
timeout -k 10 200 bash tools/kstats.sh tools/td3_trace.py > $OUT/${TAG}_kstats.txt 2>&1 || { tail -20 $OUT/${TAG}_kstats.txt; exit 1; }
head -5 $OUT/${TAG}_kstats.txt
