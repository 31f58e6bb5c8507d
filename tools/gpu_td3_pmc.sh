#!/bin/bash
# quick TD3 iteration + matrix-pipe / wait counters of the step's kernels
OUT=gpurun_out; TAG=${1:-td3p}
bash tools/gpu_td3_quick.sh $TAG || exit 1
timeout -k 10 500 bash tools/pmc_pipe.sh $OUT/${TAG}_pmc_pipe.json tools/td3_trace.py > $OUT/${TAG}_pmc_pipe.txt 2>&1 || { tail -20 $OUT/${TAG}_pmc_pipe.txt; exit 1; }
cat $OUT/${TAG}_pmc_pipe.txt
