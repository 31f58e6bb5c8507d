#!/bin/bash
# tools/ab_kstats.sh <tag> "<cfg>" v1 v2 ...: per-kernel average durations (rocprofv3 --kernel-trace --stats) of one minibatch gradient per variant, same box
TAG=$1; CFG=$2; shift 2
OUT=gpurun_out/${TAG}.log; rm -f $OUT
for rep in 1 2; do
  for v in "$@"; do
    echo "== $v (rep $rep)" >> $OUT
    PIME_ALLOW_LIB_OVERRIDE=1 PIME_LIB_PATH=$PWD/variants/$v.so timeout -k 10 200 bash tools/kstats.sh tools/grad_ab.py $CFG 2>&1 | grep "ppo_\|pime::" | head -4 >> $OUT || exit 1
  done
done
cat $OUT
