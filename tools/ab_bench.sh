#!/bin/bash
# tools/ab_bench.sh <tag> v1 v2 ...: the headline bench (short) with each variants/<v>.so, interleaved, 3 repetitions, same box
TAG=$1; shift
OUT=gpurun_out/${TAG}.log; rm -f $OUT
for rep in 1 2 3; do
  for v in "$@"; do
    r=$(PIME_ALLOW_LIB_OVERRIDE=1 PIME_LIB_PATH=$PWD/variants/$v.so timeout -k 10 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print(round(d['value']/1e6,3), 'M', round(d['roofline']['avg_launch_ms']*1e3,1), 'us/minibatch')") || exit 1
    echo "$v: $r" >> $OUT
  done
done
sort $OUT
