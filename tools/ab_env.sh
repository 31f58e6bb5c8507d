#!/bin/bash
# tools/ab_env.sh <tag> "<ENV=..>" "<ENV=..>" ...: the headline bench (short) under each environment setting ("-" = none), interleaved,
# 3 repetitions, same box, same library
TAG=$1; shift
OUT=gpurun_out/${TAG}.log; rm -f $OUT
for rep in 1 2 3; do
  for e in "$@"; do
    ev=$e; [ "$e" = "-" ] && ev="PIME_NOOP=1"
    r=$(env $ev timeout -k 10 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print(round(d['value']/1e6,3), 'M', round(d['roofline']['avg_launch_ms']*1e3,1), 'us/minibatch')") || exit 1
    echo "$e: $r" >> $OUT
  done
done
sort $OUT
