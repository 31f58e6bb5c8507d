#!/bin/bash
# tools/ab_variants.sh <tag> "<cfg>" v1 v2 ...: time one minibatch gradient with each variants/<v>.so, interleaved, 3 repetitions, same box
TAG=$1; CFG=$2; shift 2
OUT=gpurun_out/${TAG}.log; rm -f $OUT
for rep in 1 2 3; do
  for v in "$@"; do
    r=$(PIME_ALLOW_LIB_OVERRIDE=1 PIME_LIB_PATH=$PWD/variants/$v.so timeout -k 10 120 python tools/grad_ab.py $CFG 2>&1 | grep "us per minibatch") || exit 1
    echo "$v: $r" >> $OUT
  done
done
sort $OUT
