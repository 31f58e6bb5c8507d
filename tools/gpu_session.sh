#!/bin/bash
# One GPU-box session: the -m gpu suite, then the A/B timings of the gradient kernels (per-kernel via rocprofv3 --stats).
# usage: tools/gpu_session.sh <tag>     (outputs under gpurun_out/<tag>_*)
TAG=${1:-sess}
OUT=gpurun_out
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -m gpu -q -x > $OUT/${TAG}_pytest.log 2>&1
rc=$?
echo "pytest rc=$rc" >> $OUT/${TAG}_pytest.log
tail -12 $OUT/${TAG}_pytest.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
for cfg in "0 modular 128" "1 modular 128" "0 resid 128" "1 resid 128" "1 resid 256" "1 resid 64" "0 resid 64"; do
  set -- $cfg
  if [ "$1" = "1" ]; then export PIME_MLP16=1; else unset PIME_MLP16; fi
  echo "== MLP16=$1 actor=$2 width=$3" >> $OUT/${TAG}_ab.log
  timeout -k 10 120 bash tools/kstats.sh tools/grad_ab.py $2 $3 >> $OUT/${TAG}_ab.log 2>&1 || exit 1
  grep "us per minibatch" /tmp/kstats.out >> $OUT/${TAG}_ab.log
done
unset PIME_MLP16
cat $OUT/${TAG}_ab.log
