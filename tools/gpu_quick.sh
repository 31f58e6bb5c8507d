#!/bin/bash
# quick kernel iteration: parity of the 16-tile family, then timings + phase trace
OUT=gpurun_out; TAG=${1:-q}
mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_mlp16.py tests/test_gpu_update_golden.py -q -x > $OUT/${TAG}_pytest.log 2>&1
rc=$?; tail -3 $OUT/${TAG}_pytest.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
export PIME_MLP16=1
for cfg in "resid 128" "resid 256" "resid 64"; do
  timeout -k 10 120 python tools/grad_ab.py $cfg 2>&1 | grep "us per minibatch" >> $OUT/${TAG}_ab.log || exit 1
done
PIME_FUSED_TRACE=0 timeout -k 10 120 python tools/grad_ab.py resid 128 3 2>&1 | grep "pime trace" | tail -2 >> $OUT/${TAG}_ab.log
PIME_FUSED_TRACE=0 timeout -k 10 120 python tools/grad_ab.py resid 256 3 2>&1 | grep "pime trace" | tail -2 >> $OUT/${TAG}_ab.log
cat $OUT/${TAG}_ab.log
