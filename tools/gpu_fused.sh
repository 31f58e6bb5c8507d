#!/bin/bash
# quick iteration on the LDS-resident gradient kernels: parity (fused + golden update), then timings + phase trace
OUT=gpurun_out; TAG=${1:-f}
mkdir -p $OUT; rm -f $OUT/${TAG}_ab.log
timeout -k 10 600 python -m pytest tests/test_gpu_ppo_fused.py tests/test_gpu_update_golden.py tests/test_gpu_kernels_parity.py -q -x > $OUT/${TAG}_pytest.log 2>&1
rc=$?; tail -3 $OUT/${TAG}_pytest.log
if [ $rc -ne 0 ]; then exit 1; fi
for cfg in "modular 128" "resid 128" "resid 64" "modular 64"; do
  timeout -k 10 120 python tools/grad_ab.py $cfg 2>&1 | grep "us per minibatch" >> $OUT/${TAG}_ab.log || exit 1
done
PIME_FUSED_TRACE=0 timeout -k 10 120 python tools/grad_ab.py modular 128 2>&1 | grep "pime trace" | tail -2 >> $OUT/${TAG}_ab.log
PIME_FUSED_TRACE=0 timeout -k 10 120 python tools/grad_ab.py resid 128 2>&1 | grep "pime trace" | tail -2 >> $OUT/${TAG}_ab.log
cat $OUT/${TAG}_ab.log
