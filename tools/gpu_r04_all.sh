#!/bin/bash
# usage: tools/gpu_r04_all.sh <tag> -- the full -m gpu suite, smoke(), then the evidence set (tools/gpu_profiles.sh)
TAG=$1; OUT=gpurun_out; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -m gpu -q -x > $OUT/${TAG}_pytest.log 2>&1
rc=$?; tail -4 $OUT/${TAG}_pytest.log
if [ $rc -ne 0 ]; then tail -60 $OUT/${TAG}_pytest.log; exit 1; fi
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/${TAG}_smoke.log 2>&1 || { tail -20 $OUT/${TAG}_smoke.log; exit 1; }
tail -1 $OUT/${TAG}_smoke.log
bash tools/gpu_profiles.sh $TAG
