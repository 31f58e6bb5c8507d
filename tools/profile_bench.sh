#!/bin/bash
# usage: tools/profile_bench.sh <tag> [bench args ...]  -- rocprofv3 --kernel-trace --stats of the bench (default workload, or the given
# arguments, e.g. "--workload wt_td3"), summary CSV into gpurun_out/<tag>_kernel_stats.csv
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o run -- python3 "$ROOT/bench.py" --steps 5 --warmup 2 --no-cpu-baseline "$@" > /tmp/prof_$TAG.json 2> /tmp/prof_$TAG.err || { tail -20 /tmp/prof_$TAG.err; exit 1; }
f=$(find /tmp/prof_$TAG -name '*kernel_stats.csv' | head -1)
cp "$f" "$ROOT/gpurun_out/${TAG}_kernel_stats.csv"
tail -1 /tmp/prof_$TAG.json > "$ROOT/gpurun_out/${TAG}_bench_under_rocprof.json"
head -8 "$f" | cut -c1-160
