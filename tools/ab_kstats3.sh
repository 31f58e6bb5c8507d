#!/bin/bash
# like ab_kstats.sh with three interleaved repetitions and a per-variant mean of the two gradient kernels
TAG=$1; CFG=$2; shift 2
OUT=gpurun_out/${TAG}.log; rm -f $OUT
for rep in 1 2 3; do
  for v in "$@"; do
    PIME_ALLOW_LIB_OVERRIDE=1 PIME_LIB_PATH=$PWD/variants/$v.so timeout -k 10 200 bash tools/kstats.sh tools/grad_ab.py $CFG 2>&1 | grep "ppo_fused_kernel\|ppo16_kernel" | head -2 | sed "s/^/$v /" >> $OUT || exit 1
  done
done
python3 - $OUT <<'PY'
import sys,re,collections
d=collections.defaultdict(list)
for l in open(sys.argv[1]):
    m=re.match(r'(\S+) void pime::(\S+<[^>]*>).*avg_us=\s*([\d.]+)',l)
    if m: d[(m.group(1),m.group(2))].append(float(m.group(3)))
for k,v in sorted(d.items(), key=lambda kv:(kv[0][1],kv[0][0])): print(f"{k[1]:28s} {k[0]:10s} mean {sum(v)/len(v):7.1f} us   runs {v}")
PY
