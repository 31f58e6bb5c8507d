#!/bin/bash
# usage: tools/kstats.sh <python script> [args]   -- prints the per-kernel summary of a rocprofv3 kernel trace
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kstats_prof
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kstats_prof -o run -- python3 "$ROOT/$1" "${@:2}" > /tmp/kstats.out 2>&1
python3 - <<'PY'
import csv, glob
fs = glob.glob("/tmp/kstats_prof/**/*kernel_stats.csv", recursive=True)
if not fs:
    print(open("/tmp/kstats.out").read()[-2000:])
    raise SystemExit(1)
for r in list(csv.DictReader(open(fs[0])))[:12]:
    print(f'{r["Name"][:70]:70s} calls={r["Calls"]:>6s} avg_us={float(r["AverageNs"])/1e3:9.1f} pct={r["Percentage"]}')
PY
