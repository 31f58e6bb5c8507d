#!/bin/bash
# matrix-pipe busy cycles and clock of the in-place layer microbenchmark (tools/bin/layer16_b3_bench): one rocprofv3 --pmc pass per counter
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
o=$ROOT/gpurun_out/${1:-r04o}; mkdir -p $o
cd /tmp && export TMPDIR=/tmp
for c in GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES; do
  rm -rf /tmp/lp_$c
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/lp_$c -o run -- $ROOT/tools/bin/layer16_b3_bench > /tmp/lp_$c.out 2>&1 || { tail -5 /tmp/lp_$c.out; exit 1; }
done
python3 - $o/layer16_b3_bench_pmc.txt <<'PY'
import csv, glob, sys, collections
res = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("GRBM_GUI_ACTIVE", "SQ_VALU_MFMA_BUSY_CYCLES"):
    f = glob.glob(f"/tmp/lp_{c}/**/*counter_collection.csv", recursive=True)[0]
    for r in csv.DictReader(open(f)):
        res[r["Kernel_Name"][:70]][c].append((float(r["Counter_Value"]), float(r["End_Timestamp"]) - float(r["Start_Timestamp"])))
out = []
for k, v in res.items():
    if "bench" not in k:
        continue
    # the LAST dispatch of each kernel is the timed one with the most layers; take the longest
    g = max(v["GRBM_GUI_ACTIVE"], key=lambda t: t[1]); m = max(v["SQ_VALU_MFMA_BUSY_CYCLES"], key=lambda t: t[1])
    cyc = g[0] / 8
    out.append(f"{k:70s} {g[1] / 1e3:9.1f} us  clock {cyc / g[1]:.2f} GHz  matrix pipe busy {m[0] / 1024 / (m[1] * cyc / g[1]):.2f} of the cycles")
open(sys.argv[1], "w").write("\n".join(out) + "\n")
print("\n".join(out))
PY
