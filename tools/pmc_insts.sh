#!/bin/bash
# usage: tools/pmc_insts.sh <out.json> <python script> [args]
# Dynamic instruction mix of every kernel (per wave) and the SQ's view of where the cycles go: one counter per rocprofv3 --pmc
# pass (with --kernel-trace only).  SQ counters are summed over the chip; per-wave figures divide by SQ_WAVES.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$1; shift
cd /tmp && export TMPDIR=/tmp
CS="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_INSTS_BRANCH SQ_INSTS_VALU_TRANS_F32 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY GRBM_GUI_ACTIVE"
for c in $CS; do
  rm -rf /tmp/pmci_$c
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmci_$c -o run -- python3 "$ROOT/$1" "${@:2}" > /tmp/pmci_$c.out 2>&1 || { echo "pass $c failed"; tail -3 /tmp/pmci_$c.out; }
  echo "pass $c done"
done
python3 - "$ROOT/$OUT" <<'PY'
import csv, glob, json, sys, collections
res = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for d in glob.glob("/tmp/pmci_*/"):
    fs = glob.glob(d + "**/*counter_collection.csv", recursive=True)
    if not fs:
        continue
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"][:60]
        res[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[k][r["Counter_Name"]] += 1
out = {}
for k, v in res.items():
    o = {c: v[c] / max(cnt[k][c], 1) for c in v}   # per launch, chip-wide
    w = o.get("SQ_WAVES", 0)
    if w > 0:
        o["per_wave"] = {c[3:]: o[c] / w for c in o if c.startswith("SQ_INSTS")}
    out[k] = o
json.dump(out, open(sys.argv[1], "w"), indent=1)
for k, o in sorted(out.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CYCLES", 0))[:4]:
    print(k)
    print("   per launch:", {a: round(b) for a, b in o.items() if a != "per_wave"})
    print("   per wave  :", {a: round(b, 1) for a, b in o.get("per_wave", {}).items()})
PY
