#!/bin/bash
# usage: tools/pmc_pipe.sh <out.json> <python script> [args]
# Matrix-pipe occupancy of every kernel: SQ_VALU_MFMA_BUSY_CYCLES (busy cycles summed over the SIMDs) against
# GRBM_GUI_ACTIVE (cycles the GPU was active for the dispatch) x 1024 SIMDs, plus SQ_WAIT_ANY / SQ_WAVE_CYCLES (share of
# wave-cycles spent waiting), each counter in its own rocprofv3 --pmc pass.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$1; shift
cd /tmp && export TMPDIR=/tmp
for c in SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE; do
  rm -rf /tmp/pmc_$c
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_$c -o run -- python3 "$ROOT/$1" "${@:2}" > /tmp/pmc_$c.out 2>&1 || { echo "pass $c failed"; tail -5 /tmp/pmc_$c.out; }
done
python3 - "$ROOT/$OUT" <<'PY'
import csv, glob, json, sys, collections
res = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(int)
for d in glob.glob("/tmp/pmc_*/"):
    fs = glob.glob(d + "**/*counter_collection.csv", recursive=True)
    if not fs:
        continue
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"][:80]
        res[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            cnt[k] += 1
            if "Start_Timestamp" in r and "End_Timestamp" in r:
                res[k]["_ns"] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
out = {}
for k, v in res.items():
    n = max(cnt[k], 1)
    g = v.get("GRBM_GUI_ACTIVE", 0.0)
    if g <= 0:
        continue
    # GRBM_GUI_ACTIVE is summed over the 8 XCDs; SQ counters over all 1024 SIMDs
    o = {"launches": cnt[k], "gui_active_cycles_per_xcd": g / n / 8}
    if v.get("_ns"):
        o["duration_us_in_this_pass"] = v["_ns"] / n / 1e3
        o["clock_ghz"] = (g / n / 8) / (v["_ns"] / n)
    if "SQ_VALU_MFMA_BUSY_CYCLES" in v:
        o["mfma_busy_cycles_per_simd"] = v["SQ_VALU_MFMA_BUSY_CYCLES"] / n / 1024
        o["mfma_busy_frac"] = v["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / (g / 8)
    if v.get("SQ_WAVE_CYCLES"):
        o["wait_any_frac_of_wave_cycles"] = v.get("SQ_WAIT_ANY", 0.0) / v["SQ_WAVE_CYCLES"]
    if v.get("SQ_LDS_IDX_ACTIVE"):
        o["lds_bank_conflict_frac"] = v.get("SQ_LDS_BANK_CONFLICT", 0.0) / v["SQ_LDS_IDX_ACTIVE"]
    out[k] = o
json.dump(out, open(sys.argv[1], "w"), indent=1)
for k, o in sorted(out.items(), key=lambda kv: -kv[1]["gui_active_cycles_per_xcd"] * kv[1]["launches"])[:6]:
    print(k[:56], {a: round(b, 3) if isinstance(b, float) else b for a, b in o.items()})
PY
