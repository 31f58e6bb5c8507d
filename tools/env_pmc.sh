#!/bin/bash
# usage: tools/env_pmc.sh <out.json>
# HBM-side bytes per launch of the env step kernels over the lane sweep of tools/env_roofline_sweep.py, as
# /opt/skills/guides/MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 --pmc passes (with
# --kernel-trace only), kilobyte units, FETCH_SIZE doubled on gfx950; dispatches grouped by (kernel, grid size = lane count).
# Durations come from the same passes' dispatch timestamps (a profiled pass runs a few % slower than an un-profiled one).
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$1
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/envpmc_$c
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/envpmc_$c -o run -- python3 "$ROOT/tools/env_roofline_sweep.py" > /tmp/envpmc_$c.out 2>&1 || { tail -20 /tmp/envpmc_$c.out; exit 1; }
done
python3 - "$ROOT/$OUT" <<'PY'
import csv, glob, json, sys, collections
res = collections.defaultdict(lambda: {"launches": 0, "FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0, "ns": 0.0})
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"/tmp/envpmc_{c}/**/*counter_collection.csv", recursive=True)[0]
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != c or "step_kernel" not in r["Kernel_Name"]:
            continue
        k = (r["Kernel_Name"].split("(")[0][:90], int(r["Grid_Size"]))
        res[k][c] += float(r["Counter_Value"])
        if c == "FETCH_SIZE":
            res[k]["launches"] += 1
            res[k]["ns"] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
ALG = {"ph": {"contract_76B": 76, "kernel_97B": 97}, "wt": {"contract_84B": 84, "kernel_93B": 93},
       "ph16": {"contract_64B": 64, "kernel_79B": 79}}   # binary16 I / obs / reward (state_mode mixed16), f64 x
out = []
for (name, lanes), v in sorted(res.items()):
    n = max(v["launches"], 1)
    kind = ("ph16" if "_Float16" in name else "ph") if "ph_step" in name else "wt"
    rd, wr, us = 2 * v["FETCH_SIZE"] / n * 1024, v["WRITE_SIZE"] / n * 1024, v["ns"] / n / 1e3
    row = {"kernel": name, "lanes": lanes, "launches": v["launches"], "us_per_launch_profiled": us,
           "hbm_read_bytes_per_lane": rd / lanes, "hbm_write_bytes_per_lane": wr / lanes,
           "hbm_traffic_GBps": (rd + wr) / us / 1e3, "frac_of_8TBps_traffic": (rd + wr) / us / 1e3 / 8000,
           "resident_in_infinity_cache": bool(lanes * 100 < 256 * 2 ** 20 / 2)}
    for tag, b in ALG[kind].items():
        row[f"algorithmic_GBps_{tag}"] = b * lanes / us / 1e3
        row[f"frac_of_8TBps_{tag}"] = b * lanes / us / 1e3 / 8000
    out.append(row)
    print(json.dumps(row))
json.dump(out, open(sys.argv[1], "w"), indent=1)
PY
