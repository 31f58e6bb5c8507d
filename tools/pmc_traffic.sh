#!/bin/bash
# usage: tools/pmc_traffic.sh <out.json> <python script> [args]
# HBM-side bytes per launch of every kernel, as /opt/skills/guides/MI355X_MICROARCH.md prescribes: FETCH_SIZE and
# WRITE_SIZE in SEPARATE rocprofv3 --pmc passes (they do not fit one pass), kilobyte units, FETCH_SIZE doubled on gfx950.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$1; shift
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_$c -o run -- python3 "$ROOT/$1" "${@:2}" > /tmp/pmc_$c.out 2>&1 || { tail -20 /tmp/pmc_$c.out; exit 1; }
done
python3 - "$ROOT/$OUT" <<'PY'
import csv, glob, json, sys, collections
res = collections.defaultdict(lambda: {"launches": 0, "FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0})
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"/tmp/pmc_{c}/**/*counter_collection.csv", recursive=True)[0]
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != c:
            continue
        k = r["Kernel_Name"][:80]
        res[k][c] += float(r["Counter_Value"])
        if c == "FETCH_SIZE":
            res[k]["launches"] += 1
out = {}
for k, v in sorted(res.items(), key=lambda kv: -kv[1]["FETCH_SIZE"] - kv[1]["WRITE_SIZE"]):
    n = max(v["launches"], 1)
    out[k] = {"launches": v["launches"], "fetch_kb_raw": v["FETCH_SIZE"] / n,
              "fetch_mb_corrected": 2 * v["FETCH_SIZE"] / n / 1024, "write_mb": v["WRITE_SIZE"] / n / 1024}
json.dump(out, open(sys.argv[1], "w"), indent=1)
for k, v in list(out.items())[:8]:
    print(f'{k[:60]:60s} n={v["launches"]:5d} read={v["fetch_mb_corrected"]:9.2f} MB write={v["write_mb"]:9.2f} MB')
PY
