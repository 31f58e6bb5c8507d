#!/bin/bash
# usage: tools/spill_check.sh <file.hip> [kernel-name substring]   (no GPU)
# SGPR / VGPR spill counts and the readlane / writelane / s_nop / scratch instruction counts of every kernel in one source file.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CSRC=$ROOT/pime-robust-non-linear-set-point-control-with-reinforcement-learning_amd/csrc
OUT=${TMPDIR:-/tmp}/spill_check; mkdir -p $OUT
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -fvisibility=hidden -ffp-contract=off --offload-arch=gfx950 -I$ROOT/include -DPIME_BUILD \
    --cuda-device-only -S $CSRC/$1 -o $OUT/k.s
python3 - "$OUT/k.s" "${2:-}" <<'PY'
import re, subprocess, sys
txt = open(sys.argv[1]).read()
pat = sys.argv[2]
counts = {}
for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)^\.Lfunc_end\d+:", txt, re.S | re.M):
    body = m.group(2)
    cnt = lambda s: len(re.findall(r"^\s+" + s, body, re.M))
    counts[m.group(1)] = (f"readlane {cnt('v_readlane'):5d} writelane {cnt('v_writelane'):5d} s_nop {cnt('s_nop'):5d} scratch_ld "
                          f"{cnt('scratch_load'):4d} scratch_st {cnt('scratch_store'):4d} mfma {cnt('v_mfma'):5d} valu {cnt('v_'):6d} "
                          f"salu {cnt('s_'):6d} ds {cnt('ds_'):5d}")
for blk in re.findall(r"- \.agpr_count:.*?\.wavefront_size:\s+\d+", txt, re.S):
    nm = re.search(r"\.name:\s+(\S+)", blk).group(1)
    dn = subprocess.run(["c++filt", nm], capture_output=True, text=True).stdout.strip()
    if pat and pat not in dn:
        continue
    f = lambda k: re.search(r"\." + k + r":\s+(\d+)", blk).group(1)
    print(f"{dn[:90]:90s} sgpr_spill {f('sgpr_spill_count')} vgpr_spill {f('vgpr_spill_count')} vgpr {f('vgpr_count')} agpr {f('agpr_count')} scratch {f('private_segment_fixed_size')}\n    {counts.get(nm, '')}")
PY
