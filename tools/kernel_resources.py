#!/usr/bin/env python3
"""usage: tools/kernel_resources.py [out.txt]
Compiler-reported resources of every kernel in libpime_hip.so (hipcc -Rpass-analysis=kernel-resource-usage, gfx950):
VGPRs, AGPRs, scratch (spill) bytes per lane, static LDS, SGPRs, occupancy in waves per SIMD -- and, from the ISA of the same
compile (-S), how many scratch / private-memory INSTRUCTIONS and SGPR-spill lane moves the kernel actually contains: a frame can
be reserved (an SGPR-tuple spill slot the allocator then served from VGPR lanes) without a single access.  Needs no GPU."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "pime-robust-non-linear-set-point-control-with-reinforcement-learning_amd", "csrc")
FLAGS = ["-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-ffp-contract=off", "--offload-arch=gfx950",
         f"-I{ROOT}/include", "-DPIME_BUILD", "-Rpass-analysis=kernel-resource-usage"]


def demangle(name):
    return subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "kernel_resources.txt")
    lines = []
    for f in sorted(os.listdir(CSRC)):
        if not f.endswith(".hip"):
            continue
        p = subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, "-c", os.path.join(CSRC, f), "-o", "/dev/null"],
                           capture_output=True, text=True)
        tmp = f"/tmp/kres_{os.getpid()}_{f}.s"
        subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS[:-1], "--cuda-device-only", "-S", os.path.join(CSRC, f), "-o", tmp],
                       capture_output=True, text=True)
        asm = open(tmp).read() if os.path.exists(tmp) else ""
        if os.path.exists(tmp):
            os.remove(tmp)
        isa = {}
        for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)^\.Lfunc_end", asm, re.S | re.M):
            body = m.group(2)
            isa[m.group(1)] = (len(re.findall(r"^\s+(?:scratch_|buffer_(?:load|store)\S* .*\boff(?:en)?\b)", body, re.M)),
                               len(re.findall(r"^\s+v_(?:readlane|writelane)_b32", body, re.M)))
        cur, rows = None, []
        for line in p.stderr.splitlines():
            m = re.search(r"remark: .*?Function Name: (\S+)", line)
            if m:
                cur = {"name": m.group(1)}
                rows.append(cur)
                continue
            m = re.search(r"remark: [^:]*:\d+:\d+:\s+([A-Za-z][A-Za-z \[\]/]+): (\S+)", line) or \
                re.search(r"remark:\s+([A-Za-z][A-Za-z \[\]/]+): (\S+)", line)
            if m and cur is not None:
                cur[m.group(1).strip()] = m.group(2)
        for r in rows:
            g = r.get
            lines.append(f"{f}: {demangle(r['name'])[:84]:84s} VGPR {g('VGPRs', '?'):>4} AGPR {g('AGPRs', '?'):>4} "
                         f"scratch {g('ScratchSize [bytes/lane]', '?'):>5} B/lane  static LDS {g('LDS Size [bytes/block]', '?'):>6} B  "
                         f"SGPR {g('TotalSGPRs', '?'):>4}  spilled VGPRs {g('VGPRs Spill', '?'):>3}  waves/SIMD {g('Occupancy [waves/SIMD]', '?')}"
                         f"  ISA: scratch instrs {isa.get(r['name'], ('?', '?'))[0]}, SGPR-spill lane moves {isa.get(r['name'], ('?', '?'))[1]}")
    text = "\n".join(lines) + "\n"
    open(out, "w").write(text)
    sys.stdout.write(text)


if __name__ == "__main__":
    main()
