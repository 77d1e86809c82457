#!/usr/bin/env python3
"""Per-kernel register / spill / scratch table from hipcc's -Rpass-analysis=kernel-resource-usage remarks.
usage: tools/kernel_resources.py <file.hip> [filter substring]   (compiles for gfx950; no GPU needed)"""
import os, re, subprocess, sys, tempfile
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
extra = os.environ.get("SVGP_EXTRA_FLAGS", "").split()
with tempfile.TemporaryDirectory() as td:
    r = subprocess.run(["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-Wno-unused-function", *extra,
                        "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", os.path.join(td, "x.o")], capture_output=True, text=True)
cur, rows = None, []
for ln in r.stderr.splitlines():
    m = re.search(r"remark: Function Name: (\S+)", ln)
    if m:
        name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        cur = {"name": name.replace("svgp::(anonymous namespace)::", "").replace("void ", "").split("(")[0]}
        rows.append(cur)
        continue
    m = re.search(r"remark:\s+(VGPRs|AGPRs|SGPRs|ScratchSize \[bytes/lane\]|SGPRs Spill|VGPRs Spill|LDS Size \[bytes/block\]|Occupancy \[waves/SIMD\]): (\d+)", ln)
    if m and cur is not None:
        cur[m.group(1)] = int(m.group(2))
for c in rows:
    if flt in c["name"]:
        print(f'{c["name"][:90]:90s} VGPR {c.get("VGPRs", 0):3d} AGPR {c.get("AGPRs", 0):3d} spillV {c.get("VGPRs Spill", 0):3d} spillS {c.get("SGPRs Spill", 0):3d} '
              f'scratch {c.get("ScratchSize [bytes/lane]", 0):4d} occ {c.get("Occupancy [waves/SIMD]", 0)}')
