#!/usr/bin/env python3
"""Print VGPR/AGPR/SGPR/scratch/occupancy per kernel of one .hip translation unit
(hipcc -Rpass-analysis=kernel-resource-usage), optionally filtered by substring."""
import re
import subprocess
import sys

src = sys.argv[1]
filt = sys.argv[2:] if len(sys.argv) > 2 else []
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-slp-vectorize", "--offload-arch=gfx950",
       "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
name = None
rec = {}
rows = []
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        if name:
            rows.append((name, rec))
        name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        rec = {}
        continue
    m = re.search(r"remark:\s+([\w \[\]/]+?): (\S+)", line)
    if m and name:
        rec[m.group(1).strip()] = m.group(2)
if name:
    rows.append((name, rec))
for name, r in rows:
    short = re.sub(r"ilqr::|\(ilqr::Params.*", "", name).replace("void ", "")
    if filt and not all(f in short for f in filt):
        continue
    print(f"{short:90s} V={r.get('VGPRs')} A={r.get('AGPRs')} S={r.get('TotalSGPRs')} scr={r.get('ScratchSize [bytes/lane]')} "
          f"occ={r.get('Occupancy [waves/SIMD]')} sspill={r.get('SGPRs Spill')} vspill={r.get('VGPRs Spill')}")
