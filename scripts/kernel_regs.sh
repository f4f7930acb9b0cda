#!/bin/bash
# scripts/kernel_regs.sh [extra hipcc flags]: registers, spills and scratch of the kernels (compile only, no GPU needed)
cd "$(dirname "$0")/../raytracing_project_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -fPIC -c rtx.hip -o /dev/null \
  -Rpass-analysis=kernel-resource-usage "$@" 2>&1 | python3 -c '
import re, sys, subprocess
cur = None; rows = {}
for line in sys.stdin:
    m = re.search(r"remark: [^:]*:\d+:\d+:\s+(.*?)\s+\[-Rpass", line) or re.search(r"remark:\s+(.*?)\s+\[-Rpass", line)
    if not m: continue
    t = m.group(1)
    if t.startswith("Function Name:") or t.startswith("Name:"):
        cur = t.split(":", 1)[1].strip(); rows[cur] = {}
    elif cur and ":" in t:
        k, v = t.split(":", 1); rows[cur][k.strip()] = v.strip()
names = list(rows)
dem = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.splitlines()
for n, d in zip(names, dem):
    r = rows[n]; d = re.sub(r"\(.*", "", d).replace("void rtx::", "")
    print("%-46s VGPRs %3s  spilled VGPRs %2s  SGPR spill %2s  scratch %4s B  occupancy %s" % (d[:46], r.get("VGPRs"), r.get("VGPRs Spill", r.get("VGPR Spill", "?")), r.get("SGPRs Spill", r.get("SGPR Spill", "?")), r.get("ScratchSize [bytes/lane]"), r.get("Occupancy [waves/SIMD]")))
'
