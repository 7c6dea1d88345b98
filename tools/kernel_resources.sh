#!/bin/bash
# Register, scratch and occupancy figures of every kernel, as the compiler reports them.
# usage: tools/kernel_resources.sh [extra hipcc flags...]
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -std=c++17 -Wno-unused-value "$@" \
  -Rpass-analysis=kernel-resource-usage -c raytracer_amd/csrc/rt_device.hip -o /tmp/kernel_resources.o 2>&1 |
python3 -c '
import re, sys
cur = None
rows = {}
for line in sys.stdin:
    m = re.search(r"remark:\s+([\w \[\]/]+?):\s+(\S+)", line)
    if not m: continue
    k, v = m.group(1).strip(), m.group(2)
    if k == "Function Name" or k == "Name": cur = v; rows[cur] = {}
    elif cur: rows[cur][k] = v
import subprocess
for name, r in rows.items():
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().split("(")[0]
    print("%-30s VGPR %-4s SGPR %-4s scratch %-5s occupancy %-2s SGPR spills %-4s VGPR spills %s" % (dem, r.get("VGPRs"), r.get("TotalSGPRs"), r.get("ScratchSize [bytes/lane]"), r.get("Occupancy [waves/SIMD]"), r.get("SGPRs Spill"), r.get("VGPRs Spill")))
'
