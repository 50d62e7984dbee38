#!/bin/bash
# tools/kernel_resources.sh [extra hipcc flags]: VGPRs / SGPRs / spills / scratch / LDS / occupancy of every chain kernel
# (-Rpass-analysis=kernel-resource-usage on reorder_kernels.hip, device pass only).
root=$(cd "$(dirname "$0")/.." && pwd)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -x hip -I$root/include -I$root/spring_amd/csrc "$@" \
  --cuda-device-only -Rpass-analysis=kernel-resource-usage -c $root/spring_amd/csrc/reorder_kernels.hip -o /dev/null 2>&1 |
python3 -c '
import re, sys, subprocess
cur = None; rows = {}
for ln in sys.stdin:
    m = re.search(r"remark: .*Function Name: (\S+)", ln)
    if m:
        cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip(); rows[cur] = {}; continue
    m = re.search(r"remark: .*?\s+([A-Za-z ]+(?:\[[^\]]*\])?[A-Za-z ]*): (\d+)", ln)
    if m and cur: rows[cur][m.group(1).strip()] = int(m.group(2))
for k, r in rows.items():
    if not re.search(r"k_round|k_search|k_long|k_apply|k_mg_", k): continue
    print("%-64s VGPR %3d SGPR %3d spillS %3d spillV %3d scratch %3d occ %d LDS %5d" % (k[:64], r.get("VGPRs", -1), r.get("SGPRs", -1),
          r.get("SGPRs Spill", -1), r.get("VGPRs Spill", -1), r.get("ScratchSize [bytes/lane]", -1), r.get("Occupancy [waves/SIMD]", -1), r.get("LDS Size [bytes/block]", -1)))
'
