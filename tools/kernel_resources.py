#!/usr/bin/env python3
"""Print VGPR / SGPR / scratch / occupancy / LDS per kernel from hipcc's kernel-resource-usage remarks.
Usage: python tools/kernel_resources.py [substring filter]"""
import re
import subprocess
import sys

out = subprocess.run(["make", "-C", "fm-for-online-recommendation_amd/csrc", "asm"], capture_output=True, text=True)
cur, rows = None, {}
for line in (out.stdout + out.stderr).splitlines():
    m = re.search(r"remark:\s+Function Name: (\S+)", line)
    if m:
        cur = re.sub(r"_ZN12_GLOBAL__N_1\d+", "", m.group(1))
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+([\w \[\]/]+?):\s+(\S+)\s+\[-Rpass", line)
    if m and cur:
        rows[cur][m.group(1).strip()] = m.group(2)
flt = sys.argv[1] if len(sys.argv) > 1 else ""
for k, v in rows.items():
    if flt in k:
        print(f"{k[:48]:48s} VGPR={v.get('VGPRs'):>4} SGPR={v.get('TotalSGPRs'):>4} scratch={v.get('ScratchSize [bytes/lane]'):>3} "
              f"occ={v.get('Occupancy [waves/SIMD]')} LDS={v.get('LDS Size [bytes/block]')}")
