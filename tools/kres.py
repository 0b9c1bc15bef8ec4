#!/usr/bin/env python3
"""Kernel resource table from hipcc -Rpass-analysis=kernel-resource-usage:
   python tools/kres.py spark-tts_amd/csrc/smi_llm.hip"""
import re, subprocess, sys
src = sys.argv[1]
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Iinclude", "-Ispark-tts_amd/csrc",
       "-ffp-contract=off", "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur, rows = None, {}
for line in out.splitlines():
    if "error" in line:
        print(line)
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = m.group(1); rows[cur] = {}
        continue
    m = re.search(r"remark:\s+([A-Za-z ]+(?:\[[^\]]*\])?[A-Za-z ]*): (\d+)", line)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
for name, r in rows.items():
    dm = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    dm = re.sub(r"\(anonymous namespace\)::", "", dm).split("(")[0]
    print(f"{dm:55s} vgpr {r.get('VGPRs', -1):4d} agpr {r.get('AGPRs', 0):3d} spill {r.get('VGPRs Spill', 0):3d} "
          f"occ {r.get('Occupancy [waves/SIMD]', -1)} lds {r.get('LDS Size [bytes/block]', 0)}")
