#!/usr/bin/env python3
"""Compile one .hip file for gfx950 and print VGPR / AGPR / scratch / occupancy per kernel
(hipcc -Rpass-analysis=kernel-resource-usage).  Usage: kernel_resources.py file.hip [filter]"""
import re
import subprocess
import sys

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-c", "-o", "/dev/null",
                      src, "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True).stderr
cur = {}
for line in out.splitlines():
    if "error:" in line:
        print(line)
    m = re.search(r"remark:\s+(.*?)\s+\[-Rpass", line)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = {"name": t.split(":", 1)[1].strip()}
    elif ":" in t:
        k, v = t.rsplit(":", 1)
        cur[k.strip()] = v.strip()
        if k.strip().startswith("LDS Size"):
            n = re.sub(r"EEvP.*", "", cur["name"])[-44:]
            if flt in n:
                print(f"{n:44s} VGPR {cur.get('VGPRs','?'):>4s} AGPR {cur.get('AGPRs','?'):>3s} scratch "
                      f"{cur.get('ScratchSize [bytes/lane]','?'):>4s} occ {cur.get('Occupancy [waves/SIMD]','?')}")
