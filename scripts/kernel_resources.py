#!/usr/bin/env python3
"""usage: scripts/kernel_resources.py csrc/file.hip [extra hipcc flags] -- one line per kernel: registers, spills, LDS, occupancy
(hipcc -Rpass-analysis=kernel-resource-usage, condensed)."""
import re
import subprocess
import sys

out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", sys.argv[1], "-o", "/dev/null",
                      "-Rpass-analysis=kernel-resource-usage"] + sys.argv[2:], capture_output=True, text=True).stderr
cur, rows = None, {}
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = m.group(1)
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
    if m and cur:
        rows[cur][m.group(1).strip()] = m.group(2)
    if "error" in line:
        print(line)
for k, v in rows.items():
    name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()[:90]
    g = lambda key: v.get(key, "?")
    print(f"{name:92s} VGPR {g('VGPRs'):>4} AGPR {g('AGPRs'):>4} spill {g('VGPRs Spill'):>3} LDS {g('LDS Size'):>6} occ {g('Occupancy')}")
