#!/usr/bin/env python3
"""Per-kernel register / spill / occupancy table of one .hip file (hipcc -Rpass-analysis=kernel-resource-usage, gfx950).
usage: python scripts/kernel_resources.py spatialrgpt_amd/csrc/skinny.hip [extra hipcc flags]"""
import re
import subprocess
import sys

src = sys.argv[1]
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-mllvm", "-amdgpu-kernarg-preload-count=16",
       "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"] + sys.argv[2:]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = []
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        name = subprocess.run(["/usr/bin/c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        name = name.replace("(anonymous namespace)::", "")
        cur = {"name": name[:name.index(">(") + 1] if ">(" in name else re.sub(r"\(.*", "", name)}
        rows.append(cur)
        continue
    m = re.search(r"remark: +([A-Za-z ]+(?:\[[^\]]*\])?): (\d+)", line)
    if m and cur is not None:
        cur[m.group(1).strip()] = int(m.group(2))
print(f"{'VGPR':>5} {'AGPR':>5} {'SGPR':>5} {'vspill':>6} {'sspill':>6} {'occ':>4} {'LDS':>7}  kernel")
for r in rows:
    print(f"{r.get('VGPRs', -1):>5} {r.get('AGPRs', -1):>5} {r.get('TotalSGPRs', -1):>5} {r.get('VGPRs Spill', -1):>6} "
          f"{r.get('SGPRs Spill', -1):>6} {r.get('Occupancy [waves/SIMD]', -1):>4} {r.get('LDS Size [bytes/block]', -1):>7}  {r['name']}")
