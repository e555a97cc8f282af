#!/usr/bin/env python3
"""VGPR / spill / occupancy table of one .hip file's kernels (hipcc -Rpass-analysis=kernel-resource-usage).

usage: python tools/kernel_resources.py categoricalnf_amd/csrc/cnf_mixture_tok.hip [extra hipcc flags]"""
import re
import subprocess
import sys
import os

src = sys.argv[1]
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-c", src,
       "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"] + sys.argv[2:]
err = subprocess.run(cmd, stderr=subprocess.PIPE, text=True, cwd=os.getcwd()).stderr
cur = None
rows = {}
for line in err.splitlines():
    m = re.search(r"remark: Function Name: (\S+)", line)
    if m:
        cur = m.group(1)
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+([A-Za-z \[\]/]+): (\S+) \[-Rpass", line)
    if m and cur:
        rows[cur][m.group(1).strip()] = m.group(2)
dem = subprocess.run(["c++filt"], input="\n".join(rows), stdout=subprocess.PIPE, text=True).stdout.splitlines()
print(f"{'kernel':100s} vgpr agpr vspill sspill waves/SIMD")
for name, d in zip(dem, rows.values()):
    name = re.sub(r"\(.*", "", name).replace("void cnf::", "")
    print(f"{name[:100]:100s} {d.get('VGPRs', '?'):>4s} {d.get('AGPRs', '?'):>4s} {d.get('VGPRs Spill', '?'):>6s} "
          f"{d.get('SGPRs Spill', '?'):>6s} {d.get('Occupancy [waves/SIMD]', '?'):>5s}")
