#!/bin/bash
# Register / LDS / occupancy table of every kernel of one translation unit (compile-only, no GPU needed):
#   tools/kernel_resources.sh categoricalnf_amd/csrc/cnf_backward.hip [extra hipcc flags]
src=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -c --cuda-device-only \
    -Rpass-analysis=kernel-resource-usage "$@" -o /dev/null "$src" 2>&1 | python3 -c '
import re, subprocess, sys
rows, cur = [], {}
for line in sys.stdin:
    m = re.search(r"remark: +([A-Za-z ]+\[?[A-Za-z/ ]*\]?): +(\S+)", line)
    if not m: continue
    k, v = m.group(1).strip(), m.group(2)
    if k == "Function Name":
        cur = {"name": v}; rows.append(cur)
    else:
        cur[k] = v
names = subprocess.run(["/usr/bin/c++filt"] + [r["name"] for r in rows], capture_output=True, text=True).stdout.split("\n")
print("%-100s %5s %5s %6s %7s %4s" % ("kernel", "VGPR", "AGPR", "spill", "LDS", "occ"))
for r, n in zip(rows, names):
    n = re.sub(r"\(.*", "", n).replace("void cnf::", "")
    print("%-100s %5s %5s %6s %7s %4s" % (n[:100], r.get("VGPRs"), r.get("AGPRs"), r.get("VGPRs Spill", r.get("ScratchSize [bytes/lane]")), r.get("LDS Size [bytes/block]"), r.get("Occupancy [waves/SIMD]")))
'
