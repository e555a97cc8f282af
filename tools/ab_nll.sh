#!/bin/bash
# A/B of two builds of the library on the bench's dominant kernel, interleaved: tools/ab_nll.sh <variant .so>
cd "$(dirname "$0")/.." || exit 1
for rep in 1 2 3; do
  for lib in "" "$1"; do
    CNF_LIB_OVERRIDE=$lib python tools/sustained_probe.py 2>/dev/null | grep "forward+NLL" | sed "s|^|${lib:-current} |"
  done
done
