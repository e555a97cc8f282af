#!/bin/bash
# A/B of builds of the library on the bench's dominant kernel, interleaved: tools/ab_nll.sh <variant .so> [<variant .so> ...]
# (variants: tools/build_variant.sh <tag> "<flags>").  Prints the burst and the sustained forward-only line per build.
cd "$(dirname "$0")/.." || exit 1
for rep in 1 2 3; do
  for lib in "" "$@"; do
    CNF_LIB_OVERRIDE=$lib python tools/sustained_probe.py 2>/dev/null | grep "forward+NLL" | grep -v "(c)" | sed "s|^|$(basename ${lib:-current}) |"
  done
done
