#!/bin/bash
# Build an alternative libcnf_hip.so with extra compiler flags for A/B timing:
#   tools/build_variant.sh <tag> "<extra flags>"   ->  categoricalnf_amd/lib/var_<tag>.so
set -e
cd "$(dirname "$0")/../categoricalnf_amd/csrc"
make -j4 OBJDIR=../../build/var_$1 LIB=../lib/var_$1.so CXXFLAGS="-O3 -std=c++17 -ffp-contract=off -fPIC -Wno-unused-function $2" 2>&1 | grep -E "error|rror:" || true
ls -la ../lib/var_$1.so
