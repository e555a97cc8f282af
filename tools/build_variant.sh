#!/bin/bash
# Build an A/B variant of libcnf_hip.so with extra compiler flags:  tools/build_variant.sh <name> <flags...>
#   -> categoricalnf_amd/lib/var_<name>.so  (git-ignored, travels to the GPU box); select it with
#   CNF_LIB_OVERRIDE=categoricalnf_amd/lib/var_<name>.so
set -e
cd "$(dirname "$0")/.."
name=$1; shift
out=build/variants/$name
mkdir -p "$out"
make -s -C categoricalnf_amd/csrc OBJDIR=../../$out LIB=../lib/var_$name.so \
    CXXFLAGS="-O3 -std=c++17 -ffp-contract=off -fPIC -Wall -Wno-unused-function $*" 2>&1 | grep -E "error|Error" || true
ls -la categoricalnf_amd/lib/var_$name.so
