#!/bin/bash
# MFMA utilisation of the coupling sub-network GEMMs (north_star: "MFMA utilisation reported against the chip's
# peak"): counter passes over one training configuration.   bash tools/mfma_util.sh <outdir> <python script + args>
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$1; shift
mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp; cd "$ROOT"
i=0
for set in \
  "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE" ; do
  i=$((i+1))
  rm -rf "$OUT/pass$i"
  timeout 600 rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$OUT/pass$i" -o pmc -- "$@" > "$OUT/pass$i.log" 2>&1
  echo "pass$i: exit $?"; tail -1 "$OUT/pass$i.log"
done
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o st -- "$@" > "$OUT/stats.log" 2>&1
python tools/mfma_util.py "$OUT" "$OUT/mfma_util.json" | tee "$OUT/mfma_util.txt"
find "$OUT" -name "*kernel_trace.csv" -delete; find "$OUT" -name "*counter_collection.csv" -size +16M -delete
