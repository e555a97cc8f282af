#!/bin/bash
# fp64 instruction mix of the reference-precision mixture kernels:  bash tools/pmc_fp64.sh <outdir under gpurun_out>
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$1
mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp; cd "$ROOT"
export CNF_MANIFEST=$OUT/manifest.json
i=0
for set in \
  "SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_TRANS_F32" \
  "GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32" \
  "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS" ; do
  i=$((i+1))
  rm -rf "$OUT/pass$i"
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$OUT/pass$i" -o pmc -- python tools/pmc_mixture_fp64_workload.py > "$OUT/pass$i.log" 2>&1
  echo "pass$i: exit $?"; tail -1 "$OUT/pass$i.log"
done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o st -- python tools/pmc_mixture_fp64_workload.py > "$OUT/stats.log" 2>&1
python tools/fp64_ceilings.py "$OUT" | tee "$OUT/fp64_ceilings.txt"
find "$OUT" -name "*kernel_trace.csv" -size +4M -delete
