#!/bin/bash
# rocprofv3 counter passes (each its own run, --kernel-trace only) over one workload.
#   bash tools/pmc_passes.sh <outdir under gpurun_out> <python workload and args...>
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$1; shift
mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp; cd "$ROOT"
i=0
for set in \
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
  "SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_SALU" \
  "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
  "GRBM_GUI_ACTIVE GRBM_COUNT SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_WAVES_EQ_64 SQ_THREAD_CYCLES_VALU" \
  "FETCH_SIZE" \
  "WRITE_SIZE" ; do
  i=$((i+1))
  if [ -n "${PMC_ONLY:-}" ] && ! echo " $PMC_ONLY " | grep -q " $i "; then continue; fi
  rm -rf "$OUT/pass$i"
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$OUT/pass$i" -o pmc -- "$@" > "$OUT/pass$i.log" 2>&1
  echo "pass$i ($set): exit $?"; tail -2 "$OUT/pass$i.log"
done
[ -n "${PMC_ONLY:-}" ] || timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o st -- "$@" > "$OUT/stats.log" 2>&1
python tools/pmc_table.py "$OUT" > "$OUT/table.txt" 2>&1; cat "$OUT/table.txt"
find "$OUT" -name "*kernel_trace.csv" -delete; find "$OUT" -name "*counter_collection.csv" -size +8M -delete
