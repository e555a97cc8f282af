#!/bin/bash
# Memory-path counters (L1 -> L2 request counts and latencies, L2 -> fabric request sizes and stalls) of a workload, one rocprofv3
# pass per counter set (--kernel-trace only):   bash tools/pmc_memory_path.sh <outdir under gpurun_out> <command...>
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$1; shift
mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp; cd "$ROOT"
i=0
for set in \
  "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
  "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum" \
  "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" \
  "TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_RDREQ_LEVEL_sum TCC_BUSY_sum GRBM_GUI_ACTIVE" ; do
  i=$((i+1))
  rm -rf "$OUT/pass$i"
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$OUT/pass$i" -o pmc -- "$@" > "$OUT/pass$i.log" 2>&1
  echo "pass$i: exit $?"; tail -1 "$OUT/pass$i.log"
done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o st -- "$@" > "$OUT/stats.log" 2>&1
python tools/pmc_table.py "$OUT" > "$OUT/table.txt" 2>&1
find "$OUT" -name "*kernel_trace.csv" -delete; find "$OUT" -name "*counter_collection.csv" -size +8M -delete
