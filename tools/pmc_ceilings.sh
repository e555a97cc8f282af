#!/bin/bash
# Counter passes for tools/ceilings.py.   bash tools/pmc_ceilings.sh <outdir under gpurun_out> [workload command]
# (default workload: the forward kernels, tools/pmc_ceilings_workload.py; the backward ones: python tools/bwd_probe.py --pmc)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$1
mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp; cd "$ROOT"
export CNF_MANIFEST=$OUT/manifest.json
shift
WORK=${*:-python tools/pmc_ceilings_workload.py}
i=0
for set in \
  "SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F32 SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE" \
  "FETCH_SIZE" \
  "WRITE_SIZE" ; do
  i=$((i+1))
  rm -rf "$OUT/pass$i"
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$OUT/pass$i" -o pmc -- $WORK > "$OUT/pass$i.log" 2>&1
  echo "pass$i: exit $?"
done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o st -- $WORK > "$OUT/stats.log" 2>&1
python tools/ceilings.py "$OUT" "$OUT/ceilings.json" | tee "$OUT/ceilings.txt"
find "$OUT" -name "*kernel_trace.csv" -delete
