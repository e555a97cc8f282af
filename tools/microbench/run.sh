#!/bin/bash
# Builds and runs tools/microbench/enc_micro.hip on the GPU box, then repeats its calibration part under rocprofv3 --pmc so
# that the SQ counters can be read against instruction streams whose issue rate is known.
#   bash tools/microbench/run.sh [outdir under gpurun_out] [args for enc_micro]
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-micro}
mkdir -p "$OUT"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off "$ROOT/tools/microbench/enc_micro.hip" -o /tmp/enc_micro || exit 1
timeout 300 /tmp/enc_micro ${2:-} > "$OUT/enc_micro.txt" 2>&1
echo "enc_micro exit $?"
cat "$OUT/enc_micro.txt"
if [ -z "${2:-}" ]; then
  cd /tmp; export TMPDIR=/tmp
  rm -rf "$OUT/pmc"
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F32 GRBM_GUI_ACTIVE \
      --output-format csv -d "$OUT/pmc" -o pmc -- /tmp/enc_micro calib > "$OUT/pmc.log" 2>&1
  echo "pmc exit $?"; tail -3 "$OUT/pmc.log"
  cd "$ROOT"
  python tools/microbench/calib_table.py "$OUT/pmc" > "$OUT/calib_pmc.txt" 2>&1
  cat "$OUT/calib_pmc.txt"
  find "$OUT/pmc" -name "*kernel_trace.csv" -delete
fi
