#!/bin/bash
# Compact parameter layout at the model level: the last Linear's GEMM (tools/compact_params_probe.py), the set-modelling training step
# and its kernel table with and without CNF_COMPACT_PARAMS=1.   gpurun --timeout 1200 -- 'bash tools/compact_params_run.sh'
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/compact
mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp; cd "$ROOT"
( python tools/compact_params_probe.py 2>&1 | grep -v amdgpu.ids ) > "$OUT/probe.txt"; cat "$OUT/probe.txt"
for b in 1024 8192; do
  for c in 0 1; do
    ( CNF_COMPACT_PARAMS=$c timeout 300 python tools/bench_train_step.py $b 20 2>&1 | grep "^batch" | sed "s/^/compact_params=$c  /" ) >> "$OUT/train_step.txt"
  done
done
cat "$OUT/train_step.txt"
for c in 0 1; do
  rm -rf "$OUT/prof$c"
  CNF_COMPACT_PARAMS=$c timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof$c" -o tr -- python tools/bench_train_step.py 8192 12 > /dev/null 2>&1
  python tools/summarize_kernel_stats.py "$OUT/prof$c/tr_kernel_stats.csv" "$OUT/train_step_kernel_stats_compact$c.csv" "CNF_COMPACT_PARAMS=$c python tools/bench_train_step.py 8192 12 (set-modelling training step: 8 flow steps, Transformer sub-network hidden 256, D=4, K=8, |S|=16)" 30 | head -3
  rm -f "$OUT"/prof$c/*kernel_trace.csv
done
