#!/bin/bash
# Re-measure everything profiles/ holds, on the GPU box.  Run from the repo root:
#   gpurun --timeout 1500 -- 'bash tools/refresh_profiles.sh'
# then, back in the build container:  python tools/collect_profiles.py
# The --pmc passes are separate rocprofv3 runs with --kernel-trace only (never combined with sys/hip traces).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp; cd "$ROOT"

timeout 300 python bench.py > "$OUT/bench.log" 2>&1; tail -1 "$OUT/bench.log"

rm -rf "$OUT/prof_r01" "$OUT/pmc_fetch" "$OUT/pmc_write"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_r01" -o bench -- \
    python bench.py --no-cpu-baseline > "$OUT/bench_prof.log" 2>&1
tail -1 "$OUT/bench_prof.log" | cut -c1-300
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o pmc -- \
    python tools/pmc_workload.py > "$OUT/pmc_fetch.log" 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -o pmc -- \
    python tools/pmc_workload.py > "$OUT/pmc_write.log" 2>&1
python tools/pmc_summarize.py "$OUT/pmc_fetch" "$OUT/pmc_write" "$OUT/traffic.json" "$OUT/traffic.txt" | tail -8
rm -f "$OUT"/pmc_fetch/*kernel_trace.csv "$OUT"/pmc_write/*kernel_trace.csv "$OUT"/prof_r01/*kernel_trace.csv

timeout 300 python tools/sweep_affine.py > "$OUT/sweep_affine.log" 2>&1; tail -4 "$OUT/sweep_affine.log"
timeout 300 python tools/sweep_mixture.py > "$OUT/sweep_mixture.log" 2>&1; tail -4 "$OUT/sweep_mixture.log"
timeout 300 python tools/bench_kernels.py > "$OUT/bench_kernels.log" 2>&1; tail -30 "$OUT/bench_kernels.log"
timeout 300 python tools/bench_flow_graph.py > "$OUT/flow_graph.txt" 2>&1; tail -6 "$OUT/flow_graph.txt"
timeout 200 python tools/affine_probe.py > "$OUT/affine_probe.txt" 2>&1; tail -5 "$OUT/affine_probe.txt"
timeout 200 python tools/encoder_probe.py > "$OUT/encoder_probe.txt" 2>&1; tail -3 "$OUT/encoder_probe.txt"
rm -rf "$OUT/prof_layers"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_layers" -o layers -- \
    python tools/layer_probe.py > "$OUT/layer_probe.log" 2>&1
rm -f "$OUT"/prof_layers/*kernel_trace.csv
( for b in 64 256 1024; do timeout 300 python tools/bench_train_step.py $b 20 2>&1 | grep "^batch"; done ) > "$OUT/train_step.txt"; cat "$OUT/train_step.txt"
rm -rf "$OUT/prof_train"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_train" -o train -- \
    python tools/bench_train_step.py 8192 10 > "$OUT/train_prof.log" 2>&1
rm -f "$OUT"/prof_train/*kernel_trace.csv
