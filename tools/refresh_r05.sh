#!/bin/bash
# Round 5's part of tools/refresh_profiles.sh: what this round's changes can have moved (the bench line and its rocprofv3 summary, the
# dominant kernel's PMC traffic, the backward kernels, the encoder backward, the mixture backward).  The forward-side tables of round 4
# (sweeps, ceilings, flow traffic, encoder probe) measure kernels this round did not touch and are not repeated.
#   gpurun --timeout 2400 -- 'bash tools/refresh_r05.sh'   then   python tools/collect_profiles.py r05
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp; cd "$ROOT"
rm -rf "$OUT/prof_bench" "$OUT/pmc_fetch" "$OUT/pmc_write"
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_bench" -o bench -- python bench.py --no-cpu-baseline > "$OUT/bench_prof.log" 2>&1
tail -1 "$OUT/bench_prof.log" | cut -c1-300
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o pmc -- python tools/pmc_workload.py > "$OUT/pmc_fetch.log" 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -o pmc -- python tools/pmc_workload.py > "$OUT/pmc_write.log" 2>&1
python tools/pmc_summarize.py "$OUT/pmc_fetch" "$OUT/pmc_write" "$OUT/traffic.json" "$OUT/traffic.txt" | tail -8
rm -f "$OUT"/pmc_fetch/*kernel_trace.csv "$OUT"/pmc_write/*kernel_trace.csv "$OUT"/prof_bench/*kernel_trace.csv
cp "$OUT/traffic.json" "$ROOT/profiles/traffic.json"
# the reference-precision (fp64) mixture kernels: instruction counts for the bench rows' fp64 issue fractions, the A/B against the
# round-1 fp64 kernel, the cost of the fp64 functions
bash tools/pmc_fp64.sh fp64_ceilings > "$OUT/fp64_ceilings.log" 2>&1; tail -12 "$OUT/fp64_ceilings.log" | cut -c1-200
cp "$OUT/fp64_ceilings/fp64_ceilings.json" "$ROOT/profiles/r05_fp64_ceilings.json"
timeout 300 python tools/mix64_tok_ab.py > "$OUT/mix64_ab.txt" 2>&1; NOSF=1 SHAPES="configs[1],S*" timeout 300 python tools/mix64_tok_ab.py >> "$OUT/mix64_ab.txt" 2>&1; tail -4 "$OUT/mix64_ab.txt" | cut -c1-200
timeout 300 python tools/f64_math_rates.py > "$OUT/f64_math_rates.txt" 2>&1; head -6 "$OUT/f64_math_rates.txt"
timeout 500 python bench.py > "$OUT/bench.log" 2>&1; tail -1 "$OUT/bench.log" | cut -c1-400
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-mixture > "$OUT/bench_steps20.log" 2>&1; tail -1 "$OUT/bench_steps20.log" | cut -c1-200
timeout 300 python tools/sweep_mixture_bwd.py > "$OUT/sweep_mixture_bwd.log" 2>&1; tail -7 "$OUT/sweep_mixture_bwd.log" | cut -c1-200
timeout 300 python tools/bwd_probe.py --sweep > "$OUT/bwd_probe.txt" 2>&1; head -24 "$OUT/bwd_probe.txt"
bash tools/pmc_ceilings.sh ceilings_mixbwd python tools/pmc_mixture_bwd_workload.py > "$OUT/ceilings_mixbwd.log" 2>&1; tail -4 "$OUT/ceilings_mixbwd.log"
rm -rf "$OUT/prof_bwd"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_bwd" -o bwd -- python tools/bwd_probe.py --reps 30 > /dev/null 2>&1
python tools/summarize_kernel_stats.py "$OUT/prof_bwd/bwd_kernel_stats.csv" "$OUT/bwd_kernel_stats.csv" "tools/bwd_probe.py --reps 30 (every streaming backward kernel at B=16384, N=64, D=6 on four rotating buffer sets; kernel durations by themselves: the start-to-start table of r05_bwd_probe.txt includes the reduction launch behind a kernel)" 40 | head -3
rm -f "$OUT"/prof_bwd/*kernel_trace.csv
rm -rf "$OUT/prof_encbwd"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_encbwd" -o encbwd -- python tools/pmc_encoder_bwd_workload.py 16,51 1,10 > /dev/null 2>&1
python tools/summarize_kernel_stats.py "$OUT/prof_encbwd/encbwd_kernel_stats.csv" "$OUT/encoder_bwd_kernel_stats.csv" "tools/pmc_encoder_bwd_workload.py 16,51 1,10 (encoder forward once + 10 calls each of cnf_encoder_forward_bwd_tiled forced onto the two passes and of cnf_encoder_forward_bwd_cpl with the library's own choice, at 1 048 576 tokens, D = 6, C = 16 and 51: their launches one by one)" 12 | head -3
rm -f "$OUT"/prof_encbwd/*kernel_trace.csv
timeout 200 python tools/encoder_bwd_variants.py 16384,64,6,16 16384,64,6,51 1024,64,6,16 64,64,6,27 2>/dev/null > "$OUT/encoder_bwd_variants.txt"; tail -12 "$OUT/encoder_bwd_variants.txt"
( timeout 200 python tools/flow_autograd_overhead.py 2>&1 | grep -v amdgpu.ids ) > "$OUT/flow_autograd_overhead.txt"; tail -6 "$OUT/flow_autograd_overhead.txt"
( for b in 64 1024; do timeout 300 python tools/bench_train_step.py $b 20 2>&1 | grep "^batch"; done ) > "$OUT/train_step.txt"; cat "$OUT/train_step.txt"
find "$OUT" -name "*counter_collection.csv" -size +4M -delete
