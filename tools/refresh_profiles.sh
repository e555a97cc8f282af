#!/bin/bash
# Re-measure everything profiles/ holds, on the GPU box.  Run from the repo root:
#   gpurun --timeout 2400 -- 'bash tools/refresh_profiles.sh'        (CNF_REFRESH_QUICK=1: the kernel measurements only)
# then, back in the build container:  python tools/collect_profiles.py r05
# The --pmc passes are separate rocprofv3 runs with --kernel-trace only (never combined with sys/hip traces).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp; cd "$ROOT"

rm -rf "$OUT/prof_bench" "$OUT/pmc_fetch" "$OUT/pmc_write"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_bench" -o bench -- \
    python bench.py --no-cpu-baseline > "$OUT/bench_prof.log" 2>&1
tail -1 "$OUT/bench_prof.log" | cut -c1-300
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o pmc -- \
    python tools/pmc_workload.py > "$OUT/pmc_fetch.log" 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -o pmc -- \
    python tools/pmc_workload.py > "$OUT/pmc_write.log" 2>&1
python tools/pmc_summarize.py "$OUT/pmc_fetch" "$OUT/pmc_write" "$OUT/traffic.json" "$OUT/traffic.txt" | tail -8
rm -f "$OUT"/pmc_fetch/*kernel_trace.csv "$OUT"/pmc_write/*kernel_trace.csv "$OUT"/prof_bench/*kernel_trace.csv
# the bench line reports the PMC traffic only when it was collected on the kernel sources that are built: refresh it first
cp "$OUT/traffic.json" "$ROOT/profiles/traffic.json"
timeout 300 python bench.py > "$OUT/bench.log" 2>&1; tail -1 "$OUT/bench.log" | cut -c1-400
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-mixture > "$OUT/bench_steps20.log" 2>&1; tail -1 "$OUT/bench_steps20.log" | cut -c1-200


timeout 300 python tools/sweep_affine.py > "$OUT/sweep_affine.log" 2>&1; tail -4 "$OUT/sweep_affine.log"
timeout 300 python tools/sweep_nll.py > "$OUT/sweep_nll.log" 2>&1; tail -4 "$OUT/sweep_nll.log"
timeout 300 python tools/sweep_mixture.py > "$OUT/sweep_mixture.log" 2>&1; tail -6 "$OUT/sweep_mixture.log" | cut -c1-200
timeout 300 python tools/sweep_mixture_bwd.py > "$OUT/sweep_mixture_bwd.log" 2>&1; tail -7 "$OUT/sweep_mixture_bwd.log" | cut -c1-200
timeout 300 python tools/bench_kernels.py > "$OUT/bench_kernels.log" 2>&1; tail -14 "$OUT/bench_kernels.log"
# the same table's kernels by themselves (the table's microseconds include ~15 us of host time per op call, which paces every
# kernel shorter than that): rocprofv3 durations per kernel
rm -rf "$OUT/prof_layers"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_layers" -o layers -- python tools/bench_kernels.py > /dev/null 2>&1
python tools/summarize_kernel_stats.py "$OUT/prof_layers/layers_kernel_stats.csv" "$OUT/layer_kernel_stats.csv" "tools/bench_kernels.py (every forward layer kernel at B=16384, N=64, D=6 on four rotating input sets = 100 MB: inputs are re-read from the 256 MB memory-side cache, so these are kernel durations, not HBM figures)" 20 | head -3
rm -f "$OUT"/prof_layers/*kernel_trace.csv
timeout 300 python tools/bench_flow_graph.py > "$OUT/flow_graph.txt" 2>&1; tail -6 "$OUT/flow_graph.txt"
timeout 400 python tools/encoder_probe.py > "$OUT/encoder_probe.txt" 2>&1; tail -7 "$OUT/encoder_probe.txt"
timeout 300 python tools/encoder_ab.py 2>/dev/null > "$OUT/encoder_ab.txt"; tail -9 "$OUT/encoder_ab.txt" | cut -c1-160
timeout 200 python tools/encoder_fused_sampler.py 2>/dev/null > "$OUT/encoder_fused_sampler.txt"; tail -5 "$OUT/encoder_fused_sampler.txt"
timeout 200 python tools/sustained_probe.py > "$OUT/sustained_probe.txt" 2>&1; tail -6 "$OUT/sustained_probe.txt"
bash tools/pmc_ceilings.sh ceilings > "$OUT/ceilings.log" 2>&1; tail -10 "$OUT/ceilings.log"
# round 4: the backward kernels — start-to-start table + stream ceiling, counters / calibrated VALU fraction, rocprofv3 durations
timeout 300 python tools/bwd_probe.py --sweep > "$OUT/bwd_probe.txt" 2>&1; head -24 "$OUT/bwd_probe.txt"
bash tools/pmc_ceilings.sh ceilings_bwd python tools/bwd_probe.py --pmc > "$OUT/ceilings_bwd.log" 2>&1; tail -14 "$OUT/ceilings_bwd.log"
bash tools/pmc_ceilings.sh ceilings_mixbwd python tools/pmc_mixture_bwd_workload.py > "$OUT/ceilings_mixbwd.log" 2>&1; tail -4 "$OUT/ceilings_mixbwd.log"
rm -rf "$OUT/prof_bwd"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_bwd" -o bwd -- python tools/bwd_probe.py --reps 30 > /dev/null 2>&1
python tools/summarize_kernel_stats.py "$OUT/prof_bwd/bwd_kernel_stats.csv" "$OUT/bwd_kernel_stats.csv" "tools/bwd_probe.py --reps 30 (every streaming backward kernel at B=16384, N=64, D=6 on four rotating buffer sets; kernel durations by themselves: the start-to-start table of r04_bwd_probe.txt includes the reduction launch behind a kernel)" 40 | head -3
rm -f "$OUT"/prof_bwd/*kernel_trace.csv
# the encoder backward per kernel at the benchmark token count, C = 16 and 51: the two passes (variant 1) and what
# cnf_encoder_forward_bwd_cpl picks by shape (variant 10: the pair kernel at 16 classes, the two passes at 51)
rm -rf "$OUT/prof_encbwd"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_encbwd" -o encbwd -- python tools/pmc_encoder_bwd_workload.py 16,51 1,10 > /dev/null 2>&1
python tools/summarize_kernel_stats.py "$OUT/prof_encbwd/encbwd_kernel_stats.csv" "$OUT/encoder_bwd_kernel_stats.csv" "tools/pmc_encoder_bwd_workload.py 16,51 1,10 (encoder forward once + 10 calls each of cnf_encoder_forward_bwd_tiled forced onto the two passes and of cnf_encoder_forward_bwd_cpl with the library's own choice, at 1 048 576 tokens, D = 6, C = 16 and 51: their launches one by one)" 12 | head -3
rm -f "$OUT"/prof_encbwd/*kernel_trace.csv
timeout 200 python tools/encoder_bwd_variants.py 16384,64,6,16 16384,64,6,51 1024,64,6,16 64,64,6,27 2>/dev/null > "$OUT/encoder_bwd_variants.txt"; tail -12 "$OUT/encoder_bwd_variants.txt"
( timeout 200 python tools/autograd_overhead.py 2>&1 | grep -v amdgpu.ids; echo; echo "== --single_thread (torch.autograd.set_multithreading_enabled(False): backward() on the calling thread, what the three drivers set)"; timeout 200 python tools/autograd_overhead.py --single_thread 2>&1 | grep "wall" ) > "$OUT/autograd_overhead.txt"; tail -3 "$OUT/autograd_overhead.txt"
bash tools/pmc_passes.sh pmc_small python tools/pmc_small_mixture.py > "$OUT/pmc_small.log" 2>&1
bash tools/pmc_passes.sh flow_fused python tools/flow_traffic_workload.py fused > /dev/null 2>&1
bash tools/pmc_passes.sh flow_unfused python tools/flow_traffic_workload.py unfused > /dev/null 2>&1
python tools/flow_traffic.py "$OUT/flow_fused" "$OUT/flow_unfused" "$OUT/flow_traffic.json" > "$OUT/flow_traffic.txt" 2>&1; head -12 "$OUT/flow_traffic.txt"
timeout 300 python tools/ab_mixture_inverse.py 2>&1 | grep -v amdgpu.ids > "$OUT/ab_mixture_inverse.txt"; tail -5 "$OUT/ab_mixture_inverse.txt" | cut -c1-200
# CNF_REFRESH_QUICK=1: kernels only — skip the training-step, host-profile, MFMA and training-log sections below
if [ -n "${CNF_REFRESH_QUICK:-}" ]; then exit 0; fi
( for b in 64 256 1024; do timeout 300 python tools/bench_train_step.py $b 20 2>&1 | grep "^batch"; done ) > "$OUT/train_step.txt"; cat "$OUT/train_step.txt"
( for b in 64 1024; do timeout 300 python tools/bench_train_step.py $b 30 2>&1 | grep "^batch"; timeout 300 python tools/bench_train_step.py $b 30 flat 2>&1 | grep "^batch"; done ) > "$OUT/train_step_flat.txt"; cat "$OUT/train_step_flat.txt"
timeout 200 python tools/host_profile_train_step.py 64 30 > "$OUT/host_profile.txt" 2>&1; head -6 "$OUT/host_profile.txt"
bash tools/pmc_passes.sh enc python tools/pmc_encoder_workload.py > "$OUT/pmc_enc.log" 2>&1
bash tools/mfma_util.sh mfma_set python tools/bench_train_step.py 8192 6 > "$OUT/mfma.log" 2>&1; tail -3 "$OUT/mfma.log" | cut -c1-300
rm -rf "$OUT/prof_train"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_train" -o train -- \
    python tools/bench_train_step.py 8192 10 > "$OUT/train_prof.log" 2>&1
rm -f "$OUT"/prof_train/*kernel_trace.csv
find "$OUT" -name "*counter_collection.csv" -size +4M -delete
# training logs of the drivers (each stopped by its timeout; the logs are flushed line by line)
timeout 330 python -m categoricalnf_amd.experiments.run_language_modeling --eval_freq 500 --print_freq 100 > "$OUT/train_lm.txt" 2>&1; grep validation "$OUT/train_lm.txt" | tail -2
timeout 330 python -m categoricalnf_amd.experiments.run_language_modeling --variable_length --max_seq_len 288 --vocab_size 51 --coupling_num_mixtures 51 --coupling_hidden_layers 1 --coupling_dropout 0.3 --coupling_input_dropout 0.1 --eval_freq 500 --print_freq 100 > "$OUT/train_lm_ptb.txt" 2>&1; grep validation "$OUT/train_lm_ptb.txt" | tail -2
rm -rf "$OUT/prof_lm"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_lm" -o lm -- \
    python -m categoricalnf_amd.experiments.run_language_modeling --max_iterations 40 --print_freq 20 --eval_freq 100000 --num_val 128 > "$OUT/prof_lm.log" 2>&1
python tools/summarize_kernel_stats.py "$OUT/prof_lm/lm_kernel_stats.csv" "$OUT/lm_kernel_stats.csv" "40 training steps of the language-modelling flow (text8 recipe, batch 128 x 256 characters) + data-dependent init + one evaluation of 128 sentences" 40 | head -3
rm -f "$OUT"/prof_lm/*kernel_trace.csv
