#!/bin/bash
# Round 3, one GPU call: new tests, the training-graph soak with its bisection switches, MFMA utilisation of the
# language-modelling and graph-colouring training steps.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
cd "$ROOT"
timeout 900 python -m pytest tests -m gpu -x -q -k "graph_cnf or rehearses or statistics_meet or encoder_backward or self_launches" > "$OUT/r3_new_tests.log" 2>&1; tail -4 "$OUT/r3_new_tests.log"
timeout 600 python tools/graph_train_soak.py --steps 3000 > "$OUT/r3_soak_radam.log" 2>&1; tail -4 "$OUT/r3_soak_radam.log"
if ! grep -q "SOAK OK" "$OUT/r3_soak_radam.log"; then
  timeout 300 python tools/graph_train_soak.py --steps 1500 --optimizer sgd > "$OUT/r3_soak_sgd.log" 2>&1; tail -2 "$OUT/r3_soak_sgd.log"
  timeout 300 python tools/graph_train_soak.py --steps 1500 --optimizer adam > "$OUT/r3_soak_adam.log" 2>&1; tail -2 "$OUT/r3_soak_adam.log"
  timeout 300 python tools/graph_train_soak.py --steps 1500 --no_clip > "$OUT/r3_soak_noclip.log" 2>&1; tail -2 "$OUT/r3_soak_noclip.log"
  timeout 300 python tools/graph_train_soak.py --steps 1500 --sync_each > "$OUT/r3_soak_sync.log" 2>&1; tail -2 "$OUT/r3_soak_sync.log"
  timeout 300 python tools/graph_train_soak.py --steps 1500 --capture_stream warmup > "$OUT/r3_soak_warmstream.log" 2>&1; tail -2 "$OUT/r3_soak_warmstream.log"
fi
bash tools/mfma_util.sh mfma_lm python -m categoricalnf_amd.experiments.run_language_modeling --max_iterations 12 --print_freq 20 --eval_freq 100000 --num_val 128 > "$OUT/mfma_lm.log" 2>&1; tail -12 "$OUT/mfma_lm.log" | cut -c1-200
bash tools/mfma_util.sh mfma_lm_ptb python -m categoricalnf_amd.experiments.run_language_modeling --variable_length --max_seq_len 288 --vocab_size 51 --coupling_num_mixtures 51 --coupling_hidden_layers 1 --coupling_dropout 0.3 --coupling_input_dropout 0.1 --max_iterations 12 --print_freq 20 --eval_freq 100000 --num_val 128 > "$OUT/mfma_lm_ptb.log" 2>&1; tail -12 "$OUT/mfma_lm_ptb.log" | cut -c1-200
bash tools/mfma_util.sh mfma_gc python -m categoricalnf_amd.experiments.run_graph_coloring --dataset tiny_3 --generate_data --num_graphs 4000 --max_iterations 12 --print_freq 20 --eval_freq 100000 --data_root /tmp/gc_data/ > "$OUT/mfma_gc.log" 2>&1; tail -12 "$OUT/mfma_gc.log" | cut -c1-200
bash tools/mfma_util.sh mfma_gc_large python -m categoricalnf_amd.experiments.run_graph_coloring --dataset large_3 --generate_data --num_graphs 2000 --batch_size 128 --encoding_dim 6 --coupling_num_mixtures 16 --max_iterations 12 --print_freq 20 --eval_freq 100000 --data_root /tmp/gc_data/ > "$OUT/mfma_gc_large.log" 2>&1; tail -12 "$OUT/mfma_gc_large.log" | cut -c1-200
