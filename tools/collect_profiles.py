"""Copy the summaries tools/refresh_profiles.sh left under gpurun_out/ into profiles/ (tracked).
Usage: python tools/collect_profiles.py [round_tag]   (default r03)"""
import csv, os, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r05"


def last_json_line(path):
    return [l for l in open(path) if l.startswith("{")][-1]


line = last_json_line(os.path.join(G, "bench.log"))
open(os.path.join(P, tag + "_bench.json"), "w").write(line)
if os.path.exists(os.path.join(G, "bench_steps20.log")):
    open(os.path.join(P, tag + "_bench_steps20.json"), "w").write(last_json_line(os.path.join(G, "bench_steps20.log")))

rows = list(csv.reader(open(os.path.join(G, "prof_bench", "bench_kernel_stats.csv"))))
with open(os.path.join(P, tag + "_bench_kernel_stats.csv"), "w", newline="") as fh:
    csv.writer(fh).writerows([rows[0]] + [[r[0][:120]] + r[1:] for r in rows[1:]])
for r in rows[1:8]:
    print(r[0][:80], r[1], r[3])

shutil.copy(os.path.join(G, "traffic.json"), os.path.join(P, "traffic.json"))
shutil.copy(os.path.join(G, "traffic.txt"), os.path.join(P, tag + "_pmc_traffic.txt"))
for src, dst in (("sweep_affine.log", "_sweep_affine.txt"), ("sweep_nll.log", "_sweep_nll.txt"), ("sweep_mixture.log", "_sweep_mixture.txt"),
                 ("sweep_mixture_bwd.log", "_sweep_mixture_bwd.txt"), ("bench_kernels.log", "_bench_kernels.txt"), ("layer_kernel_stats.csv", "_layer_kernel_stats.csv"),
                 ("flow_graph.txt", "_flow_graph.txt"), ("encoder_probe.txt", "_encoder_probe.txt"), ("encoder_ab.txt", "_encoder_ab.txt"), ("sustained_probe.txt", "_sustained_probe.txt"), ("train_step.txt", "_train_step.txt"),
                 ("ceilings/ceilings.txt", "_ceilings.txt"), ("ceilings/ceilings.json", "_ceilings.json"),
                 ("ceilings_bwd/ceilings.txt", "_ceilings_backward.txt"), ("ceilings_bwd/ceilings.json", "_ceilings_backward.json"),
                 ("ceilings_mixbwd/ceilings.txt", "_ceilings_mixture_backward.txt"),
                 ("bwd_probe.txt", "_bwd_probe.txt"), ("bwd_kernel_stats.csv", "_bwd_kernel_stats.csv"),
                 ("encoder_bwd_kernel_stats.csv", "_encoder_bwd_kernel_stats.csv"), ("encoder_bwd_variants.txt", "_encoder_bwd_variants.txt"),
                 ("autograd_overhead.txt", "_autograd_overhead.txt"), ("flow_autograd_overhead.txt", "_flow_autograd_overhead.txt"),
                 ("fp64_ceilings/fp64_ceilings.txt", "_fp64_ceilings.txt"), ("fp64_ceilings/fp64_ceilings.json", "_fp64_ceilings.json"),
                 ("mix64_ab.txt", "_mixture_fp64_token_pass_ab.txt"), ("f64_math_rates.txt", "_f64_math.txt"),
                 ("dep_latency.txt", "_dep_latency.txt"), ("mix64_iters.txt", "_mixture_fp64_newton_evaluations.txt"),
                 ("pmc_small/table.txt", "_pmc_small_mixture.txt"), ("ab_mixture_inverse.txt", "_ab_mixture_inverse.txt"), ("flow_traffic.txt", "_flow_traffic.txt"),
                 ("flow_traffic.json", "_flow_traffic.json"), ("mfma_set/mfma_util.txt", "_mfma_util_set_modelling.txt"),
                 ("mfma_set/mfma_util.json", "_mfma_util_set_modelling.json"),
                 ("train_step_flat.txt", "_train_step_flat_optimizer.txt"), ("host_profile.txt", "_host_profile_train_step.txt"),
                 ("enc/table.txt", "_pmc_encoder.txt"), ("train_lm.txt", "_train_language_modelling.txt"),
                 ("train_lm_ptb.txt", "_train_language_modelling_ptb_shape.txt"),
                 ("lm_kernel_stats.csv", "_train_language_modelling_kernel_stats.csv")):
    # only what THIS refresh produced: gpurun_out/ keeps the files of earlier rounds (a lean refresh re-measures a subset)
    if os.path.exists(os.path.join(G, src)) and os.path.getmtime(os.path.join(G, src)) > os.path.getmtime(os.path.join(G, "bench.log")) - 7200:
        shutil.copy(os.path.join(G, src), os.path.join(P, tag + dst))
tr = os.path.join(G, "prof_train", "train_kernel_stats.csv")
if os.path.exists(tr) and os.path.getmtime(tr) > os.path.getmtime(os.path.join(G, "bench.log")) - 7200:
    rows = list(csv.reader(open(tr)))
    tot = sum(float(r[2]) for r in rows[1:])
    ours = sum(float(r[2]) for r in rows[1:] if "cnf::" in r[0])
    with open(os.path.join(P, tag + "_train_step_kernel_stats.csv"), "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["# training step of the default set-modelling flow at batch 8192: total kernel time %.1f ms, kernels of this "
                    "library (cnf::) %.2f %% of it; the rest is the Transformer sub-network (hipBLASLt fp32 GEMMs, attention, "
                    "layer norm) and the optimiser" % (tot / 1e6, 100.0 * ours / tot)])
        w.writerows([rows[0]] + [[r[0][:110]] + r[1:] for r in rows[1:31]])
print(line[:400])
