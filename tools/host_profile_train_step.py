"""Where the HOST spends a training step of the default set-modelling flow at a small batch (the step is host-paced there):
cProfile over `steps` steps with no device sync inside, split into this package's Python (ops.py / functional.py / layers:
argument checks, ctypes marshalling, launches), torch's forward ops of the sub-networks, autograd's backward, the optimiser
and gradient clipping.  python tools/host_profile_train_step.py [batch] [steps]"""
import cProfile, os, pstats, sys, io, contextlib, time
sys.argv = [sys.argv[0]] + (sys.argv[1:] or ["64", "30"])
B, steps = int(sys.argv[1]), int(sys.argv[2])
sys.argv = [sys.argv[0], str(B), "5"]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = open(os.path.join(root, "tools", "bench_train_step.py")).read()
ns = {"__name__": "bench_train_step", "__file__": os.path.join(root, "tools", "bench_train_step.py")}
with contextlib.redirect_stdout(io.StringIO()):
    exec(compile(src, ns["__file__"], "exec"), ns)                 # builds the model, warms up, defines step()
import torch
step = ns["step"]
torch.cuda.synchronize()
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
for i in range(steps):
    step(i)
pr.disable()
host = (time.perf_counter() - t0) / steps
torch.cuda.synchronize()
total = (time.perf_counter() - t0) / steps
st = pstats.Stats(pr)
pkg = os.path.join(root, "categoricalnf_amd")
buckets = {"this package (own time of its Python functions)": 0.0, "ctypes calls into libcnf_hip.so": 0.0,
           "autograd engine (run_backward, incl. our backward Functions' callers)": 0.0, "optimizer + clip_grad_norm_ (own + callees)": 0.0}
for (fn, line, name), (cc, nc, tt, ct, callers) in st.stats.items():
    if fn.startswith(pkg):
        buckets["this package (own time of its Python functions)"] += tt
    if "_FuncPtr" in name or ("ctypes" in fn and tt > 0):
        buckets["ctypes calls into libcnf_hip.so"] += tt
    if name == "run_backward" or "_engine_run_backward" in name:
        buckets["autograd engine (run_backward, incl. our backward Functions' callers)"] = max(
            buckets["autograd engine (run_backward, incl. our backward Functions' callers)"], ct)
    if (fn.endswith("radam.py") and name == "step") or name == "clip_grad_norm_":
        buckets["optimizer + clip_grad_norm_ (own + callees)"] += ct
print("batch %d: host enqueues a step in %.2f ms (with the final device sync: %.2f ms per step)" % (B, host * 1e3, total * 1e3))
for k, v in buckets.items():
    print("  %-75s %7.2f ms / step" % (k, v / steps * 1e3))
out = io.StringIO()
pstats.Stats(pr, stream=out).sort_stats("tottime").print_stats(18)
print("\n".join(l[:170] for l in out.getvalue().splitlines() if l.strip())[-4500:])
