"""Host cost of ONE differentiable layer call (torch.autograd.Function around the forward and the backward kernel) against the
GPU time of its kernels, at the benchmark shape: wall per forward + backward, kernel time (dispatch-bound events), and a
cProfile of the host side (no sync inside the loop).  GPU only.   python tools/autograd_overhead.py [--profile]"""
import argparse, cProfile, ctypes, os, pstats, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from categoricalnf_amd import _lib, functional as Fn, ops as ops_
ap = argparse.ArgumentParser(); ap.add_argument("--profile", action="store_true"); ap.add_argument("--B", type=int, default=16384)
ap.add_argument("--single_thread", action="store_true", help="torch.autograd.set_multithreading_enabled(False): backward() runs on the calling thread")
args = ap.parse_args()
if args.single_thread:
    torch.autograd.set_multithreading_enabled(False)
dev = torch.device("cuda:0"); lib = _lib.load()
B, N, D, R = args.B, 64, 6, 4
g = torch.Generator(device=dev).manual_seed(0)
zs = [torch.randn(B, N, D, generator=g, device=dev) for _ in range(R)]
nns = [0.5 * torch.randn(B, N, 2 * D, generator=g, device=dev) for _ in range(R)]
mask = torch.tensor([[1., 1., 1., 0., 0., 0.]], device=dev)
sf = torch.zeros(D, device=dev, requires_grad=True)
bias, scales = torch.randn(1, 1, D, device=dev, requires_grad=True), (0.1 * torch.randn(1, 1, D, device=dev)).requires_grad_(True)
gz, gl = torch.randn(B, N, D, device=dev), torch.randn(B, device=dev)


def affine(i):
    z, nn_ = zs[i].detach().requires_grad_(True), nns[i].detach().requires_grad_(True)
    o, l = Fn.AffineCouplingFn.apply(z, nn_, sf, None, mask, False)
    torch.autograd.backward([o, l], [gz, gl])


def actnorm(i):
    z = zs[i].detach().requires_grad_(True)
    o, l = Fn.ActNormFn.apply(z, bias, scales, None, None, None, False)
    torch.autograd.backward([o, l], [gz, gl])


ws = torch.empty(int(lib.cnf_bwd_workspace_floats(2 * D + 1)), device=dev)
from categoricalnf_amd.ops import _ptr, _stream
o1, o2, lo = torch.empty(B, N, D, device=dev), torch.empty(B, N, 2 * D, device=dev), torch.empty(B, device=dev)
g_sf, g_b, g_s = torch.empty(D, device=dev), torch.empty(D, device=dev), torch.empty(D, device=dev)
fwd_aff = [ops_.affine_coupling_launch(zs[r], nns[r], sf.detach(), mask, o1, lo) for r in range(R)]
flags = _ptr(ops_.flag_word(dev))
bias_c, scales_c = bias.detach().reshape(-1).contiguous(), scales.detach().reshape(-1).contiguous()


def kernels_affine(i):                # the same kernels on pre-bound arguments: a few microseconds of host time per launch
    fwd_aff[i]()
    lib.cnf_affine_coupling_bwd(_ptr(zs[i]), _ptr(nns[i]), _ptr(sf), _ptr(mask), 1, D, _ptr(gz), _ptr(gl), _ptr(o1), _ptr(o2), _ptr(g_sf), _ptr(ws),
                                B, N, D, 0, _stream(dev))


def kernels_actnorm(i):
    lib.cnf_actnorm(_ptr(zs[i]), _ptr(bias_c), _ptr(scales_c), None, None, None, _ptr(o1), _ptr(lo), B, N, D, 0, flags, _stream(dev))
    lib.cnf_actnorm_bwd(_ptr(zs[i]), _ptr(bias_c), _ptr(scales_c), None, None, _ptr(gz), _ptr(gl), _ptr(o1), _ptr(g_b), _ptr(g_s), _ptr(ws), B, N, D, 0, _stream(dev))


def gpu_time(fn):
    for i in range(8):
        fn(i % R)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(200):
        fn(i % R)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / 200 * 1e3


for name, fn, kfn in (("affine coupling fwd + bwd", affine, kernels_affine), ("actnorm fwd + bwd", actnorm, kernels_actnorm)):
    for i in range(20):
        fn(i % R)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        for i in range(100):
            fn(i % R)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 100 * 1e6)
        enq = (t1 - t0) / 100 * 1e6
    kern = gpu_time(kfn)
    print("%-28s wall %.1f us per call through torch.autograd (host enqueue %.1f us); the same kernels back to back on pre-bound arguments %.1f us"
          " -> wall / kernels = %.2f" % (name, best, enq, kern, best / kern), flush=True)
    if args.profile:
        pr = cProfile.Profile(); pr.enable()
        for i in range(300):
            fn(i % R)
        pr.disable(); torch.cuda.synchronize()
        st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(14)
print("(a single layer per backward() call is the worst case: the autograd engine hands every backward() to its device thread and "
      "waits for it, ~2 thread wake-ups per call; in a flow the whole chain of layers runs in ONE such hand-off, and a captured "
      "step (--graph_step) has no host cost per layer at all)")
