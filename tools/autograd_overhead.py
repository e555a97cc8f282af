"""Host cost of ONE differentiable layer call (torch.autograd.Function around the forward and the backward kernel) against the
GPU time of its kernels, at the benchmark shape: wall per forward + backward, kernel time (dispatch-bound events), and a
cProfile of the host side (no sync inside the loop).  GPU only.   python tools/autograd_overhead.py [--profile]"""
import argparse, cProfile, ctypes, os, pstats, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from categoricalnf_amd import _lib, functional as Fn
ap = argparse.ArgumentParser(); ap.add_argument("--profile", action="store_true"); ap.add_argument("--B", type=int, default=16384)
args = ap.parse_args()
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


for name, fn, nk in (("affine coupling fwd + bwd", affine, 3), ("actnorm fwd + bwd", actnorm, 3)):
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
    lib.cnf_prof_arm(nk * 10)
    for i in range(10):
        fn(i % R)
    buf = (ctypes.c_float * 64)()
    n = lib.cnf_prof_collect(buf, 64)
    kern = sum(buf[i] for i in range(n)) / 10 * 1e3
    print("%-28s wall %.1f us per call (host enqueue %.1f us), its %d kernels %.1f us -> wall / kernels = %.2f" % (name, best, enq, n // 10, kern, best / kern), flush=True)
    if args.profile:
        pr = cProfile.Profile(); pr.enable()
        for i in range(300):
            fn(i % R)
        pr.disable(); torch.cuda.synchronize()
        st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(14)
