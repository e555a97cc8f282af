"""Per-kernel timing of every forward hot-path kernel at the north-star shape (B=16384, N=64, D=6) with
rotating buffers: microseconds, algorithmic GB/s and fraction of the 8 TB/s HBM peak.  GPU only."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from categoricalnf_amd import ops, functional as Fn
dev = torch.device("cuda:0")
B, N, D, R = int(os.environ.get("B", 16384)), int(os.environ.get("N", 64)), int(os.environ.get("D", 6)), 4
C = 16
g = torch.Generator(device=dev).manual_seed(0)
elems = B * N * D
zs = [torch.randn(B, N, D, generator=g, device=dev) for _ in range(R)]
nn2 = [0.5 * torch.randn(B, N, 2 * D, generator=g, device=dev) for _ in range(R)]
mask = torch.cat([torch.ones(1, D // 2), torch.zeros(1, D - D // 2)], 1).to(dev)
sf = torch.zeros(D, device=dev)
bias, scales = torch.randn(1, 1, D, device=dev), 0.1 * torch.randn(1, 1, D, device=dev)
w = torch.linalg.qr(torch.randn(D, D))[0].to(dev)
sldj = torch.zeros((), device=dev)
ln = torch.full((B,), float(N), device=dev)
pad = torch.ones(B, N, 1, device=dev)
ldj = torch.zeros(B, device=dev)
cats = [torch.randint(0, C, (B, N), generator=g, device=dev) for _ in range(R)]
table = torch.randn(C, 2 * D, generator=g, device=dev)
prior = torch.log_softmax(torch.zeros(C, device=dev), 0)
us = [torch.rand(B * N, 1, D, generator=g, device=dev) for _ in range(R)]
sums = torch.zeros(2, dtype=torch.float64, device=dev)


def timeit(fn, reps=20):
    for i in range(R):
        fn(i)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for i in range(reps):
            fn(i % R)
        b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / reps * 1e3)
    return min(ts)


rows = [
    ("affine_coupling fwd", 16 * elems, lambda i: ops.affine_coupling(zs[i], nn2[i], sf, mask)),
    ("affine_coupling inv", 16 * elems, lambda i: ops.affine_coupling(zs[i], nn2[i], sf, mask, reverse=True)),
    ("actnorm fwd", 8 * elems, lambda i: ops.actnorm(zs[i], bias, scales, length=ln, channel_padding_mask=pad)),
    ("invconv fwd", 8 * elems, lambda i: ops.invconv(zs[i], w, sldj, length=ln, channel_padding_mask=pad)),
    ("ext_actnorm fwd", 16 * elems, lambda i: ops.ext_actnorm(zs[i], nn2[i])),
    ("sigmoid_flow fwd", 8 * elems, lambda i: ops.sigmoid_flow(zs[i])),
    ("prior_nll (+sum)", 4 * elems, lambda i: ops.prior_nll(zs[i], ldj, ln, sums=sums)),
    ("logistic_log_prob", 8 * elems, lambda i: ops.logistic_log_prob(zs[i])),
    ("logistic_from_uniform", 8 * elems, lambda i: ops.logistic_from_uniform(us[i])),
    ("encoder_forward C=16", (8 + 8 * D) * B * N, lambda i: ops.encoder_forward(cats[i], zs[i], table, prior)),
    ("encoder_decode C=16", (8 + 4 * D) * B * N, lambda i: ops.encoder_decode(zs[i], table, prior)),
]
gz = torch.randn(B, N, D, device=dev)
gl = torch.randn(B, device=dev)
zo, _ = ops.affine_coupling(zs[0], nn2[0], sf, mask)


def aff_bwd(i):
    zz, nn_ = zs[i].detach().requires_grad_(True), nn2[i].detach().requires_grad_(True)
    o, l = Fn.AffineCouplingFn.apply(zz, nn_, sf, None, mask, False)
    torch.autograd.backward([o, l], [gz, gl])


rows.append(("affine fwd+bwd (autograd)", (16 + 28) * elems, aff_bwd))


def aff_bwd_single_thread(i):             # what the drivers set: backward() on the calling thread (one process per GPU)
    torch.autograd.set_multithreading_enabled(False)
    try:
        aff_bwd(i)
    finally:
        torch.autograd.set_multithreading_enabled(True)


rows.append(("  ... single-threaded engine", (16 + 28) * elems, aff_bwd_single_thread))
print("shape B=%d N=%d D=%d (%.2f M elems); includes Python op overhead (~30 us host, hidden when GPU-bound)" % (B, N, D, elems / 1e6))
print("%-28s %10s %12s %8s" % ("kernel", "us", "alg GB/s", "of 8TB/s"))
for name, nbytes, fn in rows:
    t = timeit(fn)
    print("%-28s %10.1f %12.0f %7.1f%%" % (name, t, nbytes / t / 1e3, nbytes / t / 1e3 / 8000 * 100), flush=True)
