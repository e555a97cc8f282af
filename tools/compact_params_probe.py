"""The last Linear of a coupling sub-network on the reference layout (all D parameter blocks) and on the compact layout (the
transformed channels' rows only: RowSlicedLinear), forward and backward GEMM time, at the set-modelling shapes.
    python tools/compact_params_probe.py"""
import os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
dev = torch.device("cuda:0")


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / reps * 1e3)
    return sorted(ts)[len(ts) // 2]


for name, B, N, D, K, hidden in (("configs[1]-like (|S|=16, D=4)", 16384, 16, 4, 8, 256), ("S* (N=64, D=6)", 16384, 64, 6, 8, 256),
                                 ("training batch 1024 x 16, D=4", 1024, 16, 4, 8, 256)):
    P, DA = 2 + 3 * K, D - D // 2
    h = torch.randn(N, B, hidden, device=dev, requires_grad=True)
    w = torch.randn(D * P, hidden, device=dev, requires_grad=True)
    b = torch.randn(D * P, device=dev, requires_grad=True)
    rows = slice((D - DA) * P, D * P)
    out = []
    for layout, fwd in (("reference [.., %d]" % (D * P), lambda: F.linear(h, w, b)), ("compact [.., %d]" % (DA * P), lambda: F.linear(h, w[rows], b[rows]))):
        y = fwd()
        g = torch.randn_like(y)

        def bwd():
            h.grad = w.grad = b.grad = None
            fwd().backward(g)
        with torch.no_grad():
            tf = timeit(fwd)
        tfb = timeit(bwd)
        out.append("%s: forward %.1f us, forward + backward %.1f us" % (layout, tf, tfb))
    print("%-34s %d tokens, hidden %d | %s" % (name, B * N, hidden, " | ".join(out)), flush=True)
