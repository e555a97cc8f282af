"""Encoder kernels, interleaved A/B on one box: steady-state us per launch of cnf_encoder_forward / cnf_encoder_decode at the
benchmark shape (B=16384, N=64, D=6) for several vocabulary sizes, and at configs[1]'s shape (B=16384, N=16, D=4, C=16).
Columns: the one-token kernel forced (cnf_set_encoder_kernel(1): 256-token tiles) -> the automatic choice (one-token kernel
on 64- / 128-token tiles); the automatic forward with class_prob_log written; the two-tokens-per-lane kernel forced
(256-token tiles).  CNF_LIB_OVERRIDE=<other build> runs the same table on another build of the library."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from categoricalnf_amd import _lib, ops
lib = _lib.load()
dev = torch.device("cuda:0")


def steady(fn, reps=30, blocks=5):
    m = [torch.cuda.Event(enable_timing=True) for _ in range(blocks + 1)]
    fn()
    torch.cuda.synchronize()
    m[0].record()
    for b in range(blocks):
        for i in range(reps):
            fn()
        m[b + 1].record()
    torch.cuda.synchronize()
    return float(np.median([m[b].elapsed_time(m[b + 1]) / reps * 1e3 for b in range(1, blocks)]))


print("shape                      C | forward forced(1) -> automatic (us) | decode forced(1) -> automatic (us) | forward + class_prob_log | forward, two tokens per lane forced")
for (B, N, D), Cs in (((16384, 64, 6), (16, 3, 9, 32, 51)), ((16384, 16, 4), (16,)), ((16384, 64, 2), (2, 3)), ((16384, 64, 8), (16,))):
    for C in Cs:
        g = torch.Generator(device=dev).manual_seed(0)
        categ = torch.randint(0, C, (B, N), generator=g, device=dev)
        table = 0.5 * torch.randn(C, 2 * D, generator=g, device=dev)
        prior = torch.log_softmax(torch.randn(C, generator=g, device=dev), 0)
        eps = ops.logistic_from_uniform(torch.rand(B * N, D, generator=g, device=dev))
        z, _, _ = ops.encoder_forward(categ, eps, table, prior, tiled=False)
        res = {}
        for rnd in range(2):
            for which in (1, 0):
                lib.cnf_set_encoder_kernel(which)
                f = steady(lambda: ops.encoder_forward(categ, eps, table, prior, tiled=False))
                d = steady(lambda: ops.encoder_decode(z, table, prior, tiled=False))
                res.setdefault(which, []).append((f, d))
        lib.cnf_set_encoder_kernel(0)
        fc = steady(lambda: ops.encoder_forward(categ, eps, table, prior, want_class_prob=True, tiled=False))
        lib.cnf_set_encoder_kernel(2)
        fp = min(steady(lambda: ops.encoder_forward(categ, eps, table, prior, tiled=False)) for _ in range(2))
        lib.cnf_set_encoder_kernel(0)
        r2 = np.min(np.array(res[1]), 0)
        r3 = np.min(np.array(res[0]), 0)
        print("B=%5d N=%3d D=%d  C=%3d | %7.2f -> %7.2f | %7.2f -> %7.2f | %7.2f | %7.2f" % (B, N, D, C, r2[0], r3[0], r2[1], r3[1], fc, fp), flush=True)
