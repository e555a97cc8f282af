"""Steady-state time of the categorical-encoder kernels at the benchmark shape (B=16384, N=64, D=6, C=16)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from categoricalnf_amd import _lib
if os.environ.get("CNF_LIB_OVERRIDE"):
    _lib.LIB_PATH = os.environ["CNF_LIB_OVERRIDE"]
from categoricalnf_amd import ops
dev = torch.device("cuda:0")
B, N, D = 16384, 64, 6
def steady(fn, reps=30, blocks=4):
    m = [torch.cuda.Event(enable_timing=True) for _ in range(blocks + 1)]
    m[0].record()
    for b in range(blocks):
        for i in range(reps):
            fn()
        m[b + 1].record()
    torch.cuda.synchronize()
    return float(np.median([m[b].elapsed_time(m[b + 1]) / reps * 1e3 for b in range(1, blocks)]))
for C in (16, 3, 51):
    g = torch.Generator(device=dev).manual_seed(0)
    categ = torch.randint(0, C, (B, N), generator=g, device=dev)
    table = 0.5 * torch.randn(C, 2 * D, generator=g, device=dev)
    prior = torch.log_softmax(torch.randn(C, generator=g, device=dev), 0)
    u = torch.rand(B * N, D, generator=g, device=dev)
    eps = ops.logistic_from_uniform(u)
    z, ldj, _ = ops.encoder_forward(categ, eps, table, prior)
    t_s = steady(lambda: ops.logistic_from_uniform(u))
    t_f = steady(lambda: ops.encoder_forward(categ, eps, table, prior))
    t_d = steady(lambda: ops.encoder_decode(z, table, prior))
    ok = (ops.encoder_decode(z, table, prior) == categ).float().mean().item()
    print("C=%2d  sample %6.1f us | forward %6.1f us | decode %6.1f us | decode==categ %.4f" % (C, t_s, t_f, t_d, ok), flush=True)
