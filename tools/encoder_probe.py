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

if os.environ.get("CNF_PROBE_SHORT"):
    sys.exit(0)

# ---- LDS-resident vs class-tiled kernels at the same shapes, backward included; large vocabularies ----------------
from categoricalnf_amd import functional as Fn


def bwd_time(categ, eps, table, prior, tiled):
    tg = table.clone().requires_grad_()
    gz, gl = torch.randn(categ.shape[0], categ.shape[1], eps.shape[-1], device=dev), torch.randn(categ.shape[0], device=dev)

    def step():
        z, ldj, _ = Fn.EncoderForwardFn.apply(tg, categ, eps, prior, None, 1.0, False, tiled)
        torch.autograd.backward([z, ldj], [gz, gl])
        tg.grad = None
    return steady(step, reps=10)


print("\nB*N tokens x C classes, D=6: forward / decode / forward+backward, us (LDS-resident | default: forward by size, tiled backward | class-tiled)")
for T_B, T_N, C in ((16384, 64, 16), (16384, 64, 51), (4096, 64, 32), (4096, 64, 64), (4096, 64, 96), (4096, 64, 128), (2048, 64, 160),
                    (512, 64, 2000), (128, 64, 10000), (128, 288, 10000)):
    g = torch.Generator(device=dev).manual_seed(1)
    categ = torch.randint(0, C, (T_B, T_N), generator=g, device=dev)
    table = 0.5 * torch.randn(C, 2 * D, generator=g, device=dev)
    prior = torch.log_softmax(torch.randn(C, generator=g, device=dev), 0)
    eps = ops.logistic_from_uniform(torch.rand(T_B * T_N, D, generator=g, device=dev))
    z, _, _ = ops.encoder_forward(categ, eps, table, prior)
    row = []
    for tiled in ((False, None, True) if ops.encoder_fused_supported(C, D) else (True,)):
        f = steady(lambda: ops.encoder_forward(categ, eps, table, prior, tiled=tiled), reps=10)
        d = steady(lambda: ops.encoder_decode(z, table, prior, tiled=tiled), reps=10)
        fb = bwd_time(categ, eps, table, prior, tiled)
        row.append("%8.1f /%8.1f /%9.1f" % (f, d, fb))
    evals = T_B * T_N * C * D
    print("T=%8d C=%6d  %s   (%.2f G class-channel evaluations per pass)" % (T_B * T_N, C, "  |  ".join(row), evals / 1e9), flush=True)
