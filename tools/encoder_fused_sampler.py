"""LogisticDistribution.sample + the encoder forward as two calls (cnf_logistic_from_uniform, cnf_encoder_forward) against the fused
kernel (cnf_encoder_forward_sampled): steady-state us per pair of calls / per call (profiles/r03_encoder_fused_sampler.txt)."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from categoricalnf_amd import _lib, ops
lib = _lib.load(); dev = torch.device("cuda:0")
def steady(fn, reps=30, blocks=5):
    m = [torch.cuda.Event(enable_timing=True) for _ in range(blocks + 1)]
    fn(); torch.cuda.synchronize(); m[0].record()
    for b in range(blocks):
        for i in range(reps): fn()
        m[b + 1].record()
    torch.cuda.synchronize()
    return float(np.median([m[b].elapsed_time(m[b + 1]) / reps * 1e3 for b in range(1, blocks)]))
for (B, N, D, C) in ((16384, 64, 6, 16), (16384, 64, 6, 51), (16384, 16, 4, 16), (1024, 16, 4, 16), (128, 288, 3, 51)):
    g = torch.Generator(device=dev).manual_seed(0)
    categ = torch.randint(0, C, (B, N), generator=g, device=dev)
    table = 0.5 * torch.randn(C, 2 * D, generator=g, device=dev)
    prior = torch.log_softmax(torch.randn(C, generator=g, device=dev), 0)
    u = torch.rand(B * N, D, generator=g, device=dev)
    two = lambda: ops.encoder_forward(categ, ops.logistic_from_uniform(u), table, prior)
    one = lambda: ops.encoder_forward(categ, u, table, prior, uniform_squeeze=1e-4)
    a = min(steady(two) for _ in range(2)); b = min(steady(one) for _ in range(2))
    print("B=%5d N=%3d D=%d C=%3d | sampler + encoder forward %7.2f us -> fused %7.2f us" % (B, N, D, C, a, b), flush=True)
