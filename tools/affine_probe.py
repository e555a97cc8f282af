"""Steady-state start-to-start time of the row-streaming fp32 kernels at the benchmark shape (one uninterrupted
stream of 5 x 200 launches per kernel on 4 rotating buffer sets, first block discarded)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from categoricalnf_amd import ops
dev = torch.device("cuda:0")
B, N, D, R = 16384, 64, 6, 4
g = torch.Generator(device=dev).manual_seed(0)
zs = [torch.randn(B, N, D, generator=g, device=dev) for _ in range(R)]
nns = [0.5 * torch.randn(B, N, 2 * D, generator=g, device=dev) for _ in range(R)]
sf = torch.zeros(D, device=dev); mask = torch.tensor([[1., 1., 1., 0., 0., 0.]], device=dev)
zo = [torch.empty_like(zs[0]) for _ in range(R)]; lo = [torch.empty(B, device=dev) for _ in range(R)]
ln = torch.full((B,), float(N), device=dev)
neglog, nll = torch.empty(B, device=dev), torch.empty(B, device=dev)
sums = torch.zeros(2, dtype=torch.float64, device=dev)
kern = {
    "affine fwd": ([ops.affine_coupling_launch(zs[r], nns[r], sf, mask, zo[r], lo[r]) for r in range(R)], 16),
    "affine fwd + NLL epilogue": ([ops.affine_coupling_nll_launch(zs[r], nns[r], sf, mask, zo[r], lo[r], ln, neglog, nll, None) for r in range(R)], 16),
    "affine inv": ([ops.affine_coupling_launch(zs[r], nns[r], sf, mask, zo[r], lo[r], reverse=True) for r in range(R)], 16),
    "prior_nll (no sum)": ([ops.prior_nll_launch(zs[r], lo[r], ln, neglog, nll, None) for r in range(R)], 4),
    "nll_sum": ([ops.nll_sum_launch(nll, sums)] * R, 0),
}
def steady(launches, reps=200, blocks=5):
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(blocks + 1)]
    marks[0].record()
    for k in range(blocks):
        for i in range(reps):
            launches[i % R]()
        marks[k + 1].record()
    torch.cuda.synchronize()
    return float(np.median([marks[k].elapsed_time(marks[k + 1]) / reps * 1e3 for k in range(1, blocks)]))
for rnd in range(2):
    for name, (l, bpe) in kern.items():
        t = steady(l)
        print("%-28s %7.2f us  %6.0f GB/s algorithmic" % (name, t, bpe * B * N * D / (t * 1e-6) / 1e9))
