"""Tiling / unroll sweep of the bench's dominant kernel (affine forward + NLL + batch-sum epilogue), interleaved rounds."""
import os, sys, itertools
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from categoricalnf_amd import _lib, ops
dev = torch.device("cuda:0")
lib = _lib.load()
B, N, D, R = 16384, 64, 6, 4
g = torch.Generator(device=dev).manual_seed(0)
zs = [torch.randn(B, N, D, generator=g, device=dev) for _ in range(R)]
nns = [0.5 * torch.randn(B, N, 2 * D, generator=g, device=dev) for _ in range(R)]
sf = torch.zeros(D, device=dev); mask = torch.tensor([[1., 1., 1., 0., 0., 0.]], device=dev)
zo = [torch.empty_like(zs[0]) for _ in range(R)]; lo = [torch.empty(B, device=dev) for _ in range(R)]
ln = torch.full((B,), float(N), device=dev)
neglog, nll = torch.empty(B, device=dev), torch.empty(B, device=dev)
acc = torch.zeros(ops.NLL_ACC_SLOTS, dtype=torch.int64, device=dev)
k = [ops.affine_coupling_nll_acc_launch(zs[r], nns[r], sf, mask, zo[r], lo[r], ln, neglog, nll, acc) for r in range(R)]
cfgs = list(itertools.product([64, 96, 128, 192, 256, 384, 512], [1, 2]))
res = {c: [] for c in cfgs}
for _ in range(300):
    k[0]()
for rnd in range(5):
    for c in cfgs:
        lib.cnf_set_tile_chunks(c[0]); lib.cnf_set_unroll(c[1])
        for r in range(R):
            k[r]()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for i in range(100):
            k[i % R]()
        b.record(); torch.cuda.synchronize()
        res[c].append(a.elapsed_time(b) / 100 * 1e3)
print("tile_chunks unroll | med us  min us")
for c in cfgs:
    v = np.array(res[c][1:])
    print("%5d %3d | %7.2f %7.2f" % (c[0], c[1], np.median(v), v.min()))
