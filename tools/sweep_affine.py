"""Within-process interleaved A/B of the affine coupling kernel's tiling knobs (GPU only).
Prints median / min kernel time per (tile_chunks, unroll) over interleaved rounds."""
import itertools
import sys
import os
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from categoricalnf_amd import _lib, ops  # noqa: E402

B, N, D = 16384, 64, 6
dev = torch.device("cuda:0")
lib = _lib.load()
g = torch.Generator(device=dev).manual_seed(0)
R = 4
zs = [torch.randn(B, N, D, generator=g, device=dev) for _ in range(R)]
nns = [0.5 * torch.randn(B, N, 2 * D, generator=g, device=dev) for _ in range(R)]
sf = torch.zeros(D, device=dev)
mask = torch.tensor([[1., 1., 1., 0., 0., 0.]], device=dev)
zo = [torch.empty_like(zs[0]) for _ in range(R)]
lo = [torch.empty(B, device=dev) for _ in range(R)]
fwd = [ops.affine_coupling_launch(zs[r], nns[r], sf, mask, zo[r], lo[r]) for r in range(R)]
inv = [ops.affine_coupling_launch(zs[r], nns[r], sf, mask, zo[r], lo[r], reverse=True) for r in range(R)]
configs = list(itertools.product([128, 192, 256, 384, 512], [0, 1, 2], [0, 1]))
res = {c: [] for c in configs}
resi = {c: [] for c in configs}
reps = 20
for rnd in range(6):
    for c in configs:
        lib.cnf_set_tile_chunks(c[0]); lib.cnf_set_unroll(c[1]); lib.cnf_set_math_mode(c[2])
        for which, store in ((fwd, res), (inv, resi)):
            for r in range(R):
                which[r]()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for i in range(reps):
                which[i % R]()
            b.record()
            torch.cuda.synchronize()
            store[c].append(a.elapsed_time(b) / reps * 1e3)
alg = 16.0 * B * N * D + 4 * B
print("tile_chunks unroll(0=prefetch) fastmath | fwd med us  min us  GB/s(min) | inv med us min us")
for c in configs:
    f, i = np.array(res[c][1:]), np.array(resi[c][1:])
    print("%5d %3d %3d | %8.2f %8.2f %8.0f | %8.2f %8.2f" % (c[0], c[1], c[2], np.median(f), f.min(), alg / (f.min() * 1e-6) / 1e9, np.median(i), i.min()))

# stream-copy ceiling measured in the same run, same byte count per launch (75.5 MB read would need a
# 3-input kernel; a d2d copy of 50.4 MB moves 100.7 MB), rotating buffers so the 256 MB Infinity Cache cannot serve it
n = int(alg // 8)
srcs = [torch.empty(n, dtype=torch.float32, device=dev).normal_() for _ in range(R)]
dsts = [torch.empty(n, dtype=torch.float32, device=dev) for _ in range(R)]
ts = []
for rnd in range(6):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(reps):
        dsts[i % R].copy_(srcs[i % R])
    b.record()
    torch.cuda.synchronize()
    ts.append(a.elapsed_time(b) / reps * 1e3)
print("torch d2d copy of %.1f MB x %d rotating buffers (read+write = %.1f MB per copy): min %.2f us -> %.0f GB/s"
      % (n * 4 / 1e6, R, n * 8 / 1e6, min(ts), n * 8 / (min(ts) * 1e-6) / 1e9))
