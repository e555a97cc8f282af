"""Probe: back-to-back launch timing vs. number of launches per event pair (host-bound or GPU-bound?)."""
import os, sys, time
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
fwd = [ops.affine_coupling_launch(zs[r], nns[r], sf, mask, zo[r], lo[r]) for r in range(R)]
for _ in range(200):
    fwd[0]()
torch.cuda.synchronize()
for reps in (10, 20, 50, 100, 200, 400, 1000):
    out, host = [], []
    for _ in range(7):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); t0 = time.perf_counter()
        for i in range(reps):
            fwd[i % R]()
        t1 = time.perf_counter(); b.record(); torch.cuda.synchronize()
        out.append(a.elapsed_time(b) / reps * 1e3); host.append((t1 - t0) / reps * 1e6)
    print("reps %5d  gpu us/launch med %.2f min %.2f | host us/launch %.2f" % (reps, np.median(out), min(out), np.median(host)))
# same inside a hipGraph
gr = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for i in range(R): fwd[i]()
    torch.cuda.synchronize()
    ops.CAPTURING = True
    with torch.cuda.graph(gr, stream=s):
        for i in range(100):
            fwd[i % R]()
    ops.CAPTURING = False
for _ in range(3):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); gr.replay(); b.record(); torch.cuda.synchronize()
    print("graph of 100 launches: %.2f us/launch" % (a.elapsed_time(b) / 100 * 1e3))
