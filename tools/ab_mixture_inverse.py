"""A/B of the mixture-CDF Newton inverse at configs[1] / S* / PTB: wave-tile size (passes per wave) and the knobs a
build exposes.  Interleaved blocks on rotating buffers (4 sets, > the 256 MB Infinity Cache at the large shapes).  GPU only."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from categoricalnf_amd import _lib, ops
dev = torch.device("cuda:0")
lib = _lib.load()
R = 4
shapes = [("configs[1]", 16384, 16, 4, 8, "channel"), ("S*", 16384, 64, 6, 8, "channel"), ("PTB", 128, 288, 3, 51, "none"),
          ("zinc edges", 512, 703, 2, 8, "channel"), ("zinc nodes", 512, 38, 6, 16, "channel")]
for name, B, N, D, K, kind in shapes:
    g = torch.Generator(device=dev).manual_seed(1)
    zs = [torch.randn(B, N, D, generator=g, device=dev) for _ in range(R)]
    nns = [0.5 * torch.randn(B, N, D * (2 + 3 * K), generator=g, device=dev) for _ in range(R)]
    mask = None if kind == "none" else torch.cat([torch.ones(1, D // 2), torch.zeros(1, D - D // 2)], 1).to(dev)
    zfs = [torch.empty_like(zs[0]) for _ in range(R)]
    zrs = [torch.empty_like(zs[0]) for _ in range(R)]
    lf, lr = torch.empty(B, device=dev), torch.empty(B, device=dev)

    def build():
        fwd = [ops.mixture_coupling_launch(zs[r], nns[r], mask, K, zfs[r], lf) for r in range(R)]
        inv = [ops.mixture_coupling_launch(zfs[r], nns[r], mask, K, zrs[r], lr, reverse=True) for r in range(R)]
        return fwd, inv

    def steady(launches, reps=40, blocks=4):
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(blocks + 1)]
        marks[0].record()
        for b in range(blocks):
            for i in range(reps):
                launches[i % R]()
            marks[b + 1].record()
        torch.cuda.synchronize(dev)
        return float(np.median([marks[b].elapsed_time(marks[b + 1]) / reps for b in range(1, blocks)])) * 1e3
    out = []
    for tile in (64, 128, 256):
        lib.cnf_set_mixture_tile(tile)
        fwd, inv = build()
        for f in fwd:
            f()
        res = []
        for _ in range(3):
            res.append((steady(fwd), steady(inv)))
        out.append("tile %3d: fwd %6.1f inv %6.1f" % (tile, min(r[0] for r in res), min(r[1] for r in res)))
    lib.cnf_set_mixture_tile(128)
    err = (zrs[0] - zs[0]).abs().max().item()
    print("%-12s B=%5d N=%3d D=%d K=%2d | %s | round-trip err %.1e" % (name, B, N, D, K, " | ".join(out), err), flush=True)
