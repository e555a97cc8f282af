"""A/B of library builds on the mixture kernels: CNF_LIB_OVERRIDE=<.so> python tools/ab_mixture.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from categoricalnf_amd import _lib, ops
from categoricalnf_amd.layers.flows.coupling_layer import CouplingLayer
dev = torch.device("cuda:0")
lib = _lib.load()
out = []
for name, B, N, D, K, masked in [("cfg1", 16384, 16, 4, 8, True), ("S*", 16384, 64, 6, 8, True), ("ptb", 128, 288, 3, 51, False),
                                 ("zedge", 512, 703, 2, 8, True), ("znode", 512, 38, 6, 16, True)]:
    g = torch.Generator(device=dev).manual_seed(1)
    R = 3
    zs = [torch.randn(B, N, D, generator=g, device=dev) for _ in range(R)]
    nns = [0.5 * torch.randn(B, N, D * (2 + 3 * K), generator=g, device=dev) for _ in range(R)]
    mask = CouplingLayer.create_channel_mask(D).to(dev) if masked else None
    zf, zr = torch.empty_like(zs[0]), torch.empty_like(zs[0])
    lf, lr = torch.empty(B, device=dev), torch.empty(B, device=dev)
    fwd = [ops.mixture_coupling_launch(zs[r], nns[r], mask, K, zf, lf) for r in range(R)]
    inv = [ops.mixture_coupling_launch(zf, nns[r], mask, K, zr, lr, reverse=True) for r in range(R)]

    def timeit(ls, reps=12):
        for l in ls:
            l()
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for i in range(reps):
                ls[i % R]()
            b.record(); torch.cuda.synchronize()
            ts.append(a.elapsed_time(b) / reps * 1e3)
        return min(ts)
    out.append("%s %.1f/%.1f" % (name, timeit(fwd), timeit(inv)))
print(os.environ.get("CNF_LIB_OVERRIDE", "default"), "| fwd/inv us:", "  ".join(out))
