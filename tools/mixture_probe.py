"""Steady-state time of the mixture-CDF coupling kernels on rotating buffers (configs[1] and S*), for A/B builds:
CNF_LIB_OVERRIDE=<alternative libcnf_hip.so> python tools/mixture_probe.py"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from categoricalnf_amd import _lib
if os.environ.get("CNF_LIB_OVERRIDE"):
    _lib.LIB_PATH = os.environ["CNF_LIB_OVERRIDE"]
from categoricalnf_amd import ops
dev = torch.device("cuda:0")
R = 4
def run(name, B, N, D, K, reps):
    g = torch.Generator(device=dev).manual_seed(1)
    zs = [torch.randn(B, N, D, generator=g, device=dev) for _ in range(R)]
    nns = [0.5 * torch.randn(B, N, D * (2 + 3 * K), generator=g, device=dev) for _ in range(R)]
    mask = torch.cat([torch.ones(1, D // 2), torch.zeros(1, D - D // 2)], 1).to(dev)
    zf = [torch.empty_like(zs[0]) for _ in range(R)]; zr = torch.empty_like(zs[0])
    lf, lr = torch.empty(B, device=dev), torch.empty(B, device=dev)
    fwd = [ops.mixture_coupling_launch(zs[r], nns[r], mask, K, zf[r], lf) for r in range(R)]
    inv = [ops.mixture_coupling_launch(zf[r], nns[r], mask, K, zr, lr, reverse=True) for r in range(R)]
    def steady(l, blocks=4):
        m = [torch.cuda.Event(enable_timing=True) for _ in range(blocks + 1)]
        m[0].record()
        for b in range(blocks):
            for i in range(reps):
                l[i % R]()
            m[b + 1].record()
        torch.cuda.synchronize()
        return float(np.median([m[b].elapsed_time(m[b + 1]) / reps * 1e3 for b in range(1, blocks)]))
    tf, ti = steady(fwd), steady(inv)
    alg = B * N * D * (16 + 12 * K)
    print("%-10s fwd %7.1f us (%5.0f GB/s alg)   inv %7.1f us" % (name, tf, alg / tf / 1e3, ti), flush=True)
run("configs[1]", 16384, 16, 4, 8, 50)
run("S* K=8", 16384, 64, 6, 8, 10)
