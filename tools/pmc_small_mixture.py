"""Workload for the rocprofv3 counter passes on the small-batch / large-K mixture shapes:
PTB language model (B=128,N=288,D=3,K=51, no mask), Zinc nodes (B=512,N=38,D=6,K=16), Zinc edges (B=512,N=703,D=2,K=8),
graph colouring large (B=128,N=50,D=6,K=16), plus configs[1] for comparison.  REP launches each, forward then inverse,
on rotating buffer sets."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from categoricalnf_amd import ops
from categoricalnf_amd.layers.flows.coupling_layer import CouplingLayer

dev = torch.device("cuda:0")
REP, R = 10, 3
SHAPES = [("ptb", 128, 288, 3, 51, False), ("zinc_nodes", 512, 38, 6, 16, True), ("zinc_edges", 512, 703, 2, 8, True),
          ("colour_large", 128, 50, 6, 16, True), ("configs1", 16384, 16, 4, 8, True)]
only = sys.argv[1:]
g = torch.Generator(device=dev).manual_seed(0)
for name, B, N, D, K, masked in SHAPES:
    if only and name not in only:
        continue
    zs = [torch.randn(B, N, D, generator=g, device=dev) for _ in range(R)]
    nns = [0.5 * torch.randn(B, N, D * (2 + 3 * K), generator=g, device=dev) for _ in range(R)]
    mask = CouplingLayer.create_channel_mask(D).to(dev) if masked else None
    zo, zr = torch.empty_like(zs[0]), torch.empty_like(zs[0])
    lf, lr = torch.empty(B, device=dev), torch.empty(B, device=dev)
    fwd = [ops.mixture_coupling_launch(zs[r], nns[r], mask, K, zo, lf) for r in range(R)]
    inv = [ops.mixture_coupling_launch(zo, nns[r], mask, K, zr, lr, reverse=True) for r in range(R)]
    for i in range(REP):
        fwd[i % R]()
    torch.cuda.synchronize()
    for i in range(REP):
        inv[i % R]()
    torch.cuda.synchronize()
    print(name, "done", flush=True)
