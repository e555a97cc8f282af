"""Histogram of the fp64 Newton evaluations per element in the reference-precision inverse (diagnostic build:
bash tools/build_variant.sh iters -DCNF_MIX64_COUNT_ITERS; CNF_LIB_OVERRIDE=categoricalnf_amd/lib/var_iters.so)."""
import os, sys, torch
sys.path.insert(0, os.getcwd())
from categoricalnf_amd import ops, _lib
from categoricalnf_amd.layers.flows.coupling_layer import CouplingLayer
dev = torch.device("cuda:0"); lib = _lib.load()
g = torch.Generator(device=dev).manual_seed(0)
lib.cnf_set_math_mode(0)
for tag, B, N, D, K, zs in (("S*", 16384, 64, 6, 8, 1.0), ("tails x8", 2048, 16, 4, 8, 8.0), ("K=51 PTB", 128, 288, 3, 51, 1.0)):
    z = zs * torch.randn(B, N, D, generator=g, device=dev)
    nn_out = 0.5 * torch.randn(B, N, D * (2 + 3 * K), generator=g, device=dev)
    mask = CouplingLayer.create_channel_mask(D).to(dev)
    zo = torch.empty_like(z); l = torch.empty(B, device=dev)
    ops.mixture_coupling_launch(z, nn_out, mask, K, zo, l, reverse=True)()
    torch.cuda.synchronize()
    m = (mask.view(-1, D)[0] == 0) if mask.dim() > 1 else (mask == 0)
    it = zo[..., m.to(dev)].flatten().round().long()
    h = torch.bincount(it, minlength=8)
    print(tag, "mean %.2f" % it.float().mean().item(), "histogram", {i: int(c) for i, c in enumerate(h.tolist()) if c})
lib.cnf_set_math_mode(1)
