"""Evaluations of the fp32 Newton loop of the mixture inverse: per element, and per wave (a wave runs until its slowest lane is done).
Diagnostic build: bash tools/build_variant.sh iters32 -DCNF_MIX32_COUNT_ITERS; CNF_LIB_OVERRIDE=categoricalnf_amd/lib/var_iters32.so"""
import os, sys, torch
sys.path.insert(0, os.getcwd())
from categoricalnf_amd import ops, _lib
from categoricalnf_amd.layers.flows.coupling_layer import CouplingLayer
dev = torch.device("cuda:0"); lib = _lib.load()
g = torch.Generator(device=dev).manual_seed(0)
for tag, B, N, D, K, zs in (("S*", 16384, 64, 6, 8, 1.0), ("configs[1]", 16384, 16, 4, 8, 1.0), ("tails x8", 2048, 16, 4, 8, 8.0)):
    z = zs * torch.randn(B, N, D, generator=g, device=dev)
    nn_out = 0.5 * torch.randn(B, N, D * (2 + 3 * K), generator=g, device=dev)
    mask = CouplingLayer.create_channel_mask(D).to(dev)
    zo = torch.empty_like(z); l = torch.empty(B, device=dev)
    ops.mixture_coupling_launch(z, nn_out, mask, K, zo, l, reverse=True)()
    torch.cuda.synchronize()
    m = (mask.view(-1, D)[0] == 0)
    it = zo[..., m.to(dev)].round().long()                 # [B, N, DA]
    DA = it.shape[-1]
    tpp = 64 // DA
    h = torch.bincount(it.flatten(), minlength=8)
    # passes of tpp tokens of a row: the lanes of one wave-pass
    npass = N // tpp
    per_pass = it[:, : npass * tpp].reshape(B, npass, tpp * DA).max(-1).values.float() if npass else it.reshape(B, -1).max(-1).values.float()
    print(tag, "mean per element %.2f" % it.float().mean().item(), "| mean of the per-pass maximum %.2f" % per_pass.mean().item(),
          "| histogram", {i: int(c) for i, c in enumerate(h.tolist()) if c})
