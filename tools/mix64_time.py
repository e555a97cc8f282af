import os, sys, torch
sys.path.insert(0, os.getcwd())
from categoricalnf_amd import ops, _lib
from categoricalnf_amd.layers.flows.coupling_layer import CouplingLayer
dev = torch.device("cuda:0"); lib = _lib.load()
g = torch.Generator(device=dev).manual_seed(0)
lib.cnf_set_math_mode(0)
for tag, (B, N, D, K) in (("configs[1]", (16384, 16, 4, 8)), ("S*", (16384, 64, 6, 8))):
    z = torch.randn(B, N, D, generator=g, device=dev); nn_out = 0.5 * torch.randn(B, N, D * (2 + 3 * K), generator=g, device=dev)
    mask = CouplingLayer.create_channel_mask(D).to(dev)
    zo, zr = torch.empty_like(z), torch.empty_like(z); lf, lr = torch.empty(B, device=dev), torch.empty(B, device=dev)
    fwd = ops.mixture_coupling_launch(z, nn_out, mask, K, zo, lf); inv = ops.mixture_coupling_launch(zo, nn_out, mask, K, zr, lr, reverse=True)
    for name, f in (("fwd", fwd), ("inv", inv)):
        f(); torch.cuda.synchronize()
        best = 1e9
        for _ in range(4):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(10): f()
            b.record(); torch.cuda.synchronize()
            best = min(best, a.elapsed_time(b) * 100)
        print("%-10s %s fp64: %7.1f us" % (tag, name, best))
lib.cnf_set_math_mode(1)
