"""fp64 (math mode 0) mixture coupling: token-pass kernel (cnf_set_mixture_kernel 0) against the round-1 fp64 kernel (1).

Per shape: largest differences of z_out / ldj in both directions, and the times of both.  GPU only."""
import os, sys, torch
sys.path.insert(0, os.getcwd())
from categoricalnf_amd import ops, _lib
from categoricalnf_amd.layers.flows.coupling_layer import CouplingLayer
dev = torch.device("cuda:0"); lib = _lib.load()
g = torch.Generator(device=dev).manual_seed(0)

def timed(f, reps=10):
    f(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(4):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps): f()
        b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) * 1000 / reps)
    return best

only = os.environ.get("SHAPES")
shapes = [("configs[1]", 16384, 16, 4, 8, 1.0), ("S*", 16384, 64, 6, 8, 1.0), ("tails x8", 2048, 16, 4, 8, 8.0),
          ("K=4", 4096, 16, 4, 4, 1.0), ("K=16", 4096, 16, 4, 16, 1.0), ("K=5", 4096, 16, 4, 5, 1.0),
          ("K=10", 1024, 64, 4, 10, 1.0), ("K=51 PTB", 128, 288, 3, 51, 1.0), ("D=2 K=8", 1024, 72, 2, 8, 1.0),
          ("nomask D=3", 2048, 16, 3, 8, 1.0)]
if only:
    shapes = [s for s in shapes if s[0] in only.split(",")]
lib.cnf_set_math_mode(0)
print("%-12s %28s | %10s %10s | %10s %10s | %9s %9s %9s %9s" % ("shape", "B,N,D,K", "dz fwd", "dldj fwd", "dz inv", "dldj inv",
                                                                 "fwd old", "fwd new", "inv old", "inv new"))
try:
    for tag, B, N, D, K, zs in shapes:
        z = zs * torch.randn(B, N, D, generator=g, device=dev)
        nn_out = 0.5 * torch.randn(B, N, D * (2 + 3 * K), generator=g, device=dev)
        mask = None if tag.startswith("nomask") else CouplingLayer.create_channel_mask(D).to(dev)
        sf = 0.3 * torch.randn(D, generator=g, device=dev); msf = 0.3 * torch.randn(D, K, generator=g, device=dev)
        res = {}
        for which in (1, 0):
            lib.cnf_set_mixture_kernel(which)
            zo, zr = torch.empty_like(z), torch.empty_like(z); lf, lr = torch.empty(B, device=dev), torch.empty(B, device=dev)
            kw = {} if os.environ.get("NOSF") else dict(scaling_factor=sf, mixture_scaling_factor=msf)
            fwd = ops.mixture_coupling_launch(z, nn_out, mask, K, zo, lf, **kw)
            fwd(); torch.cuda.synchronize()
            inv = ops.mixture_coupling_launch(zo, nn_out, mask, K, zr, lr, reverse=True, **kw)
            inv(); torch.cuda.synchronize()
            res[which] = (zo.clone(), lf.clone(), zr.clone(), lr.clone(), timed(fwd), timed(inv))
        o, n = res[1], res[0]
        d = [(o[i].double() - n[i].double()).abs().max().item() for i in range(4)]
        rt = (n[2] - z).abs().max().item()
        print("%-12s %28s | %10.2e %10.2e | %10.2e %10.2e | %9.1f %9.1f %9.1f %9.1f   round trip %.2e" %
              (tag, (B, N, D, K), d[0], d[1], d[2], d[3], o[4], n[4], o[5], n[5], rt))
finally:
    lib.cnf_set_math_mode(1); lib.cnf_set_mixture_kernel(0)
