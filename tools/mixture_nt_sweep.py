"""fp32 mixture forward / inverse with and without the nontemporal hint on the DMA loads, over launch sizes: where the staged bytes
pass the memory-side cache the hint starts to pay (cnf_set_mixture_nt_mb).  Rotating buffer sets as in bench.py.  GPU only."""
import os, sys, torch
sys.path.insert(0, os.getcwd())
from categoricalnf_amd import ops, _lib
from categoricalnf_amd.layers.flows.coupling_layer import CouplingLayer
dev = torch.device("cuda:0"); lib = _lib.load()
g = torch.Generator(device=dev).manual_seed(0)

def timed(fs, reps=30):
    for f in fs: f()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for i in range(reps): fs[i % len(fs)]()
        b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) * 1000 / reps)
    return best

print("%-34s %10s | %9s %9s | %9s %9s" % ("shape", "staged MB", "fwd", "fwd nt", "inv", "inv nt"))
for B, N, D, K in ((16384, 16, 4, 8), (16384, 16, 6, 8), (16384, 32, 6, 8), (16384, 48, 6, 8), (16384, 64, 6, 8), (16384, 64, 4, 8), (32768, 64, 6, 8),
                   (8192, 64, 6, 16), (16384, 64, 6, 4), (4096, 288, 3, 51)):
    R = 2 if B * N * D * (2 + 3 * K) * 4 > 5e8 else 3
    zs = [torch.randn(B, N, D, generator=g, device=dev) for _ in range(R)]
    nns = [0.5 * torch.randn(B, N, D * (2 + 3 * K), generator=g, device=dev) for _ in range(R)]
    mask = None if D == 3 else CouplingLayer.create_channel_mask(D).to(dev)
    zo = [torch.empty_like(zs[0]) for _ in range(R)]; lf = torch.empty(B, device=dev)
    fw = [ops.mixture_coupling_launch(zs[r], nns[r], mask, K, zo[r], lf) for r in range(R)]
    for f in fw: f()
    iv = [ops.mixture_coupling_launch(zo[r], nns[r], mask, K, zs[r], lf, reverse=True) for r in range(R)]
    DA = D if mask is None else D - D // 2
    staged = B * N * DA * (2 + 3 * K) * 4 / 2**20
    row = []
    for fs in (fw, iv):
        for mb in (0, 1):
            lib.cnf_set_mixture_nt_mb(mb)
            row.append(timed(fs))
        for f in fw: f()
    lib.cnf_set_mixture_nt_mb(-1)
    print("%-34s %10.0f | %9.1f %9.1f | %9.1f %9.1f" % ("B=%d N=%d D=%d K=%d" % (B, N, D, K), staged, *row))
    del zs, nns, zo, fw, iv
