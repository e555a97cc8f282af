"""Workload for rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over the fp32 mixture backward at S* (B=16384, N=64, D=6, K=8):
a d2d copy of KNOWN size for calibration (MI355X_MICROARCH.md, HBM section), then REP calls on the compact layout, then REP on the
reference layout (tools/pmc_mixture_bwd_summarize.py tells the two groups apart by dispatch order).  Counters in their own runs:
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_bwd_fetch -o pmc -- python tools/pmc_mixture_bwd_traffic_workload.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from categoricalnf_amd import _lib, ops
from categoricalnf_amd.layers.flows.coupling_layer import CouplingLayer
dev = torch.device("cuda:0")
lib = _lib.load()
P_ = ops._ptr
B, N, D, K, R, REP = 16384, 64, 6, 8, 2, 8
g = torch.Generator(device=dev).manual_seed(0)
srcs = [torch.randn(B * N * D * 4, generator=g, device=dev) for _ in range(R)]
dsts = [torch.empty_like(srcs[0]) for _ in range(R)]
for i in range(REP):
    dsts[i % R].copy_(srcs[i % R])
torch.cuda.synchronize()
del srcs, dsts
DA, P = D - D // 2, 2 + 3 * K
zs = [torch.randn(B, N, D, generator=g, device=dev) for _ in range(R)]
gzs = [torch.randn(B, N, D, generator=g, device=dev) for _ in range(R)]
gl = torch.randn(B, generator=g, device=dev)
mask = CouplingLayer.create_channel_mask(D).to(dev)
m, mr, mc = ops._mask_desc(mask, D, dev)
act, n_act = ops._act_list(mask, m, mr, mc, D)
sf, msf = torch.zeros(D, device=dev), torch.zeros(D, K, device=dev)
g_sf, g_msf, g_z = torch.empty_like(sf), torch.empty_like(msf), torch.empty_like(zs[0])
ws = torch.empty(int(lib.cnf_bwd_workspace_floats(D + D * K)), device=dev)
for width, fn in ((DA * P, lib.cnf_mixture_coupling_compact_bwd_f32), (D * P, lib.cnf_mixture_coupling_bwd_f32)):
    nns = [0.5 * torch.randn(B, N, width, generator=g, device=dev) for _ in range(R)]
    g_nn = torch.empty_like(nns[0])
    for i in range(REP):
        r = i % R
        rc = fn(P_(zs[r]), P_(nns[r]), P_(sf), P_(msf), P_(m), mr, mc, act, n_act, None, 0, 0, P_(gzs[r]), P_(gl), P_(g_z), P_(g_nn),
                P_(g_sf), P_(g_msf), P_(ws), B, N, D, K, -1.0, 1.0, 1, ops._stream(dev))
        assert rc == 0, lib.cnf_last_error()
    torch.cuda.synchronize()
    del nns, g_nn
print("done")
