"""Workload for counter runs: fp32 mixture forward and backward on the compact parameter layout at S* (and the reference layout's
backward), a few launches each:  bash tools/pmc_passes.sh <out> python tools/pmc_mixture_compact_workload.py
(SQ / FETCH_SIZE / WRITE_SIZE passes.  The L1 -> L2 / L2 -> fabric request counters — TCP_TCC_*_sum, TCC_EA0_*_sum — hung rocprofv3 until its
timeout on this pool in round 6, 25 GPU-minutes: do not add them to a pass.)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from categoricalnf_amd import _lib, ops
from categoricalnf_amd.layers.flows.coupling_layer import CouplingLayer
dev = torch.device("cuda:0")
lib = _lib.load()
P_ = ops._ptr
B, N, D, K = 16384, 64, 6, 8
g = torch.Generator(device=dev).manual_seed(1)
DA, P = D - D // 2, 2 + 3 * K
z, gz, gl = torch.randn(B, N, D, generator=g, device=dev), torch.randn(B, N, D, generator=g, device=dev), torch.randn(B, generator=g, device=dev)
mask = CouplingLayer.create_channel_mask(D).to(dev)
m, mr, mc = ops._mask_desc(mask, D, dev)
act, n_act = ops._act_list(mask, m, mr, mc, D)
sf, msf = torch.zeros(D, device=dev), torch.zeros(D, K, device=dev)
g_sf, g_msf, g_z = torch.empty_like(sf), torch.empty_like(msf), torch.empty_like(z)
ws = torch.empty(int(lib.cnf_bwd_workspace_floats(D + D * K)), device=dev)
zf, lf = torch.empty_like(z), torch.empty(B, device=dev)
lib.cnf_set_mixture_bwd_waves(int(os.environ.get("BWD_MODE", "-1")))
for layout, width, entry in (("compact", DA * P, "cnf_mixture_coupling_compact_bwd_f32"), ("ref", D * P, "cnf_mixture_coupling_bwd_f32")):
    nn = 0.5 * torch.randn(B, N, width, generator=g, device=dev)
    g_nn = torch.empty_like(nn)
    fwd = ops.mixture_coupling_launch(z, nn, mask, K, zf, lf)
    for _ in range(3):
        fwd()
        rc = getattr(lib, entry)(P_(z), P_(nn), P_(sf), P_(msf), P_(m), mr, mc, act, n_act, None, 0, 0, P_(gz), P_(gl), P_(g_z), P_(g_nn),
                                 P_(g_sf), P_(g_msf), P_(ws), B, N, D, K, -1.0, 1.0, 1, ops._stream(dev))
        assert rc == 0
    torch.cuda.synchronize()
    del nn, g_nn
