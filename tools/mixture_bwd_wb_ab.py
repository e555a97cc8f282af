"""Mixture backward write-back A/B: 8-byte stores (cnf_set_mixture_bwd_prefetch(2)) against the 16-byte-grid write-back (3) where a
token's span starts or ends on an odd multiple of 8 bytes (D = 6, D = 2 masks ...): same bits, time per call.  GPU only."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from categoricalnf_amd import _lib, ops
from categoricalnf_amd.functional import _ws
from categoricalnf_amd.layers.flows.coupling_layer import CouplingLayer
dev = torch.device("cuda:0"); lib = _lib.load(); P_ = ops._ptr
shapes = [("S*", 16384, 64, 6, 8, "channel"), ("S* inverted mask", 16384, 64, 6, 8, "inv"), ("configs[1] (16-byte spans)", 16384, 16, 4, 8, "channel"),
          ("graph colouring large", 128, 50, 6, 16, "channel"), ("zinc nodes", 512, 38, 6, 16, "channel"), ("zinc edges", 512, 703, 2, 8, "channel"),
          ("D=6 K=4", 4096, 64, 6, 4, "channel"), ("D=2 K=5 ragged", 37, 101, 2, 5, "inv")]
for name, B, N, D, K, kind in shapes:
    g = torch.Generator(device=dev).manual_seed(1)
    z = torch.randn(B, N, D, generator=g, device=dev)
    nn_out = 0.5 * torch.randn(B, N, D * (2 + 3 * K), generator=g, device=dev)
    sf, msf = 0.1 * torch.randn(D, generator=g, device=dev), 0.1 * torch.randn(D, K, generator=g, device=dev)
    mask = CouplingLayer.create_channel_mask(D).to(dev)
    if kind == "inv":
        mask = 1.0 - mask
    m, mr, mc = ops._mask_desc(mask, D, dev)
    act, n_act = ops._act_list(mask, m, mr, mc, D)
    gz, gl = torch.randn(B, N, D, generator=g, device=dev), torch.randn(B, generator=g, device=dev)
    ws = _ws(D + D * K, dev)
    res = {}
    for mode in (2, 3):
        lib.cnf_set_mixture_bwd_prefetch(mode)
        g_z, g_nn = torch.full_like(z, float("nan")), torch.full_like(nn_out, float("nan"))
        g_sf, g_msf = torch.empty_like(sf), torch.empty_like(msf)
        def call():
            rc = lib.cnf_mixture_coupling_bwd_f32(P_(z), P_(nn_out), P_(sf), P_(msf), P_(m), mr, mc, act, n_act, None, 0, 0, P_(gz), P_(gl),
                                                  P_(g_z), P_(g_nn), P_(g_sf), P_(g_msf), P_(ws), B, N, D, K, -1.0, 1.0, 1, ops._stream(dev))
            assert rc == 0
        call(); torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(10): call()
            b.record(); torch.cuda.synchronize()
            best = min(best, a.elapsed_time(b) * 100)
        res[mode] = (g_z.clone(), g_nn.clone(), g_sf.clone(), g_msf.clone(), best)
    lib.cnf_set_mixture_bwd_prefetch(3)
    same = all(torch.equal(res[2][i], res[3][i]) for i in range(4))
    print("%-28s B=%5d N=%3d D=%d K=%2d   8-byte stores %7.1f us   16-byte grid %7.1f us   bit-identical %s   nan-free %s" %
          (name, B, N, D, K, res[2][4], res[3][4], same, bool(torch.isfinite(res[3][1]).all())))
