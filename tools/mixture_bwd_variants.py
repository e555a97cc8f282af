"""cnf_mixture_coupling_bwd_f32 at shapes the sweep does not hold (K = 4 and K = 16, small and large): the default against the unrolled
register-slot kernel and the rolled kernel with 1 / 2 / 4 lanes per item (cnf_set_mixture_bwd_waves), start-to-start microseconds per call
(a forced variant whose stage does not fit LDS runs the fp64 kernel: the 800+ / 5000 us entries)."""
import ctypes, os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from categoricalnf_amd import _lib, ops, functional as Fn
from categoricalnf_amd.layers.flows.coupling_layer import CouplingLayer
lib = _lib.load(); dev = torch.device("cuda:0")
for (B, N, D, K) in ((16384, 16, 4, 4), (1024, 16, 4, 4), (16384, 64, 6, 4), (512, 38, 6, 4), (16384, 16, 4, 16), (16384, 64, 6, 16)):
    g = torch.Generator(device=dev).manual_seed(0)
    z = torch.randn(B, N, D, generator=g, device=dev); nn_ = 0.5 * torch.randn(B, N, D * (2 + 3 * K), generator=g, device=dev)
    mask = CouplingLayer.create_channel_mask(D).to(dev)
    gz, gl = torch.randn(B, N, D, generator=g, device=dev), torch.randn(B, generator=g, device=dev)
    sf, msf = torch.zeros(D, device=dev), torch.zeros(D, K, device=dev)
    m, mr, mc = ops._mask_desc(mask, D, dev); act, n_act = ops._act_list(mask, m, mr, mc, D)
    g_z, g_nn = torch.empty_like(z), torch.empty_like(nn_); g_sf, g_msf = torch.empty_like(sf), torch.empty_like(msf)
    ws = torch.empty(int(lib.cnf_bwd_workspace_floats(D + D * K)), device=dev)
    P = ops._ptr
    def f():
        assert lib.cnf_mixture_coupling_bwd_f32(P(z), P(nn_), P(sf), P(msf), P(m), mr, mc, act, n_act, None, 0, 0, P(gz), P(gl), P(g_z), P(g_nn), P(g_sf), P(g_msf), P(ws), B, N, D, K, -1.0, 1.0, 1, ops._stream(dev)) == 0
    out = []
    for mode in (-1, 0, 2, 3, 4):
        lib.cnf_set_mixture_bwd_waves(mode)
        for _ in range(5): f()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(50): f()
        b.record(); torch.cuda.synchronize()
        out.append("%s %.1f" % ({-1: "default", 0: "unrolled", 2: "rolled G=1", 3: "G=2", 4: "G=4"}[mode], a.elapsed_time(b) / 50 * 1e3))
    lib.cnf_set_mixture_bwd_waves(-1)
    print("B=%d N=%d D=%d K=%d: %s us" % (B, N, D, K, " | ".join(out)), flush=True)
