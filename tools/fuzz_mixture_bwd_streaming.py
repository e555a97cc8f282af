"""Random shapes, masks (channel either way round, chess), layouts, padding: the fp32 mixture backward with the streaming write-back
forced on (cnf_set_mixture_bwd_big_mb(0)) against the ordinary write-back of the same kernel (one lane per item) — every output bit for
bit.  GPU only:  python tools/fuzz_mixture_bwd_streaming.py"""
import os, sys, random, torch
sys.path.insert(0, os.getcwd())
from categoricalnf_amd import _lib, ops
from categoricalnf_amd.layers.flows.coupling_layer import CouplingLayer
P_ = ops._ptr
lib = _lib.load(); dev = torch.device("cuda:0")
random.seed(11)
bad = 0; ran = 0; declined = 0
for it in range(160):
    D = random.choice([2, 3, 4, 5, 6, 7, 8, 10, 12])
    K = random.choice([1, 2, 3, 4, 5, 8, 10, 16, 20])
    N = random.randint(1, 80); B = random.randint(1, 300)
    compact = random.random() < 0.5
    use_pad = random.random() < 0.5; use_sf = random.random() < 0.7
    DA, P = D - D // 2, 2 + 3 * K
    z = torch.randn(B, N, D, device=dev); gz = torch.randn_like(z); gl = torch.randn(B, device=dev)
    chess = random.random() < 0.4
    if chess:
        compact = False
        mask = torch.tensor([[1.0], [0.0]], device=dev) if random.random() < 0.5 else torch.tensor([[0.0], [1.0]], device=dev)
    else:
        mask = CouplingLayer.create_channel_mask(D).to(dev)
        if random.random() < 0.5: mask = 1 - mask
    m, mr, mc = ops._mask_desc(mask, D, dev); act, n_act = ops._act_list(mask, m, mr, mc, D)
    DA = n_act if not chess else D
    width = (DA if compact else D) * P
    if compact and (B * N * width) % 4: continue
    sf, msf = (0.2 * torch.randn(D, device=dev), 0.2 * torch.randn(D, K, device=dev)) if use_sf else (None, None)
    ln = torch.randint(1, N + 1, (B,)); pad = None
    if use_pad:
        pad = (torch.arange(N)[None, :] < ln[:, None]).float().to(dev).contiguous()
    nn = 0.5 * torch.randn(B, N, width, device=dev)
    ws = torch.empty(int(lib.cnf_bwd_workspace_floats(D + D * K)), device=dev)
    fn = lib.cnf_mixture_coupling_compact_bwd_f32 if compact else lib.cnf_mixture_coupling_bwd_f32
    def run():
        g_z = torch.full_like(z, float("nan")); g_nn = torch.full_like(nn, float("nan"))
        g_sf = torch.zeros(D, device=dev); g_msf = torch.zeros(D, K, device=dev)
        rc = fn(P_(z), P_(nn), P_(sf) if use_sf else None, P_(msf) if use_sf else None, P_(m), mr, mc, act, n_act, P_(pad) if use_pad else None, 1, 1, P_(gz), P_(gl),
                P_(g_z), P_(g_nn), P_(g_sf) if use_sf else None, P_(g_msf) if use_sf else None, P_(ws), B, N, D, K, -1.0, 1.0, 1, ops._stream(dev))
        torch.cuda.synchronize()
        return rc, (g_z, g_nn, g_sf, g_msf)
    lib.cnf_set_mixture_bwd_waves(2)
    rc0, plain = run()
    lib.cnf_set_mixture_bwd_big_mb(0)
    rc1, st = run()
    lib.cnf_set_mixture_bwd_big_mb(-1); lib.cnf_set_mixture_bwd_waves(-1)
    if rc0 or rc1:
        declined += 1; assert rc0 == rc1; continue
    ran += 1
    ok = all(torch.equal(a, b) for a, b in zip(plain, st)) and not torch.isnan(st[1]).any()
    if not ok:
        bad += 1; print("MISMATCH", B, N, D, K, compact, use_pad, use_sf)
print("ran", ran, "declined", declined, "bad", bad)
