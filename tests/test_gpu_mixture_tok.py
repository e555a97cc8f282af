"""GPU tests of the fp32 token-pass mixture kernel (cnf_mixture_tok.hip): DMA-staged rows, rows split over several
workgroups through the fixed-point workspace, NLL epilogue.  Checked against the oracle, against the round-1 fp32
kernel (cnf_set_mixture_kernel(1)) and through properties at the real configuration sizes."""
import numpy as np
import pytest
import torch

from categoricalnf_amd import _lib
from oracle import cnf_oracle as O

pytestmark = pytest.mark.gpu
ELEM = dict(rtol=2e-5, atol=2e-5)


def ops():
    from categoricalnf_amd import ops as o
    return o


def g(t):
    return None if t is None else t.cuda()


def close(a, b, **kw):
    torch.testing.assert_close(a.detach().cpu(), b.detach().cpu(), **kw)


def loglik_close(actual, ref, rel=1e-4, floor=1.0):
    """BASELINE north_star bar on per-sample log-likelihood terms: max |actual - ref| <= rel * max(|ref|, floor)."""
    a, r = actual.detach().double().cpu(), ref.detach().double().cpu()
    assert a.shape == r.shape, (a.shape, r.shape)
    if a.numel() == 0:
        return
    worst = ((a - r).abs() / r.abs().clamp(min=floor)).max().item()
    assert worst <= rel, "relative deviation %.3g exceeds %.1g" % (worst, rel)


def rel_ll(a, b, floor=1.0):
    """max |a - b| / max(|b|, floor): the north-star bar is 1e-4 on per-sample log-likelihood."""
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).abs() / b.abs().clamp(min=floor)).max().item()


def _case(B, N, D, K, kind, seed, padded=True):
    gen = torch.Generator().manual_seed(seed)
    z = 1.5 * torch.randn(B, N, D, generator=gen)
    nn_out = 0.6 * torch.randn(B, N, D * (2 + 3 * K), generator=gen)
    sf, msf = 0.2 * torch.randn(D, generator=gen), 0.2 * torch.randn(D, K, generator=gen)
    if kind == "none":
        mask = None
    elif kind == "chess":
        mask = O.chess_mask()
    elif kind == "channel_inv":
        mask = 1.0 - O.channel_mask(D)
    else:
        mask = O.channel_mask(D)
    ln = torch.randint(max(1, N // 2), N + 1, (B,), generator=gen)
    ln[0] = N
    pad = O.length_mask(ln, N) if (padded and kind != "none") else None
    return z, nn_out, sf, msf, mask, ln, pad


# shapes of the reference's configurations (scaled-down batch) and awkward ones: rows shorter / longer than a pass,
# odd K (run-time K path, 1/2/4 lanes per item), D = 1 chess mask, inverted channel mask (first channels transformed)
SHAPES = [(40, 16, 4, 8, "channel"), (12, 288, 3, 51, "none"), (24, 38, 6, 16, "channel"), (6, 703, 2, 8, "channel"),
          (16, 50, 6, 16, "channel_inv"), (11, 13, 1, 8, "chess"), (70, 5, 3, 4, "channel"), (300, 1, 2, 8, "channel"),
          (5, 97, 5, 9, "channel"), (3, 400, 2, 23, "none"), (2, 1500, 3, 5, "channel_inv"), (33, 20, 2, 8, "channel"),
          (1, 2048, 4, 8, "channel"), (130, 64, 6, 8, "channel")]


@pytest.mark.parametrize("B,N,D,K,kind", SHAPES)
def test_tok_kernel_vs_oracle_and_round1_kernel(B, N, D, K, kind):
    z, nn_out, sf, msf, mask, ln, pad = _case(B, N, D, K, kind, B + 7 * N + 31 * D + K)
    kw = dict(num_mixtures=K, reg_max=3.5, reg_factor=2.0, is_training=True)
    gk = dict(scaling_factor=g(sf), mixture_scaling_factor=g(msf), channel_padding_mask=g(pad), **kw)
    zo, lo, ro = O.mixture_coupling(z, nn_out, mask, scaling_factor=sf, mixture_scaling_factor=msf, channel_padding_mask=pad, **kw)
    ldj0 = torch.randn(B, generator=torch.Generator().manual_seed(5))
    zf, lf, rf = ops().mixture_coupling(g(z), g(nn_out), g(mask), ldj=g(ldj0), **gk)
    close(zf, zo, **ELEM); loglik_close(lf, lo + ldj0); loglik_close(rf, ro)
    assert rel_ll(lf, lo + ldj0) < 1e-4
    lib = _lib.load()
    lib.cnf_set_mixture_kernel(1)
    try:
        z1, l1, r1 = ops().mixture_coupling(g(z), g(nn_out), g(mask), ldj=g(ldj0), **gk)
        zr1, lr1, _ = ops().mixture_coupling(g(zo), g(nn_out), g(mask), reverse=True, **gk)
    finally:
        lib.cnf_set_mixture_kernel(0)
    # same arithmetic per element (up to the lanes-per-item split of a run-time K); the per-sample sum order differs
    close(zf, z1, rtol=1e-5, atol=1e-5)
    close(lf, l1, rtol=1e-5, atol=5e-5)
    zr, lr, _ = ops().mixture_coupling(g(zo), g(nn_out), g(mask), reverse=True, **gk)
    close(zr, zr1, rtol=1e-5, atol=1e-5)
    close(lr, lr1, rtol=1e-5, atol=5e-5)
    zo2, lo2, _ = O.mixture_coupling(zo, nn_out, mask, scaling_factor=sf, mixture_scaling_factor=msf,
                                     channel_padding_mask=pad, reverse=True, **kw)
    close(zr, zo2, rtol=1e-4, atol=1e-4); loglik_close(lr, lo2)
    ops().check_flags(torch.device("cuda"), "tok kernel")


@pytest.mark.parametrize("B,N,D,K,kind", [(12, 288, 3, 51, "none"), (24, 38, 6, 16, "channel"), (6, 703, 2, 8, "channel"),
                                          (40, 16, 4, 8, "channel"), (2, 1500, 3, 5, "channel_inv"), (11, 13, 1, 8, "chess")])
@pytest.mark.parametrize("split_waves", [256, 4096, 65536])
def test_split_rows_are_deterministic_and_leave_the_workspace_zeroed(B, N, D, K, kind, split_waves):
    """Rows shared by S workgroups (S depends on the wave target): the fixed-point meeting point makes the per-sample
    log-det bit-identical from launch to launch, and every launch hands the workspace back zeroed."""
    z, nn_out, sf, msf, mask, ln, pad = _case(B, N, D, K, kind, 99 + B + N)
    lib = _lib.load()
    lib.cnf_set_mixture_split(split_waves)
    try:
        outs = [ops().mixture_coupling(g(z), g(nn_out), g(mask), K, scaling_factor=g(sf), mixture_scaling_factor=g(msf),
                                       channel_padding_mask=g(pad)) for _ in range(3)]
    finally:
        lib.cnf_set_mixture_split(4096)
    for zf, lf, _ in outs[1:]:
        assert torch.equal(zf, outs[0][0]) and torch.equal(lf, outs[0][1])
    zo, lo, _ = O.mixture_coupling(z, nn_out, mask, K, sf, msf, channel_padding_mask=pad)
    close(outs[0][0], zo, **ELEM); loglik_close(outs[0][1], lo)
    torch.cuda.synchronize()
    for w in ops()._mix_ws.values():
        assert int(w.count_nonzero().item()) == 0


@pytest.mark.parametrize("lanes", [1, 2, 4])
def test_lanes_per_item_agree(lanes):
    """Run-time K with 1, 2 or 4 lanes per item: same results to rounding of the partial-sum order."""
    B, N, D, K = 9, 60, 3, 23
    z, nn_out, sf, msf, mask, ln, pad = _case(B, N, D, K, "channel", 4242)
    lib = _lib.load()
    lib.cnf_set_mixture_lanes(lanes)
    try:
        zf, lf, _ = ops().mixture_coupling(g(z), g(nn_out), g(mask), K, scaling_factor=g(sf), mixture_scaling_factor=g(msf),
                                           channel_padding_mask=g(pad))
        zo, lo, _ = O.mixture_coupling(z, nn_out, mask, K, sf, msf, channel_padding_mask=pad)
        close(zf, zo, **ELEM); loglik_close(lf, lo)
        zr, lr, _ = ops().mixture_coupling(zf, g(nn_out), g(mask), K, scaling_factor=g(sf), mixture_scaling_factor=g(msf),
                                           channel_padding_mask=g(pad), reverse=True)
        keep = pad.expand(-1, -1, D) > 0
        assert ((zr.cpu() - z)[keep]).abs().max() < 5e-4
    finally:
        lib.cnf_set_mixture_lanes(0)


@pytest.mark.parametrize("K", [1, 2, 3, 5, 6, 7, 9, 10, 12, 13, 14, 15, 17, 20, 26, 27, 28, 29, 32, 33, 40, 49, 51, 52, 53, 64, 65, 70])
def test_every_mixture_count_on_register_slots(K):
    """Round 3: a run-time K other than 4 / 8 / 16 runs on predicated register slots (KT mixtures per lane, G lanes per
    item, the smallest instantiated KT * G >= K; K > 64 keeps the rolled LDS loop).  Every count from below, at and above
    the pair boundaries against the oracle and against the rolled-loop kernel (cnf_set_mixture_lanes(G) selects it),
    forward, NLL epilogue and Newton inverse."""
    for B, N, D, kind in ((7, 37, 3, "none"), (5, 21, 4, "channel"), (3, 130, 2, "channel_inv")):
        z, nn_out, sf, msf, mask, ln, pad = _case(B, N, D, K, kind, 1000 + 13 * K + D)
        kw = dict(num_mixtures=K, reg_max=3.5, reg_factor=2.0, is_training=True)
        gk = dict(scaling_factor=g(sf), mixture_scaling_factor=g(msf), channel_padding_mask=g(pad), **kw)
        zo, lo, ro = O.mixture_coupling(z, nn_out, mask, scaling_factor=sf, mixture_scaling_factor=msf, channel_padding_mask=pad, **kw)
        zf, lf, rf = ops().mixture_coupling(g(z), g(nn_out), g(mask), **gk)
        close(zf, zo, **ELEM); loglik_close(rf, ro)
        assert rel_ll(lf, lo) < 1e-4
        zr, lr, _ = ops().mixture_coupling(g(zo), g(nn_out), g(mask), reverse=True, **gk)
        zo2, lo2, _ = O.mixture_coupling(zo, nn_out, mask, scaling_factor=sf, mixture_scaling_factor=msf,
                                         channel_padding_mask=pad, reverse=True, **kw)
        close(zr, zo2, rtol=1e-4, atol=1e-4)
        assert rel_ll(lr, lo2) < 1e-4
        lib = _lib.load()
        lib.cnf_set_mixture_lanes(4 if K > 32 else (2 if K > 16 else 1))
        try:
            z1, l1, _ = ops().mixture_coupling(g(z), g(nn_out), g(mask), **gk)
            zr1, lr1, _ = ops().mixture_coupling(g(zo), g(nn_out), g(mask), reverse=True, **gk)
        finally:
            lib.cnf_set_mixture_lanes(0)
        close(zf, z1, rtol=1e-5, atol=1e-5); close(lf, l1, rtol=1e-5, atol=5e-5)
        close(zr, zr1, rtol=1e-5, atol=1e-5); close(lr, lr1, rtol=1e-5, atol=5e-5)
        # NLL epilogue on the same slots
        nll = ops().mixture_coupling_nll(g(z), g(nn_out), g(mask), length=g(ln.float()), **gk)
        assert torch.equal(nll[0], zf) and torch.equal(nll[1], lf)
    ops().check_flags(torch.device("cuda"), "register slots")


@pytest.mark.parametrize("B,N,D,K,kind", [(6, 703, 2, 8, "channel"), (33, 20, 2, 8, "channel"), (300, 1, 2, 8, "channel"),
                                          (9, 50, 2, 8, "channel_inv"), (7, 31, 2, 4, "channel"), (5, 64, 3, 2, "channel"),
                                          (4, 90, 2, 11, "channel"), (4, 90, 2, 10, "channel_inv")])
def test_whole_token_staging_is_bit_identical_to_span_staging(B, N, D, K, kind):
    """Where a transformed span plus one line covers the token's stride (D = 2: 104-byte spans at a 208-byte stride),
    forward / inverse stage whole tokens as one contiguous range; the arithmetic is the same, so are the results.  K = 10 / 11
    at D = 2: whole tokens (280 bytes x 64) no longer fit four stages in 64 KiB of LDS — the launch falls back to spans, it
    does not leave the token-pass kernel."""
    z, nn_out, sf, msf, mask, ln, pad = _case(B, N, D, K, kind, 321 + B + N)
    gk = dict(scaling_factor=g(sf), mixture_scaling_factor=g(msf), channel_padding_mask=g(pad))
    lib = _lib.load()
    outs = []
    for on in (1, 0):
        lib.cnf_set_mixture_whole_tokens(on)
        try:
            zf, lf, _ = ops().mixture_coupling(g(z), g(nn_out), g(mask), K, **gk)
            zr, lr, _ = ops().mixture_coupling(zf, g(nn_out), g(mask), K, reverse=True, **gk)
            nll = ops().mixture_coupling_nll(g(z), g(nn_out), g(mask), K, length=g(ln.float()), **gk)
        finally:
            lib.cnf_set_mixture_whole_tokens(1)
        outs.append((zf, lf, zr, lr, nll[0], nll[1], nll[4]))
    for a_, b_ in zip(*outs):
        assert torch.equal(a_, b_)
    zo, lo, _ = O.mixture_coupling(z, nn_out, mask, K, sf, msf, channel_padding_mask=pad)
    close(outs[0][0], zo, **ELEM)
    assert rel_ll(outs[0][1], lo) < 1e-4


@pytest.mark.parametrize("N,D,K,kind", [(64, 4, 10, "channel"), (288, 3, 51, "none"), (256, 3, 27, "none"), (38, 6, 16, "channel"),
                                        (20, 2, 8, "channel")])
def test_a_samples_result_does_not_depend_on_its_batch(N, D, K, kind):
    """The lanes-per-item / slots-per-lane choice of a run-time K is a function of K only and split rows meet in integer
    sums, so the latents of a sample are BIT-identical whether it is transformed alone, in a training-sized batch (rows
    split over workgroups) or in a large one (whole rows per wave), forward and inverse; its log-det to fp32 rounding of
    the row sum (the split and whole-row paths add the same terms in different groupings)."""
    big = 4096 if N * D <= 256 else 600
    z, nn_out, sf, msf, mask, ln, pad = _case(big, N, D, K, kind, 77 + N + K, padded=False)
    gk = dict(scaling_factor=g(sf), mixture_scaling_factor=g(msf))
    zb, lb, _ = ops().mixture_coupling(g(z), g(nn_out), g(mask), K, **gk)
    zrb, lrb, _ = ops().mixture_coupling(zb, g(nn_out), g(mask), K, reverse=True, **gk)
    for b in (1, 3, 64):
        zs, ls, _ = ops().mixture_coupling(g(z[:b]), g(nn_out[:b]), g(mask), K, **gk)
        assert torch.equal(zs, zb[:b])
        close(ls, lb[:b], rtol=2e-6, atol=2e-5)
        zr, lr, _ = ops().mixture_coupling(zb[:b].contiguous(), g(nn_out[:b]), g(mask), K, reverse=True, **gk)
        assert torch.equal(zr, zrb[:b])
        close(lr, lrb[:b], rtol=2e-6, atol=2e-5)


@pytest.mark.parametrize("B,N,D,K,kind", SHAPES)
def test_mixture_nll_epilogue_equals_coupling_then_prior_nll(B, N, D, K, kind):
    """cnf_mixture_coupling_nll == cnf_mixture_coupling followed by cnf_prior_nll on its outputs (z', ldj bit-equal;
    per-sample NLL within 1e-4 relative of the oracle's assembly, task.py:96-118)."""
    z, nn_out, sf, msf, mask, ln, pad = _case(B, N, D, K, kind, 17 + B + N + K)
    if pad is None:
        ln = torch.full((B,), N)
    gk = dict(scaling_factor=g(sf), mixture_scaling_factor=g(msf), channel_padding_mask=g(pad))
    ldj0 = torch.randn(B, generator=torch.Generator().manual_seed(6))
    zf, lf, _ = ops().mixture_coupling(g(z), g(nn_out), g(mask), K, ldj=g(ldj0), **gk)
    neglog_s, nll_s = ops().prior_nll(zf, lf, g(ln), g(pad))
    acc = torch.zeros(ops().NLL_ACC_SLOTS, dtype=torch.int64, device="cuda")
    zn, lnl, _, neglog, nll = ops().mixture_coupling_nll(g(z), g(nn_out), g(mask), K, ldj=g(ldj0), length=g(ln), acc=acc, **gk)
    assert torch.equal(zn, zf) and torch.equal(lnl, lf)
    close(neglog, neglog_s, rtol=1e-5, atol=1e-4)
    close(nll, nll_s, rtol=1e-5, atol=1e-5)
    zo, lo, _ = O.mixture_coupling(z, nn_out, mask, K, sf, msf, channel_padding_mask=pad)
    nll_o = O.nll_per_sample(zo, lo + ldj0, ln.float(), pad)
    assert rel_ll(nll, nll_o) < 1e-4
    sums = ops().nll_acc_read(acc, B)
    assert abs(sums[0].item() - nll.double().sum().item()) < 1e-6 * max(1.0, abs(nll.double().sum().item())) + 1e-6 * B
    assert sums[1].item() == B


def test_config_shapes_round_trip_at_full_size():
    """PTB (B=128,N=288,D=3,K=51, no mask), Zinc nodes / edges, graph colouring large: forward then inverse recovers z,
    ldj antisymmetric, a slice equals the oracle."""
    for (B, N, D, K, kind) in [(128, 288, 3, 51, "none"), (512, 38, 6, 16, "channel"), (512, 703, 2, 8, "channel"),
                               (128, 50, 6, 16, "channel"), (384, 20, 2, 8, "channel")]:
        gen = torch.Generator(device="cuda").manual_seed(B + N)
        z = torch.randn(B, N, D, generator=gen, device="cuda")
        nn_out = 0.5 * torch.randn(B, N, D * (2 + 3 * K), generator=gen, device="cuda")
        mask = None if kind == "none" else g(O.channel_mask(D))
        zf, lf, _ = ops().mixture_coupling(z, nn_out, mask, K)
        zr, lr, _ = ops().mixture_coupling(zf, nn_out, mask, K, reverse=True)
        assert (zr - z).abs().max().item() < 3e-4, (B, N, D, K)
        assert ((lf + lr).abs() / lf.abs().clamp(min=1.0)).max().item() < 1e-4
        zo, lo, _ = O.mixture_coupling(z[:4].cpu(), nn_out[:4].cpu(), None if mask is None else mask.cpu(), K, None, None)
        close(zf[:4], zo, **ELEM); loglik_close(lf[:4], lo)
        assert rel_ll(lf[:4], lo) < 1e-4
    ops().check_flags(torch.device("cuda"), "config shapes")


def test_unaligned_nn_out_takes_the_fallback_kernel():
    """A view that starts 4 bytes into an allocation is not 16-byte aligned: the DMA kernel declines, the round-1
    kernel serves it, results unchanged."""
    B, N, D, K = 6, 10, 4, 8
    z, nn_out, sf, msf, mask, ln, pad = _case(B, N, D, K, "channel", 77)
    buf = torch.empty(nn_out.numel() + 1, device="cuda")
    view = buf[1:].view_as(nn_out)
    view.copy_(nn_out)
    assert view.data_ptr() % 16 != 0
    zf, lf, _ = ops().mixture_coupling(g(z), view, g(mask), K, scaling_factor=g(sf), mixture_scaling_factor=g(msf),
                                       channel_padding_mask=g(pad))
    zo, lo, _ = O.mixture_coupling(z, nn_out, mask, K, sf, msf, channel_padding_mask=pad)
    close(zf, zo, **ELEM); loglik_close(lf, lo)


@pytest.mark.parametrize("B,N,D,K,kind", [(40, 16, 4, 8, "channel"), (24, 38, 6, 16, "channel"), (6, 703, 2, 8, "channel"),
                                          (16, 50, 6, 16, "channel_inv"), (70, 5, 3, 4, "channel"), (5, 97, 3, 9, "channel"),
                                          (130, 64, 6, 8, "channel"), (3, 400, 2, 23, "none"), (9, 21, 5, 8, "channel")])
@pytest.mark.parametrize("with_length", [True, False])
def test_coupling_actnorm_conv_fusion_is_bit_identical_to_the_chain(B, N, D, K, kind, with_length):
    """cnf_mixture_coupling_actconv == cnf_mixture_coupling_ws then cnf_actnorm_invconv (z and log-det bit for bit):
    the coupling of flow step i with the ActNorm + 1x1 conv of step i+1 (D = 5 has no epilogue build and takes the
    composed path inside the library; K = 23 takes the run-time-K kernel)."""
    z, nn_out, sf, msf, mask, ln, pad = _case(B, N, D, K, kind, 1000 + B + N + D)
    gen = torch.Generator().manual_seed(B * N + D)
    bias, scales = torch.randn(1, 1, D, generator=gen), 0.3 * torch.randn(1, 1, D, generator=gen)
    w = torch.linalg.qr(torch.randn(D, D, generator=gen))[0] * 1.1
    sldj = torch.slogdet(w)[1]
    ldj0 = torch.randn(B, generator=gen)
    length = g(ln) if with_length else None
    kw = dict(scaling_factor=g(sf), mixture_scaling_factor=g(msf), channel_padding_mask=g(pad), reg_max=3.5, reg_factor=2.0,
              is_training=True)
    z1, l1, _ = ops().mixture_coupling(g(z), g(nn_out), g(mask), K, ldj=g(ldj0), **kw)
    z2, l2 = ops().actnorm_invconv(z1, g(bias), g(scales), g(w), g(sldj), length=length, channel_padding_mask=g(pad), ldj=l1)
    zf, lf, _ = ops().mixture_coupling_actconv(g(z), g(nn_out), g(mask), K, g(bias), g(scales), g(w), g(sldj), length=length,
                                               ldj=g(ldj0), **kw)
    assert torch.equal(zf, z2), (zf - z2).abs().max().item()
    assert torch.equal(lf, l2), (lf - l2).abs().max().item()
    # and against the oracle's three layers
    zo, lo, _ = O.mixture_coupling(z, nn_out, mask, K, sf, msf, channel_padding_mask=pad, reg_max=3.5, reg_factor=2.0, is_training=True)
    okw = {}
    if pad is not None:
        okw["channel_padding_mask"] = pad
    if with_length:
        okw["length"] = ln
    za, la = O.actnorm(zo, bias, scales, ldj=lo + ldj0, **okw)
    zc, lc = O.invconv(za, w, sldj, ldj=la, **({k: v for k, v in okw.items()}))
    close(zf, zc, rtol=5e-5, atol=5e-5); loglik_close(lf, lc)


def test_flow_model_uses_the_fused_kernels_and_matches_the_layer_by_layer_pass():
    """FlowModel.forward / .nll on a set-modelling style stack (ActNorm, 1x1 conv, mixture coupling) x 3: with fusion
    (coupling_i + ActNorm_{i+1} + conv_{i+1}; last coupling + NLL) the outputs equal the unfused pass bit for bit, and the
    fused entry points are the ones that ran."""
    import torch.nn as nn
    from categoricalnf_amd.layers.flows.flow_model import FlowModel
    from categoricalnf_amd.layers.flows.activation_normalization import ActNormFlow
    from categoricalnf_amd.layers.flows.permutation_layers import InvertibleConv
    from categoricalnf_amd.layers.flows.mixture_cdf_layer import MixtureCDFCoupling
    from categoricalnf_amd.layers.flows.coupling_layer import CouplingLayer
    torch.manual_seed(0)
    D, K, B, N = 4, 8, 96, 16

    class Net(nn.Module):
        def __init__(self, c_out):
            super().__init__()
            self.lin = nn.Linear(D, c_out)

        def forward(self, x, **kw):
            return 0.3 * self.lin(x)

    layers = []
    for i in range(3):
        mask = CouplingLayer.create_channel_mask(D)
        layers += [ActNormFlow(D), InvertibleConv(D),
                   MixtureCDFCoupling(D, mask if i % 2 == 0 else 1 - mask, model_func=lambda c_out: Net(c_out), num_mixtures=K)]
    model = FlowModel(layers).cuda().eval()
    for m in model.modules():
        if isinstance(m, ActNormFlow):
            m.bias.data.normal_(); m.scales.data.normal_(std=0.2)
    z = torch.randn(B, N, D, device="cuda")
    ln = torch.randint(N // 2, N + 1, (B,), device="cuda")
    pad = (torch.arange(N, device="cuda")[None, :] < ln[:, None]).float().unsqueeze(-1)
    z = z * pad
    calls = []
    o = ops()
    real = o._launch

    # (with CNF_COMPACT_PARAMS=1 the layers run the compact-layout twins of the same entry points)
    twin = {"cnf_mixture_coupling_compact": "cnf_mixture_coupling_ws", "cnf_mixture_coupling_compact_nll": "cnf_mixture_coupling_nll",
            "cnf_mixture_coupling_compact_actconv": "cnf_mixture_coupling_actconv"}

    def spy(dev, name, *a, **kw):
        calls.append(twin.get(name, name))
        return real(dev, name, *a, **kw)
    o._launch = spy
    try:
        with torch.no_grad():
            zf, lf, nf = model.nll(z, length=ln, channel_padding_mask=pad)
            fused_calls = list(calls)
            o.FUSE_LAYERS = False
            calls.clear()
            zu, lu = model(z, length=ln, channel_padding_mask=pad)
            _, nu = o.prior_nll(zu, lu, ln, pad)
    finally:
        o._launch = real
        o.FUSE_LAYERS = True
    assert fused_calls.count("cnf_mixture_coupling_actconv") == 2 and fused_calls.count("cnf_mixture_coupling_nll") == 1
    assert fused_calls.count("cnf_actnorm_invconv") == 1 and "cnf_prior_nll" not in fused_calls
    assert "cnf_mixture_coupling_actconv" not in calls and calls.count("cnf_mixture_coupling_ws") == 3
    assert torch.equal(zf, zu) and torch.equal(lf, lu)
    close(nf, nu, rtol=1e-5, atol=1e-5)


def _bwd(kernel, z, nn_out, sf, msf, mask, pad, K, gz, gl, reg):
    """gradients of the mixture coupling through MixtureCouplingFn with the fp32 token-pass backward (kernel 0) or the
    fp64 backward (kernel 1 selects the round-1 forward and the fp64 backward)."""
    from categoricalnf_amd import functional as Fn
    lib = _lib.load()
    lib.cnf_set_mixture_kernel(kernel)
    try:
        zz, nn_ = g(z).requires_grad_(True), g(nn_out).requires_grad_(True)
        sf_, msf_ = g(sf).requires_grad_(True), g(msf).requires_grad_(True)
        zo, lo, _ = Fn.MixtureCouplingFn.apply(zz, nn_, sf_, msf_, None, g(mask), g(pad), K, reg[0], reg[1], True, True, True)
        torch.autograd.backward([zo, lo], [g(gz), g(gl)])
        return [t.grad.detach().cpu() for t in (zz, nn_, sf_, msf_)]
    finally:
        lib.cnf_set_mixture_kernel(0)


@pytest.mark.parametrize("B,N,D,K,kind", SHAPES)
def test_fp32_token_pass_backward_matches_fp64_backward(B, N, D, K, kind):
    """cnf_mixture_coupling_bwd_f32 (DMA-staged rows, in-place gradients, coalesced write-back, zeros for untransformed
    blocks written by the kernel) against the fp64 backward kernel: g_z, g_nn, g_scaling_factor, g_mixture_scaling_factor,
    with the CDF regulariser, padding, every mask kind, rows split over workgroups and 1 / 2 / 4 lanes per item."""
    z, nn_out, sf, msf, mask, ln, pad = _case(B, N, D, K, kind, 555 + B + N)
    gen = torch.Generator().manual_seed(B + K)
    gz, gl = torch.randn(B, N, D, generator=gen), torch.randn(B, generator=gen)
    new = _bwd(0, z, nn_out, sf, msf, mask, pad, K, gz, gl, (3.5, 2.0))
    old = _bwd(1, z, nn_out, sf, msf, mask, pad, K, gz, gl, (3.5, 2.0))
    for name, a, b in zip(("g_z", "g_nn", "g_sf", "g_msf"), new, old):
        scale = b.abs().max().item() + 1e-6
        err = (a - b).abs().max().item()
        assert err <= 2e-4 * scale + 2e-5, (name, err, scale)
    # untransformed parameter blocks get exact zeros
    P = 2 + 3 * K
    g_nn = new[1].reshape(B, N, D, P)
    if kind == "channel":
        assert g_nn[:, :, : D // 2].abs().max().item() == 0.0
    if pad is not None:
        dead = (pad.squeeze(-1) == 0)
        assert g_nn[dead].abs().max().item() == 0.0 if dead.any() else True


def test_fp32_backward_is_deterministic_and_handles_tails():
    """Same inputs twice -> bit-identical gradients (no atomics in the parameter-gradient reduction); inputs far in
    the tails (|z| up to 40 sigma) take the kernel's fp64 branch and still agree with the fp64 kernel."""
    B, N, D, K = 48, 16, 4, 8
    z, nn_out, sf, msf, mask, ln, pad = _case(B, N, D, K, "channel", 31337)
    z = z * 12.0
    gen = torch.Generator().manual_seed(9)
    gz, gl = torch.randn(B, N, D, generator=gen), torch.randn(B, generator=gen)
    a = _bwd(0, z, nn_out, sf, msf, mask, pad, K, gz, gl, (-1.0, 1.0))
    b = _bwd(0, z, nn_out, sf, msf, mask, pad, K, gz, gl, (-1.0, 1.0))
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    ref = _bwd(1, z, nn_out, sf, msf, mask, pad, K, gz, gl, (-1.0, 1.0))
    for name, x, y in zip(("g_z", "g_nn", "g_sf", "g_msf"), a, ref):
        fin = torch.isfinite(y)
        assert torch.equal(fin, torch.isfinite(x)), name
        scale = y[fin].abs().max().item() + 1e-6
        assert (x[fin] - y[fin]).abs().max().item() <= 3e-4 * scale + 2e-5, name


@pytest.mark.parametrize("B,N,D,K", [(48, 16, 4, 8), (256, 64, 6, 8), (16, 288, 3, 51)])
def test_fp64_backward_is_deterministic_too(B, N, D, K):
    """The reference-precision mixture backward (cnf_mixture_coupling_bwd, kernel 1 / math mode 0) meets its workgroup's
    parameter-gradient sums in 64-bit fixed-point LDS words (integer atomics: order-independent), so it is bit-reproducible
    like the fp32 kernels — rounds 1-4 added fp32 with atomicAdd there.  Three runs, torch.equal on every gradient; the
    split (static API) form through MixtureParamsFn / MixtureTransformFn likewise."""
    from categoricalnf_amd import functional as Fn
    z, nn_out, sf, msf, mask, ln, pad = _case(B, N, D, K, "channel", 99 + B)
    gen = torch.Generator().manual_seed(4)
    gz, gl = torch.randn(B, N, D, generator=gen), torch.randn(B, generator=gen)
    ref = _bwd(1, z, nn_out, sf, msf, mask, pad, K, gz, gl, (3.5, 2.0))
    for _ in range(2):
        for x, y in zip(ref, _bwd(1, z, nn_out, sf, msf, mask, pad, K, gz, gl, (3.5, 2.0))):
            assert torch.equal(x, y)
    # the fixed-point sums cost no accuracy against the fp32 token-pass kernel's fp64 row sums
    new = _bwd(0, z, nn_out, sf, msf, mask, pad, K, gz, gl, (3.5, 2.0))
    for name, x, y in zip(("g_sf", "g_msf"), ref[2:], new[2:]):
        scale = y.abs().max().item() + 1e-6
        assert (x - y).abs().max().item() <= 3e-4 * scale + 2e-5, name

    def split_grads():
        nn_ = g(nn_out).requires_grad_(True)
        sf_, msf_ = g(sf).requires_grad_(True), g(msf).requires_grad_(True)
        t, log_s, log_pi, mu, ls = Fn.MixtureParamsFn.apply(nn_, sf_, msf_, g(mask), K)
        torch.autograd.backward([t, log_s, log_pi, mu, ls], [torch.ones_like(t), 0.5 * torch.ones_like(log_s), torch.ones_like(log_pi),
                                                             0.25 * torch.ones_like(mu), torch.ones_like(ls)])
        return [x.grad.detach().cpu() for x in (nn_, sf_, msf_)]
    a = split_grads()
    for x, y in zip(a, split_grads()):
        assert torch.equal(x, y)


# ---- math mode 0 (fp64 like mixture_cdf_layer.py:62,173-178) on the same token passes ---------------------------------------

def _mode0(fn):
    lib = _lib.load()
    lib.cnf_set_math_mode(0)
    try:
        return fn()
    finally:
        lib.cnf_set_math_mode(1)
        lib.cnf_set_mixture_kernel(0)


@pytest.mark.parametrize("B,N,D,K,kind", SHAPES)
def test_fp64_token_pass_against_the_oracle_and_the_round1_fp64_kernel(B, N, D, K, kind):
    """Reference precision on the token-pass kernel: the oracle's fp64 results at fp32 output rounding, and the bits of the
    round-1 fp64 kernel for the forward's z (same fp64 expressions per element; the inverse is a Newton polish of the fp32
    root that meets the round-1 kernel's root to ~1e-11 of the smallest scale) — per-sample sums differ by their order only."""
    z, nn_out, sf, msf, mask, ln, pad = _case(B, N, D, K, kind, 3 * B + 5 * N + 7 * D + K)
    kw = dict(num_mixtures=K, reg_max=3.5, reg_factor=2.0, is_training=True)
    gk = dict(scaling_factor=g(sf), mixture_scaling_factor=g(msf), channel_padding_mask=g(pad), **kw)
    zo, lo, ro = O.mixture_coupling(z, nn_out, mask, scaling_factor=sf, mixture_scaling_factor=msf, channel_padding_mask=pad, **kw)
    lib = _lib.load()

    def run(which):
        lib.cnf_set_mixture_kernel(which)
        f = ops().mixture_coupling(g(z), g(nn_out), g(mask), **gk)
        r = ops().mixture_coupling(g(zo), g(nn_out), g(mask), reverse=True, **gk)
        return f, r
    (zf, lf, rf), (zr, lr, _) = _mode0(lambda: run(0))
    (z1, l1, r1), (zr1, lr1, _) = _mode0(lambda: run(1))
    # the bound parameters come from fp32 tanh (library versions differ by an ulp between host and device): 2e-6
    close(zf, zo, rtol=2e-6, atol=2e-6); loglik_close(lf, lo, rel=2e-6); loglik_close(rf, ro, rel=2e-6)
    assert torch.equal(zf, z1)
    # the inverse stops at the Newton step whose remaining error is below 1e-11 of the smallest scale (the round-1 kernel
    # iterates until the STEP is below 1e-10): the same double to ~1e-11, an fp32 ulp apart where that crosses a rounding boundary
    close(zr, zr1, rtol=3e-7, atol=1e-9)
    close(lf, l1, rtol=2e-6, atol=2e-6); close(lr, lr1, rtol=2e-6, atol=2e-6); close(rf, r1, rtol=2e-6, atol=2e-6)
    zo2, lo2, _ = O.mixture_coupling(zo, nn_out, mask, scaling_factor=sf, mixture_scaling_factor=msf,
                                     channel_padding_mask=pad, reverse=True, **kw)
    close(zr, zo2, rtol=1e-5, atol=1e-5); loglik_close(lr, lo2, rel=1e-5)
    ops().check_flags(torch.device("cuda"), "fp64 tok kernel")


def test_fp64_token_pass_tails_and_flat_regions():
    """Latents out to 12 sigma and narrow, far-apart components: the forward's clamps (safe_log, the log-space pdf) and the
    inverse's bisection fallback inside the widened quantile bracket give the round-1 fp64 kernel's z."""
    B, N, D, K = 64, 32, 4, 8
    gen = torch.Generator().manual_seed(77)
    z = 12.0 * torch.randn(B, N, D, generator=gen)
    nn_out = torch.randn(B, N, D * (2 + 3 * K), generator=gen)
    nn_out.view(B, N, D, 2 + 3 * K)[..., 2 + K:2 + 2 * K] *= 6.0          # means far apart
    nn_out.view(B, N, D, 2 + 3 * K)[..., 2 + 2 * K:] -= 2.5                # narrow components
    mask = O.channel_mask(D)
    lib = _lib.load()

    def run(which):
        lib.cnf_set_mixture_kernel(which)
        f = ops().mixture_coupling(g(z), g(nn_out), g(mask), K)
        r = ops().mixture_coupling(f[0], g(nn_out), g(mask), K, reverse=True)
        return f, r
    (zf, lf, _), (zr, lr, _) = _mode0(lambda: run(0))
    (z1, l1, _), (zr1, lr1, _) = _mode0(lambda: run(1))
    assert torch.equal(zf, z1)
    close(zr, zr1, rtol=1e-6, atol=1e-6)
    close(lf, l1, rtol=2e-6, atol=1e-5); close(lr, lr1, rtol=2e-6, atol=1e-5)
    # (no oracle here: with components this narrow and far apart the reference's own `1 - u` is rounding noise of its
    # log-space evaluation on the plateaus between them — tests/test_gpu_parity.py::test_mixture_fast_vs_exact has the
    # oracle comparison out to 12 sigma with ordinary parameters, and in math mode 0 it runs this kernel)


def test_fp64_token_pass_keeps_nans_where_the_round1_kernel_has_them():
    B, N, D, K = 8, 24, 4, 8
    z, nn_out, sf, msf, mask, ln, pad = _case(B, N, D, K, "channel", 5)
    z[1, 3, 3] = float("nan"); z[2, 5, 2] = float("inf")
    nn_out.view(B, N, D, 2 + 3 * K)[3, 7, 3, 2 + K + 1] = float("nan")       # a mean
    nn_out.view(B, N, D, 2 + 3 * K)[4, 9, 2, 2 + 2] = float("nan")           # a mixture weight
    lib = _lib.load()

    def run(which):
        lib.cnf_set_mixture_kernel(which)
        out = ops().mixture_coupling(g(z), g(nn_out), g(mask), K, scaling_factor=g(sf), mixture_scaling_factor=g(msf))
        inv = ops().mixture_coupling(g(z), g(nn_out), g(mask), K, scaling_factor=g(sf), mixture_scaling_factor=g(msf), reverse=True)
        word = ops().flag_word(torch.device("cuda", torch.cuda.current_device()))
        assert int(word.item()) & (_lib.FLAG_NAN_Z | _lib.FLAG_NAN_LDJ)
        word.zero_()
        return out, inv
    (zf, lf, _), (zr, lr, _) = _mode0(lambda: run(0))
    (z1, l1, _), (zr1, lr1, _) = _mode0(lambda: run(1))
    # z carries the NaNs element by element like the round-1 kernel, and so do the rows' log-dets: the fixed-point row sums of
    # the token-pass kernels cannot hold a NaN or an infinity, a row that met one is marked and comes out NaN
    assert torch.equal(torch.isnan(zf), torch.isnan(z1))
    ok = ~torch.isnan(z1)
    assert torch.equal(zf[ok], z1[ok])
    assert torch.isnan(lf[[1, 3, 4]]).all() and torch.isnan(lr[[1, 3, 4]]).all() and torch.isnan(l1[[1, 3, 4]]).all()
    assert not torch.isfinite(lf[2]) and not torch.isfinite(lr[2])                  # the row with an infinite latent
    clean = torch.tensor([0, 5, 6, 7])
    close(lf[clean], l1[clean], rtol=1e-5, atol=1e-5); close(lr[clean], lr1[clean], rtol=1e-5, atol=1e-5)
    # the same in the default fp32 mode, whole rows per wave and rows shared by several workgroups (long rows)
    for Bx, Nx in ((8, 24), (2, 1500)):
        zx, nnx, sfx, msfx, maskx, _, _ = _case(Bx, Nx, D, K, "channel", 9)
        zx[1, 3, 3] = float("nan")
        nnx.view(Bx, Nx, D, 2 + 3 * K)[0, 7, 3, 2 + K + 1] = float("nan")
        for rev in (False, True):
            _, lx, _ = ops().mixture_coupling(g(zx), g(nnx), g(maskx), K, scaling_factor=g(sfx), mixture_scaling_factor=g(msfx), reverse=rev)
            word = ops().flag_word(torch.device("cuda", torch.cuda.current_device()))
            assert int(word.item()) & _lib.FLAG_NAN_LDJ
            word.zero_()
            assert torch.isnan(lx[:2]).all() and torch.isfinite(lx[2:]).all()
    # ... and the workspace of the shared rows is handed back clean (ticket counters included)
    torch.cuda.synchronize()
    for w in ops()._mix_ws.values():
        assert int(w.count_nonzero().item()) == 0


def test_nontemporal_dma_loads_change_no_bit_at_the_north_star_size():
    """S* (B = 16384, N = 64, D = 6, K = 8: 312 MB of staged parameter spans) takes the nontemporal DMA loads by the 64 MB rule
    (cnf_set_mixture_nt_mb); forward and Newton inverse give the bits of the plain loads, and the round trip closes."""
    B, N, D, K = 16384, 64, 6, 8
    dev = torch.device("cuda")
    gen = torch.Generator(device=dev).manual_seed(11)
    z = torch.randn(B, N, D, generator=gen, device=dev)
    nn_out = 0.5 * torch.randn(B, N, D * (2 + 3 * K), generator=gen, device=dev)
    mask = g(O.channel_mask(D))
    lib = _lib.load()
    outs = []
    try:
        for mb in (0, -1):
            lib.cnf_set_mixture_nt_mb(mb)
            zf, lf, _ = ops().mixture_coupling(z, nn_out, mask, K)
            zr, lr, _ = ops().mixture_coupling(zf, nn_out, mask, K, reverse=True)
            outs.append((zf, lf, zr, lr))
    finally:
        lib.cnf_set_mixture_nt_mb(-1)
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    assert (outs[1][2] - z).abs().max().item() < 5e-4
    assert (outs[1][1] + outs[1][3]).abs().max().item() < 2e-2          # log-dets of forward and inverse cancel (sums of 192 terms)
    ops().check_flags(dev, "north-star size")
