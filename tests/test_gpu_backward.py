"""The streaming backward kernels (csrc/cnf_backward.hip) on seeded random shapes against autograd through the CPU oracle
(the reference has no backward code: it differentiates its eager op chains, general/train.py:144-155), their run-to-run
bit-reproducibility, and the NULL-upstream-gradient forms of the C ABI.

Shapes cover every chunk width (row lengths that are multiples of 4, of 2, odd), chess masks over odd position counts,
1..10 channels (templated and generic 1x1 convolution), a single row, rows shorter than a wave, padding and lengths.
Tolerance: 2e-4 of the gradient's scale (the golden-vector tests of test_gpu_parity.py hold the tight, reference-noise
based budget; here the point is shape coverage)."""
import ctypes

import numpy as np
import pytest
import torch

from categoricalnf_amd import _lib
from oracle import cnf_oracle as O

pytestmark = pytest.mark.gpu


def g(t):
    return None if t is None else t.cuda()


def grad_close(a, b, what, rel=2e-4):
    a, b = a.detach().double().cpu().reshape(-1), b.detach().double().cpu().reshape(-1)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    if a.numel() == 0:
        return
    scale = max(b.abs().max().item(), 1.0)
    worst = (a - b).abs().max().item()
    assert worst <= rel * scale, "%s: |dev| %.3g > %.1g x scale %.3g" % (what, worst, rel, scale)


def _shapes(seed, n, dims=(1, 2, 3, 4, 5, 6, 7, 8, 10)):
    r = np.random.RandomState(seed)
    out = []
    for _ in range(n):
        B = int(r.choice([1, 2, 5, 33, 64, 129, int(r.randint(1, 300))]))
        N = int(r.choice([1, 2, 3, 5, 8, 16, 17, 64, int(r.randint(1, 50))]))
        D = int(r.choice(dims))
        out.append((B, N, D, int(r.randint(0, 1 << 30))))
    return out


def _leaf(t, dev=False):
    t = t.clone()
    return (t.cuda() if dev else t).requires_grad_(True)


def _mask(kind, D):
    if kind == "chess" or D == 1:
        return O.chess_mask()
    return O.channel_mask(D)


@pytest.mark.parametrize("B,N,D,seed", _shapes(11, 36))
@pytest.mark.parametrize("mode", [1, 0])
def test_affine_backward_vs_oracle_autograd(B, N, D, seed, mode):
    from categoricalnf_amd import functional as Fn
    gen = torch.Generator().manual_seed(seed)
    z, nn_out = 1.2 * torch.randn(B, N, D, generator=gen), 0.7 * torch.randn(B, N, 2 * D, generator=gen)
    sf = 0.5 * torch.randn(D, generator=gen) if seed % 4 else None      # both signs: the clamp(min=1) branch and the other
    mask = _mask(["channel", "chess"][seed % 2], D)
    ldj0 = torch.randn(B, generator=gen)
    wz, wl = torch.randn(B, N, D, generator=gen), torch.randn(B, generator=gen)
    reverse = bool((seed >> 3) % 2)
    lib = _lib.load()
    lib.cnf_set_math_mode(mode)
    try:
        zc, nc, lc = _leaf(z), _leaf(nn_out), _leaf(ldj0)
        sc = _leaf(sf) if sf is not None else None
        zo, lo = O.affine_coupling(zc, nc, mask, sc, reverse=reverse, ldj=lc)
        ((zo * wz).sum() + (lo * wl).sum()).backward()
        zg, ng, lg = _leaf(z, True), _leaf(nn_out, True), _leaf(ldj0, True)
        sg = _leaf(sf, True) if sf is not None else None
        zh, lh = Fn.AffineCouplingFn.apply(zg, ng, sg, lg, g(mask), reverse)
        ((zh * g(wz)).sum() + (lh * g(wl)).sum()).backward()
        grad_close(zg.grad, zc.grad, "g_z"); grad_close(ng.grad, nc.grad, "g_nn"); grad_close(lg.grad, lc.grad, "g_ldj")
        if sf is not None:
            grad_close(sg.grad, sc.grad, "g_scaling_factor", rel=5e-4)
    finally:
        lib.cnf_set_math_mode(1)


@pytest.mark.parametrize("B,N,D,seed", _shapes(12, 30))
def test_actnorm_invconv_backward_vs_oracle_autograd(B, N, D, seed):
    from categoricalnf_amd import functional as Fn
    gen = torch.Generator().manual_seed(seed)
    z = torch.randn(B, N, D, generator=gen)
    bias, scales = torch.randn(1, 1, D, generator=gen), 0.3 * torch.randn(1, 1, D, generator=gen)
    w = torch.linalg.qr(torch.randn(D, D, generator=gen))[0].contiguous() + 0.05 * torch.randn(D, D, generator=gen)
    sldj = torch.slogdet(w)[1].detach()
    ln = torch.randint(1, N + 1, (B,), generator=gen)
    ln[0] = N
    pad = O.length_mask(ln, N) if seed % 3 else None
    length = ln.float() if seed % 2 else None
    ldj0, wz, wl = torch.randn(B, generator=gen), torch.randn(B, N, D, generator=gen), torch.randn(B, generator=gen)
    reverse = bool((seed >> 2) % 2)
    # ActNorm
    zc, bc, sc, lc = _leaf(z), _leaf(bias), _leaf(scales), _leaf(ldj0)
    zo, lo = O.actnorm(zc, bc, sc, reverse=reverse, length=length, channel_padding_mask=pad, ldj=lc * 1.0)
    ((zo * wz).sum() + (lo * wl).sum()).backward()
    zg, bg, sg, lg = _leaf(z, True), _leaf(bias, True), _leaf(scales, True), _leaf(ldj0, True)
    zh, lh = Fn.ActNormFn.apply(zg, bg, sg, lg, g(length), g(pad), reverse)
    ((zh * g(wz)).sum() + (lh * g(wl)).sum()).backward()
    grad_close(zg.grad, zc.grad, "actnorm g_z"); grad_close(lg.grad, lc.grad, "actnorm g_ldj")
    grad_close(bg.grad, bc.grad, "g_bias", rel=5e-4); grad_close(sg.grad, sc.grad, "g_scales", rel=5e-4)
    # 1x1 convolution (the matrix that is applied; the module inverts it outside the kernel)
    zc, wc, slc, lc = _leaf(z), _leaf(w), _leaf(sldj), _leaf(ldj0)
    zo, lo = O.invconv(zc, wc, slc, reverse=False, length=length, channel_padding_mask=pad, ldj=lc)
    ((zo * wz).sum() + (lo * wl).sum()).backward()
    zg, wg, slg, lg = _leaf(z, True), _leaf(w, True), _leaf(sldj, True), _leaf(ldj0, True)
    zh, lh = Fn.InvConvFn.apply(zg, wg, slg, lg, g(length), g(pad), False)
    ((zh * g(wz)).sum() + (lh * g(wl)).sum()).backward()
    grad_close(zg.grad, zc.grad, "conv g_x"); grad_close(lg.grad, lc.grad, "conv g_ldj")
    grad_close(wg.grad, wc.grad, "g_weight", rel=5e-4); grad_close(slg.grad, slc.grad, "g_sldj", rel=5e-4)


@pytest.mark.parametrize("B,N,D,seed", _shapes(13, 24))
@pytest.mark.parametrize("mode", [1, 0])
def test_ext_actnorm_prior_sigmoid_backward_vs_oracle_autograd(B, N, D, seed, mode):
    from categoricalnf_amd import functional as Fn
    gen = torch.Generator().manual_seed(seed)
    z, nn_out = torch.randn(B, N, D, generator=gen), 0.6 * torch.randn(B, N, 2 * D, generator=gen)
    ln = torch.randint(1, N + 1, (B,), generator=gen)
    ln[0] = N
    pad = O.length_mask(ln, N) if seed % 2 else None
    ldj0, wz, wl = torch.randn(B, generator=gen), torch.randn(B, N, D, generator=gen), torch.randn(B, generator=gen)
    reverse = bool((seed >> 2) % 2)
    lib = _lib.load()
    lib.cnf_set_math_mode(mode)
    try:
        zc, nc, lc = _leaf(z), _leaf(nn_out), _leaf(ldj0)
        zo, lo = O.ext_actnorm(zc, nc, reverse=reverse, channel_padding_mask=pad, ldj=lc * 1.0)
        ((zo * wz).sum() + (lo * wl).sum()).backward()
        zg, ng, lg = _leaf(z, True), _leaf(nn_out, True), _leaf(ldj0, True)
        zh, lh = Fn.ExtActNormFn.apply(zg, ng, lg, g(pad), reverse)
        ((zh * g(wz)).sum() + (lh * g(wl)).sum()).backward()
        grad_close(zg.grad, zc.grad, "ext g_z"); grad_close(ng.grad, nc.grad, "ext g_nn"); grad_close(lg.grad, lc.grad, "ext g_ldj")
        # prior NLL
        zc, lc = _leaf(z), _leaf(ldj0)
        (O.nll_per_sample(zc, lc, ln.float(), pad) * wl).sum().backward()
        zg, lg = _leaf(z, True), _leaf(ldj0, True)
        (Fn.PriorNllFn.apply(zg, lg, g(ln.float()), g(pad)) * g(wl)).sum().backward()
        grad_close(zg.grad, zc.grad, "nll g_z"); grad_close(lg.grad, lc.grad, "nll g_ldj")
        # logistic log-prob, sigmoid flow (both orientations)
        zc = _leaf(z)
        (O.logistic_log_prob(zc) * wz).sum().backward()
        zg = _leaf(z, True)
        (Fn.LogisticLogProbFn.apply(zg, 0.0, O.LOGISTIC_SIGMA, O.LOGISTIC_LOG_SIGMA) * g(wz)).sum().backward()
        grad_close(zg.grad, zc.grad, "log_prob g_x")
        for rev, x in ((False, z), (True, torch.rand(B, N, D, generator=gen) * 0.98 + 0.01)):
            xc, lc = _leaf(x), _leaf(ldj0)
            zo, lo = O.sigmoid_flow(xc, reverse=rev, ldj=lc)
            ((zo * wz).sum() + (lo * wl).sum()).backward()
            xg, lg = _leaf(x, True), _leaf(ldj0, True)
            zh, lh = Fn.SigmoidFlowFn.apply(xg, lg, rev, 1e-5)
            ((zh * g(wz)).sum() + (lh * g(wl)).sum()).backward()
            grad_close(xg.grad, xc.grad, "sigmoid g_z rev=%s" % rev, rel=1e-3 if rev else 2e-4)
            grad_close(lg.grad, lc.grad, "sigmoid g_ldj")
    finally:
        lib.cnf_set_math_mode(1)


def _ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


@pytest.mark.parametrize("B,N,D", [(2048, 64, 6), (300, 17, 5), (64, 703, 2), (1000, 33, 1), (257, 16, 7)])
def test_parameter_gradients_are_bit_identical_over_repeated_runs(B, N, D):
    """Every backward kernel that reduces over the batch (affine both directions, its split form, ActNorm, 1x1 conv —
    templated and generic): the parameter gradients of 3 runs on the same inputs are torch.equal, and so are the
    element-wise gradients.  (The sums go through lane-private LDS words / registers, fixed-order DPP sums per wave and an
    fp64 reduction in row order: no floating-point atomic anywhere.)"""
    lib = _lib.load()
    dev = torch.device("cuda:0")
    gen = torch.Generator(device=dev).manual_seed(B + N + D)
    rn = lambda *s: torch.randn(*s, generator=gen, device=dev)
    zo, nn2, gzo, gl = rn(B, N, D), 0.5 * rn(B, N, 2 * D), rn(B, N, D), rn(B)
    gs_, gt_ = rn(B, N, D), rn(B, N, D)
    mask = (O.chess_mask() if D == 1 else O.channel_mask(D)).to(dev).contiguous()
    mr, mc = mask.shape
    sf, bias, scales = 0.3 * rn(D), rn(D), 0.2 * rn(D)
    w = rn(D, D).contiguous()
    pad = (torch.rand(B, N, generator=gen, device=dev) > 0.2).float()
    ln = pad.sum(1)
    ws = torch.empty(int(lib.cnf_bwd_workspace_floats(D * D + 1)), device=dev)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def run_all():
        out = {}
        for rev in (0, 1):
            gz, gnn, gsf = torch.empty_like(zo), torch.empty_like(nn2), torch.empty(D, device=dev)
            assert lib.cnf_affine_coupling_bwd(_ptr(zo), _ptr(nn2), _ptr(sf), _ptr(mask), mr, mc, _ptr(gzo), _ptr(gl), _ptr(gz), _ptr(gnn),
                                               _ptr(gsf), _ptr(ws), B, N, D, rev, st) == 0, lib.cnf_last_error()
            out["affine%d" % rev] = (gz, gnn, gsf)
        gnn, gsf = torch.empty_like(nn2), torch.empty(D, device=dev)
        assert lib.cnf_affine_params_bwd(_ptr(nn2), _ptr(sf), _ptr(mask), mr, mc, _ptr(gs_), _ptr(gt_), _ptr(gnn), _ptr(gsf), _ptr(ws),
                                         B, N, D, st) == 0, lib.cnf_last_error()
        out["affine_params"] = (gnn, gsf)
        for rev in (0, 1):
            gz, gb, gsc = torch.empty_like(zo), torch.empty(D, device=dev), torch.empty(D, device=dev)
            assert lib.cnf_actnorm_bwd(_ptr(zo), _ptr(bias), _ptr(scales), _ptr(pad), _ptr(ln), _ptr(gzo), _ptr(gl), _ptr(gz), _ptr(gb),
                                       _ptr(gsc), _ptr(ws), B, N, D, rev, st) == 0, lib.cnf_last_error()
            out["actnorm%d" % rev] = (gz, gb, gsc)
        gx, gw, gsl = torch.empty_like(zo), torch.empty(D, D, device=dev), torch.empty(1, device=dev)
        assert lib.cnf_invconv_bwd(_ptr(zo), _ptr(w), _ptr(pad), _ptr(ln), _ptr(gzo), _ptr(gl), _ptr(gx), _ptr(gw), _ptr(gsl), _ptr(ws),
                                   B, N, D, 0, st) == 0, lib.cnf_last_error()
        out["invconv"] = (gx, gw, gsl)
        torch.cuda.synchronize()
        return out

    ref = run_all()
    for _ in range(2):
        cur = run_all()
        for k in ref:
            for a, b in zip(ref[k], cur[k]):
                assert torch.equal(a, b), "%s differs between runs" % k
    for k in ref:
        for t in ref[k]:
            assert torch.isfinite(t).all(), k


def test_missing_upstream_gradients_are_zeros():
    """g_zout == NULL / g_ldj == NULL of the C ABI (include/cnf_hip.h: 'either may be NULL = zero') against explicit zeros."""
    lib = _lib.load()
    dev = torch.device("cuda:0")
    B, N, D = 37, 19, 6
    gen = torch.Generator(device=dev).manual_seed(5)
    rn = lambda *s: torch.randn(*s, generator=gen, device=dev)
    zo, nn2, gzo, gl = rn(B, N, D), 0.5 * rn(B, N, 2 * D), rn(B, N, D), rn(B)
    mask = O.channel_mask(D).to(dev).contiguous()
    sf, bias, scales, w = 0.3 * rn(D), rn(D), 0.2 * rn(D), rn(D, D).contiguous()
    ws = torch.empty(int(lib.cnf_bwd_workspace_floats(D * D + 1)), device=dev)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    zeros_z, zeros_l = torch.zeros_like(gzo), torch.zeros_like(gl)

    def affine(gz_in, gl_in):
        gz, gnn, gsf = torch.empty_like(zo), torch.empty_like(nn2), torch.empty(D, device=dev)
        assert lib.cnf_affine_coupling_bwd(_ptr(zo), _ptr(nn2), _ptr(sf), _ptr(mask), 1, D, _ptr(gz_in), _ptr(gl_in), _ptr(gz), _ptr(gnn),
                                           _ptr(gsf), _ptr(ws), B, N, D, 0, st) == 0
        return gz, gnn, gsf

    def actnorm(gz_in, gl_in):
        gz, gb, gsc = torch.empty_like(zo), torch.empty(D, device=dev), torch.empty(D, device=dev)
        assert lib.cnf_actnorm_bwd(_ptr(zo), _ptr(bias), _ptr(scales), None, None, _ptr(gz_in), _ptr(gl_in), _ptr(gz), _ptr(gb), _ptr(gsc),
                                   _ptr(ws), B, N, D, 0, st) == 0
        return gz, gb, gsc

    def conv(gz_in, gl_in):
        gx, gw, gsl = torch.empty_like(zo), torch.empty(D, D, device=dev), torch.empty(1, device=dev)
        assert lib.cnf_invconv_bwd(_ptr(zo), _ptr(w), None, None, _ptr(gz_in), _ptr(gl_in), _ptr(gx), _ptr(gw), _ptr(gsl), _ptr(ws),
                                   B, N, D, 0, st) == 0
        return gx, gw, gsl

    def ext(gz_in, gl_in):
        gz, gnn = torch.empty_like(zo), torch.empty_like(nn2)
        assert lib.cnf_ext_actnorm_bwd(_ptr(zo), _ptr(nn2), None, _ptr(gz_in), _ptr(gl_in), _ptr(gz), _ptr(gnn), B, N, D, 0, st) == 0
        return gz, gnn

    for fn in (affine, actnorm, conv, ext):
        for a, b in ((None, gl), (gzo, None)):
            got = fn(a, b)
            want = fn(zeros_z if a is None else a, zeros_l if b is None else b)
            torch.cuda.synchronize()
            for x, y in zip(got, want):
                assert torch.equal(x, y), fn.__name__


# ---- fused training groups (functional.ActConvFn / MixtureActConvFn / EncoderActConvFn) ---------------------------------------
def _actconv_params(D, gen):
    bias, scales = torch.randn(1, 1, D, generator=gen), 0.3 * torch.randn(1, 1, D, generator=gen)
    w = torch.linalg.qr(torch.randn(D, D, generator=gen))[0].contiguous() + 0.05 * torch.randn(D, D, generator=gen)
    return bias, scales, w, torch.slogdet(w)[1].detach()


def _pad_len(B, N, seed, gen):
    ln = torch.randint(1, N + 1, (B,), generator=gen)
    ln[0] = N
    pad = O.length_mask(ln, N) if seed % 3 else None
    length = ln.float() if seed % 2 else None
    return pad, length


@pytest.mark.parametrize("B,N,D,seed", _shapes(21, 30, dims=(1, 2, 3, 4, 5, 6, 8)) + [(512, 64, 6, 5), (300, 38, 6, 7), (256, 16, 4, 6)])
def test_actconv_fn_vs_oracle_autograd(B, N, D, seed):
    """ActNorm -> 1x1 convolution as one Function (one forward, ONE backward kernel that recomputes the intermediate from the
    saved input) against autograd through O.actnorm -> O.invconv; and its output against the two layers' kernels (equal bits)."""
    from categoricalnf_amd import functional as Fn
    gen = torch.Generator().manual_seed(seed)
    z = torch.randn(B, N, D, generator=gen)
    bias, scales, w, sldj = _actconv_params(D, gen)
    pad, length = _pad_len(B, N, seed, gen)
    ldj0, wz, wl = torch.randn(B, generator=gen), torch.randn(B, N, D, generator=gen), torch.randn(B, generator=gen)
    zc, bc, sc, wc, slc, lc = _leaf(z), _leaf(bias), _leaf(scales), _leaf(w), _leaf(sldj), _leaf(ldj0)
    za, la = O.actnorm(zc, bc, sc, length=length, channel_padding_mask=pad, ldj=lc * 1.0)
    zo, lo = O.invconv(za, wc, slc, length=length, channel_padding_mask=pad, ldj=la)
    ((zo * wz).sum() + (lo * wl).sum()).backward()
    zg, bg, sg, wg, slg, lg = _leaf(z, True), _leaf(bias, True), _leaf(scales, True), _leaf(w, True), _leaf(sldj, True), _leaf(ldj0, True)
    zh, lh = Fn.ActConvFn.apply(zg, bg, sg, wg, slg, lg, g(length), g(pad))
    ((zh * g(wz)).sum() + (lh * g(wl)).sum()).backward()
    grad_close(zh, zo, "z_out", rel=2e-5); grad_close(lh, lo, "ldj_out", rel=2e-5)
    grad_close(zg.grad, zc.grad, "g_z"); grad_close(lg.grad, lc.grad, "g_ldj")
    grad_close(bg.grad, bc.grad, "g_bias", rel=5e-4); grad_close(sg.grad, sc.grad, "g_scales", rel=5e-4)
    grad_close(wg.grad, wc.grad, "g_weight", rel=5e-4); grad_close(slg.grad, slc.grad, "g_sldj", rel=5e-4)
    # the two layers' Functions: same forward bits; gradients to the rounding of another summation order
    z2, b2, s2, w2, sl2, l2 = _leaf(z, True), _leaf(bias, True), _leaf(scales, True), _leaf(w, True), _leaf(sldj, True), _leaf(ldj0, True)
    ya, la2 = Fn.ActNormFn.apply(z2, b2, s2, l2, g(length), g(pad), False)
    yo, lo2 = Fn.InvConvFn.apply(ya, w2, sl2, la2, g(length), g(pad), False)
    ((yo * g(wz)).sum() + (lo2 * g(wl)).sum()).backward()
    assert torch.equal(yo, zh) and torch.equal(lo2, lh)
    grad_close(zg.grad, z2.grad, "g_z vs chain", rel=1e-5); grad_close(wg.grad, w2.grad, "g_weight vs chain", rel=2e-5)
    grad_close(bg.grad, b2.grad, "g_bias vs chain", rel=2e-5); grad_close(sg.grad, s2.grad, "g_scales vs chain", rel=2e-5)


@pytest.mark.parametrize("B,N,D,K,seed", [(64, 16, 4, 8, 1), (33, 17, 6, 4, 2), (128, 8, 2, 8, 3), (20, 38, 6, 5, 4), (8, 5, 5, 3, 5), (256, 16, 4, 8, 6),
                                         (5, 64, 3, 16, 7), (64, 64, 6, 8, 9)])
def test_mixture_actconv_fn_vs_oracle_autograd(B, N, D, K, seed):
    """Mixture coupling + ActNorm + 1x1 convolution as one Function: the backward recovers the coupling's output from the
    group's output (W^-1 on the device) — against autograd through O.mixture_coupling -> O.actnorm -> O.invconv."""
    from categoricalnf_amd import functional as Fn
    gen = torch.Generator().manual_seed(seed)
    z = torch.randn(B, N, D, generator=gen)
    nn_out = 0.5 * torch.randn(B, N, D * (2 + 3 * K), generator=gen)
    sf, msf = 0.2 * torch.randn(D, generator=gen), 0.2 * torch.randn(D, K, generator=gen)
    mask = _mask("channel", D)
    bias, scales, w, sldj = _actconv_params(D, gen)
    pad, length = _pad_len(B, N, seed, gen)
    ldj0, wz, wl = torch.randn(B, generator=gen), torch.randn(B, N, D, generator=gen), torch.randn(B, generator=gen)
    leaves = [_leaf(t) for t in (z, nn_out, sf, msf, bias, scales, w, sldj, ldj0)]
    zc, nc, sfc, msfc, bc, sc, wc, slc, lc = leaves
    z1, l1, _ = O.mixture_coupling(zc, nc, mask, K, sfc, msfc, channel_padding_mask=pad)
    za, la = O.actnorm(z1, bc, sc, length=length, channel_padding_mask=pad, ldj=lc + l1)
    zo, lo = O.invconv(za, wc, slc, length=length, channel_padding_mask=pad, ldj=la)
    ((zo * wz).sum() + (lo * wl).sum()).backward()
    dl = [_leaf(t, True) for t in (z, nn_out, sf, msf, bias, scales, w, sldj, ldj0)]
    zg, ng, sfg, msfg, bg, sg, wg, slg, lg = dl
    zh, lh = Fn.MixtureActConvFn.apply(zg, ng, sfg, msfg, bg, sg, wg, slg, lg, g(mask), g(pad), g(length), K, -1.0, 1.0, True)
    ((zh * g(wz)).sum() + (lh * g(wl)).sum()).backward()
    grad_close(zh, zo, "z_out", rel=5e-5); grad_close(lh, lo, "ldj_out", rel=5e-5)
    for name, a, b in zip(("g_z", "g_nn", "g_sf", "g_msf", "g_bias", "g_scales", "g_weight", "g_sldj", "g_ldj"), dl, leaves):
        grad_close(a.grad, b.grad, name, rel=5e-4)
    # twice: the same bits
    first = [t.grad.clone() for t in dl]
    for t in dl:
        t.grad = None
    zh, lh = Fn.MixtureActConvFn.apply(zg, ng, sfg, msfg, bg, sg, wg, slg, lg, g(mask), g(pad), g(length), K, -1.0, 1.0, True)
    ((zh * g(wz)).sum() + (lh * g(wl)).sum()).backward()
    assert all(torch.equal(a, t.grad) for a, t in zip(first, dl))


@pytest.mark.parametrize("B,N,D,C,seed", [(64, 16, 6, 16, 1), (33, 17, 4, 7, 2), (16, 38, 6, 51, 3), (128, 8, 2, 5, 4), (9, 30, 8, 300, 5), (7, 5, 3, 12, 6)])
def test_encoder_actconv_fn_vs_oracle_autograd(B, N, D, C, seed):
    """Encoder (sampling its noise from the uniform draw) + ActNorm + 1x1 convolution as one Function against autograd through
    O.encoder_forward -> O.actnorm -> O.invconv (class-table gradient, the pair's parameter gradients, the log-det's)."""
    from categoricalnf_amd import functional as Fn
    gen = torch.Generator().manual_seed(seed)
    categ = torch.randint(0, C, (B, N), generator=gen)
    u = torch.rand(B * N, D, generator=gen)
    table = torch.randn(C, 2 * D, generator=gen)
    prior = torch.log_softmax(torch.randn(C, generator=gen), 0)
    bias, scales, w, sldj = _actconv_params(D, gen)
    pad, length = _pad_len(B, N, seed, gen)
    beta = 1.0 + 0.5 * (seed % 2)
    ldj0, wz, wl = torch.randn(B, generator=gen), torch.randn(B, N, D, generator=gen), torch.randn(B, generator=gen)
    leaves = [_leaf(t) for t in (table, bias, scales, w, sldj, ldj0)]
    tc, bc, sc, wc, slc, lc = leaves
    eps = O.logistic_from_uniform(u.reshape(B * N, 1, D))
    z1, l1, _ = O.encoder_forward(categ, eps, tc, prior, beta=beta, channel_padding_mask=pad)
    za, la = O.actnorm(z1, bc, sc, length=length, channel_padding_mask=pad, ldj=lc + l1)
    zo, lo = O.invconv(za, wc, slc, length=length, channel_padding_mask=pad, ldj=la)
    ((zo * wz).sum() + (lo * wl).sum()).backward()
    dl = [_leaf(t, True) for t in (table, bias, scales, w, sldj, ldj0)]
    tg, bg, sg, wg, slg, lg = dl
    zh, lh = Fn.EncoderActConvFn.apply(tg, bg, sg, wg, slg, lg, g(categ), g(u), g(prior), g(pad), g(length), beta, 1e-4)
    ((zh * g(wz)).sum() + (lh * g(wl)).sum()).backward()
    grad_close(zh, zo, "z_out", rel=5e-5); grad_close(lh, lo, "ldj_out", rel=5e-5)
    for name, a, b in zip(("g_table", "g_bias", "g_scales", "g_weight", "g_sldj", "g_ldj"), dl, leaves):
        grad_close(a.grad, b.grad, name, rel=5e-4)


def test_actconv_backward_is_bit_reproducible_and_inverts_on_the_device():
    """cnf_actnorm_invconv_bwd through the C ABI at a large shape: both forms (intermediate from the input / from the output through
    the device-side fp64 inverse) give the same gradients to rounding, each the same bits over repeated runs."""
    from categoricalnf_amd import functional as Fn, ops
    gen = torch.Generator().manual_seed(3)
    B, N, D = 2048, 64, 6
    z = g(torch.randn(B, N, D, generator=gen))
    bias, scales, w, sldj = (g(t) for t in _actconv_params(D, gen))
    gz, gl = g(torch.randn(B, N, D, generator=gen)), g(torch.randn(B, generator=gen))
    out, _ = ops.actnorm_invconv(z, bias, scales, w, sldj)
    runs = []
    for saved, is_out in ((z, False), (out, True)):
        res = [tuple(t.clone() for t in Fn._actconv_bwd(saved, is_out, bias, scales, w, None, None, gz, gl, Fn._Hold())) for _ in range(3)]
        assert all(torch.equal(a, b) for r in res[1:] for a, b in zip(res[0], r))
        runs.append(res[0])
    for name, a, b in zip(("g_z", "g_bias", "g_scales", "g_weight", "g_sldj"), *runs):
        grad_close(a, b, name + " from the output vs from the input", rel=2e-5)


def test_flow_training_pass_with_and_without_the_fused_groups():
    """FlowModel with autograd on: the fused groups (encoder + ActNorm + conv, coupling + ActNorm + conv — one Function and one
    backward each, ops.FUSE_TRAINING) against one Function per layer on the set-modelling flow, same noise: the loss to
    rounding, every parameter gradient within 1e-4 of its scale; and far fewer autograd nodes."""
    import contextlib, io
    from categoricalnf_amd import functional as Fn, ops
    from categoricalnf_amd.experiments.set_modeling import FlowSetModeling, SetShufflingDataset
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    params = {"set_size": 16, "coupling_hidden_layers": 2, "coupling_hidden_size": 64, "coupling_num_flows": 4,
              "coupling_mask_ratio": 0.5, "coupling_num_mixtures": 8,
              "categ_encoding": {"use_dequantization": False, "use_variational": False, "use_decoder": False, "num_dimensions": 4,
                                 "flow_config": {"num_flows": 0}, "decoder_config": {}}}
    with contextlib.redirect_stdout(io.StringIO()):
        model = FlowSetModeling(params, SetShufflingDataset).to(dev).train()
    rng = np.random.RandomState(1)
    B, S, D = 96, 16, 4
    draw = lambda: torch.from_numpy(np.stack([rng.permutation(S) for _ in range(B)])).long().to(dev)
    ln = torch.full((B,), S, dtype=torch.long, device=dev)
    with contextlib.redirect_stdout(io.StringIO()):
        model.initialize_data_dependent([(draw(), {"length": ln}) for _ in range(4)])
    x, noise = draw(), torch.rand(B * S, 1, D, device=dev)
    plist = [p for p in model.parameters() if p.requires_grad]

    def count_nodes(t):
        seen, stack = set(), [t.grad_fn]
        while stack:
            n = stack.pop()
            if n is None or n in seen:
                continue
            seen.add(n)
            stack.extend(f for f, _ in n.next_functions)
        return len(seen), sum(type(n).__name__.endswith("FnBackward") for n in seen)

    out = {}
    for fused in (True, False):
        ops.FUSE_TRAINING = fused
        try:
            if fused:                    # + the NLL assembly inside the last coupling layer's Function
                loss = model.nll_loss(x, length=ln, beta=1, noise=noise)[2].mean()
            else:
                z, ldj = model(x, reverse=False, length=ln, beta=1, noise=noise)
                loss = Fn.PriorNllFn.apply(z, ldj, ln, None).mean()
            out[fused] = (loss.detach().clone(), torch.autograd.grad(loss, plist, allow_unused=True), count_nodes(loss))
        finally:
            ops.FUSE_TRAINING = True
    (la, ga, na), (lb, gb, nb) = out[True], out[False]
    assert abs(float(la) - float(lb)) <= 2e-6 * max(1.0, abs(float(lb))), (float(la), float(lb))
    for p, a, b in zip(plist, ga, gb):
        assert (a is None) == (b is None)
        if a is not None:
            grad_close(a, b, "parameter of shape %s" % (tuple(p.shape),), rel=1e-4)
    # per flow step: ActNormFn + InvConvFn + MixtureCouplingFn + the log-det add -> one Function
    assert na[1] < nb[1] and na[0] < nb[0], (na, nb)


@pytest.mark.parametrize("B,N,D,K,seed", [(64, 16, 4, 8, 1), (33, 17, 6, 4, 2), (20, 38, 6, 5, 4), (5, 64, 3, 16, 7), (16, 9, 7, 3, 8)])
def test_last_coupling_nll_fns_vs_oracle_autograd(B, N, D, K, seed):
    """The last coupling layer + the NLL assembly as one Function (mixture and affine) against autograd through the oracle's
    coupling -> O.nll_per_sample (set_modeling/task.py:96-118)."""
    from categoricalnf_amd import functional as Fn, ops
    gen = torch.Generator().manual_seed(seed)
    z = torch.randn(B, N, D, generator=gen)
    pad, length = _pad_len(B, N, seed | 1, gen)                   # a length always: the NLL divides by it
    mask = _mask("channel", D)
    ldj0, wn = torch.randn(B, generator=gen), torch.randn(B, generator=gen)
    # mixture
    nn_out = 0.5 * torch.randn(B, N, D * (2 + 3 * K), generator=gen)
    sf, msf = 0.2 * torch.randn(D, generator=gen), 0.2 * torch.randn(D, K, generator=gen)
    leaves = [_leaf(t) for t in (z, nn_out, sf, msf, ldj0)]
    zc, nc, sfc, msfc, lc = leaves
    z1, l1, _ = O.mixture_coupling(zc, nc, mask, K, sfc, msfc, channel_padding_mask=pad)
    nll_o = O.nll_per_sample(z1, lc + l1, length, pad)
    (nll_o * wn).sum().backward()
    dl = [_leaf(t, True) for t in (z, nn_out, sf, msf, ldj0)]
    nll, zo, lo = Fn.MixtureCouplingNllFn.apply(*dl, g(mask), g(pad), g(length), K, -1.0, 1.0, True, ops.LOGISTIC_SIGMA, ops.LOGISTIC_LOG_SIGMA)
    (nll * g(wn)).sum().backward()
    grad_close(nll, nll_o, "nll", rel=5e-5); grad_close(zo, z1, "z_out", rel=5e-5)
    for name, a, b in zip(("g_z", "g_nn", "g_sf", "g_msf", "g_ldj"), dl, leaves):
        grad_close(a.grad, b.grad, "mixture " + name, rel=5e-4)
    # affine
    nn2 = 0.5 * torch.randn(B, N, 2 * D, generator=gen)
    leaves = [_leaf(t) for t in (z, nn2, sf, ldj0)]
    zc, nc, sfc, lc = leaves
    z1, l1 = O.affine_coupling(zc, nc, mask, sfc, ldj=lc * 1.0)
    nll_o = O.nll_per_sample(z1, l1, length, pad)
    (nll_o * wn).sum().backward()
    dl = [_leaf(t, True) for t in (z, nn2, sf, ldj0)]
    nll, zo, lo = Fn.AffineCouplingNllFn.apply(*dl, g(mask), g(pad), g(length), ops.LOGISTIC_SIGMA, ops.LOGISTIC_LOG_SIGMA)
    (nll * g(wn)).sum().backward()
    grad_close(nll, nll_o, "nll", rel=5e-5); grad_close(lo, l1, "ldj_out", rel=5e-5)
    for name, a, b in zip(("g_z", "g_nn", "g_sf", "g_ldj"), dl, leaves):
        grad_close(a.grad, b.grad, "affine " + name, rel=5e-4)


@pytest.mark.parametrize("B,N,D,seed", [(300, 38, 6, 1), (64, 16, 4, 2), (33, 17, 3, 3), (129, 64, 8, 4), (2048, 64, 6, 5), (7, 5, 5, 6), (50, 9, 7, 7)])
def test_actnorm_backward_token_owner_and_flat_tile_kernels_agree(B, N, D, seed):
    """cnf_actnorm_bwd has two kernels: token-owner wave tiles with register sums (default for D in {1..6, 8}) and flat tiles with
    lane-private LDS sums (cnf_set_actnorm_bwd_tiles(0); any D).  Same gradients to the rounding of the summation order, both
    directions, with padding and lengths; each bit-reproducible."""
    from categoricalnf_amd import functional as Fn
    lib = _lib.load()
    gen = torch.Generator().manual_seed(seed)
    z = g(torch.randn(B, N, D, generator=gen))
    bias, scales = g(torch.randn(1, 1, D, generator=gen)), g(0.3 * torch.randn(1, 1, D, generator=gen))
    pad, length = _pad_len(B, N, seed, gen)
    gz, gl = g(torch.randn(B, N, D, generator=gen)), g(torch.randn(B, generator=gen))
    for reverse in (False, True):
        out = {}
        for tiles in (1, 0):
            lib.cnf_set_actnorm_bwd_tiles(tiles)
            try:
                runs = []
                for _ in range(2):
                    zl, bl, sl, ll = (t.clone().requires_grad_(True) for t in (z, bias, scales, gl * 0))
                    zo, lo = Fn.ActNormFn.apply(zl, bl, sl, ll, g(length), g(pad), reverse)
                    torch.autograd.backward([zo, lo], [gz, gl])
                    runs.append((zl.grad, bl.grad, sl.grad))
                assert all(torch.equal(a, b) for a, b in zip(*runs))
                out[tiles] = runs[0]
            finally:
                lib.cnf_set_actnorm_bwd_tiles(1)
        for name, a, b in zip(("g_z", "g_bias", "g_scales"), out[1], out[0]):
            grad_close(a, b, name + (" reverse" if reverse else ""), rel=2e-5)


@pytest.mark.parametrize("B,N,D,seed", [(300, 38, 6, 1), (64, 16, 4, 2), (33, 17, 3, 3), (129, 64, 8, 4), (2048, 64, 6, 5), (50, 9, 2, 6), (1, 5, 6, 7)])
@pytest.mark.parametrize("mode", [1, 0])
def test_affine_backward_token_owner_and_flat_tile_kernels_agree(B, N, D, seed, mode):
    """cnf_affine_coupling_bwd has two kernels for channel masks: token-owner wave tiles (constants and scaling-factor sums in
    registers; D in {2, 3, 4, 6, 8}, the default without a scaling factor, forced here with mode 2) and flat tiles
    (cnf_set_affine_bwd_tiles(0); any mask, any D).  Same gradients to
    rounding in both directions, both math modes, with and without the scaling factor; each bit-reproducible."""
    from categoricalnf_amd import functional as Fn
    lib = _lib.load()
    gen = torch.Generator().manual_seed(seed)
    z, nn_out = g(torch.randn(B, N, D, generator=gen)), g(0.5 * torch.randn(B, N, 2 * D, generator=gen))
    sf, mask = g(0.3 * torch.randn(D, generator=gen)), g(_mask("channel", D))
    gz, gl = g(torch.randn(B, N, D, generator=gen)), g(torch.randn(B, generator=gen))
    lib.cnf_set_math_mode(mode)
    try:
        for reverse in (False, True):
            for with_sf in (True, False):
                out = {}
                for tiles in (2, 0):
                    lib.cnf_set_affine_bwd_tiles(tiles)
                    runs = []
                    for _ in range(2):
                        zl, nl, ll = (t.clone().requires_grad_(True) for t in (z, nn_out, gl * 0))
                        sl = sf.clone().requires_grad_(True) if with_sf else None
                        zo, lo = Fn.AffineCouplingFn.apply(zl, nl, sl, ll, mask, reverse)
                        torch.autograd.backward([zo, lo], [gz, gl])
                        runs.append((zl.grad, nl.grad) + ((sl.grad,) if with_sf else ()))
                    assert all(torch.equal(a, b) for a, b in zip(*runs))
                    out[tiles] = runs[0]
                for name, a, b in zip(("g_z", "g_nn", "g_sf"), out[2], out[0]):
                    grad_close(a, b, "%s reverse=%s sf=%s" % (name, reverse, with_sf), rel=2e-5)
    finally:
        lib.cnf_set_affine_bwd_tiles(1)
        lib.cnf_set_math_mode(1)


@pytest.mark.parametrize("D", [1, 2, 3, 4, 6, 8, 13, 16])
def test_lu_weight_kernel_matches_the_tensor_expression(D):
    """cnf_invconv_lu_weight / _bwd (InvertibleConv's W = P L U and sum log_s in one launch each way) against the reference's
    chain of tensor ops (permutation_layers.py:61-71) and its autograd: values to 1e-6, gradients of l, u, log_s to 1e-5 of scale;
    entries outside the strict triangles get exactly zero gradient."""
    from categoricalnf_amd import ops
    from categoricalnf_amd.layers.flows.permutation_layers import InvertibleConv
    torch.manual_seed(D)
    np.random.seed(D)
    layer = InvertibleConv(D).cuda().train()
    with torch.no_grad():                         # move away from the initialisation (l, u exactly triangular, an orthogonal W)
        for prm in (layer.l, layer.u, layer.log_s):
            prm.add_(0.1 * torch.randn_like(prm))
    gw, gs = torch.randn(D, D, device="cuda"), torch.randn((), device="cuda")
    res = {}
    for fused in (True, False):
        ops.FUSE_LU_WEIGHT = fused
        try:
            for prm in layer.parameters():
                prm.grad = None
            w, sldj = layer._build_weight()
            torch.autograd.backward([w, sldj], [gw, gs])
            res[fused] = (w.detach().clone(), sldj.detach().clone(), layer.l.grad.clone(), layer.u.grad.clone(), layer.log_s.grad.clone())
        finally:
            ops.FUSE_LU_WEIGHT = True
    a, b = res[True], res[False]
    assert tuple(a[0].shape) == (D, D) and a[1].dim() == 0
    for name, x, y in zip(("weight", "sldj", "g_l", "g_u", "g_log_s"), a, b):
        grad_close(x, y, name, rel=1e-5)
    strict_lower = torch.tril(torch.ones(D, D, device="cuda"), -1).bool()
    assert float(a[2][~strict_lower].abs().sum()) == 0.0 and float(a[3][~strict_lower.t()].abs().sum()) == 0.0
    if D > 12:              # the layer's backward kernel is built for D <= 12 (cnf_invconv_bwd)
        return
    # and through the layer: same outputs and parameter gradients of a training call
    x = torch.randn(7, 5, D, device="cuda")
    outs = {}
    for fused in (True, False):
        ops.FUSE_LU_WEIGHT = fused
        try:
            for prm in layer.parameters():
                prm.grad = None
            z, ldj = layer(x, ldj=torch.zeros(7, device="cuda"))
            (z.sum() + ldj.sum()).backward()
            outs[fused] = (z.detach(), ldj.detach(), layer.l.grad.clone(), layer.u.grad.clone(), layer.log_s.grad.clone())
        finally:
            ops.FUSE_LU_WEIGHT = True
    for name, x1, y1 in zip(("z", "ldj", "g_l", "g_u", "g_log_s"), outs[True], outs[False]):
        grad_close(x1, y1, "layer " + name, rel=2e-5)


def test_graph_colouring_flow_training_pass_with_and_without_the_fused_groups(tmp_path):
    """The same comparison on the graph-colouring flow (RGCN attention sub-network, graphs of different sizes: padding masks and
    lengths reach every fused group; the mixture couplings carry the reference's regulariser settings; the flow ends in an ActNorm,
    so the NLL is the prior kernel's): loss and parameter gradients with ops.FUSE_TRAINING on / off."""
    import contextlib, io
    from categoricalnf_amd import ops
    from categoricalnf_amd.experiments import run_graph_coloring as G
    from categoricalnf_amd.experiments.graph_coloring import GraphNodeFlow
    from categoricalnf_amd.experiments.graph_coloring_data import GraphColoringDataset, generate_planted_dataset
    dev = torch.device("cuda", 0)
    root = str(tmp_path)
    GraphColoringDataset.set_dataset(prefix="_tiny", num_colors=3)
    GraphColoringDataset.DATASET_NODES = GraphColoringDataset.DATASET_VAL_IDX = None
    with contextlib.redirect_stdout(io.StringIO()):
        generate_planted_dataset(root, prefix="_tiny", num_colors=3, num_graphs=400, n_min=10, n_max=20, seed=0)
        train = GraphColoringDataset(num_colors=3, train=True, data_root=root)
    args = G.parse(["--dataset", "tiny_3", "--coupling_hidden_size", "32", "--coupling_hidden_layers", "2", "--coupling_num_flows", "3"])
    torch.manual_seed(0)
    with contextlib.redirect_stdout(io.StringIO()):
        model = GraphNodeFlow(G.model_params(args), GraphColoringDataset).to(dev)
    rng = np.random.RandomState(0)

    def full(idx):
        items = [train[i] for i in idx]
        return (torch.from_numpy(np.stack([it[0] for it in items])).to(dev), torch.from_numpy(np.stack([it[1] for it in items])).to(dev),
                torch.from_numpy(np.array([it[2] for it in items], dtype=np.int64)).to(dev))
    init = []
    for _ in range(3):
        x, adj, ln = full(rng.randint(0, len(train), size=48))
        init.append((x, {"length": ln, "adjacency": adj}))
    with contextlib.redirect_stdout(io.StringIO()):
        model.initialize_data_dependent(init)
    model.train()
    x, adj, ln = full(rng.randint(0, len(train), size=48))
    assert int(ln.min()) < int(ln.max())                     # graphs of different sizes: padding is exercised
    noise = torch.rand(x.numel(), 1, model.embed_dim, device=dev)
    plist = [p for p in model.parameters() if p.requires_grad]
    out = {}
    for fused in (True, False):
        ops.FUSE_TRAINING = fused
        try:
            loss = model(x, adj, reverse=False, beta=1.3, length=ln, noise=noise, _nll=model.nll_request(length=ln))[2].mean()
            out[fused] = (loss.detach().clone(), torch.autograd.grad(loss, plist, allow_unused=True))
        finally:
            ops.FUSE_TRAINING = True
    (la, ga), (lb, gb) = out[True], out[False]
    assert abs(float(la) - float(lb)) <= 5e-6 * max(1.0, abs(float(lb))), (float(la), float(lb))
    for p, a, b in zip(plist, ga, gb):
        assert (a is None) == (b is None)
        if a is not None:
            grad_close(a, b, "parameter of shape %s" % (tuple(p.shape),), rel=2e-4)
    # and with beta in a device scalar (the captured step's form): the same loss and gradients as with the number
    loss_t = model(x, adj, reverse=False, beta=torch.tensor(1.3, device=dev), length=ln, noise=noise, _nll=model.nll_request(length=ln))[2].mean()
    gt = torch.autograd.grad(loss_t, plist, allow_unused=True)
    assert abs(float(loss_t.detach()) - float(la)) <= 5e-6 * max(1.0, abs(float(la)))
    for p, a, b in zip(plist, gt, ga):
        if a is not None:
            grad_close(a, b, "device-scalar beta, parameter of shape %s" % (tuple(p.shape),), rel=2e-4)


@pytest.mark.parametrize("B,N,D,K", [(64, 16, 4, 8), (33, 17, 6, 4), (20, 38, 6, 16), (8, 30, 3, 51), (256, 16, 4, 8), (16, 64, 2, 5)])
def test_mixture_backward_kernel_variants_agree(B, N, D, K):
    """cnf_mixture_coupling_bwd_f32 has several streaming kernels (cnf_set_mixture_bwd_waves): the rolled run-time-K kernel with
    1 / 2 / 4 lanes per item (the default picks among them by the amount of work) and the unrolled register-slot kernels of
    K = 4 / 8 / 16 (modes 0 / 1).  Every one gives the default's gradients to the rounding of the summation order; the default twice the same bits."""
    from categoricalnf_amd import functional as Fn
    lib = _lib.load()
    gen = torch.Generator().manual_seed(B + K)
    z = g(torch.randn(B, N, D, generator=gen))
    nn_out = g(0.5 * torch.randn(B, N, D * (2 + 3 * K), generator=gen))
    sf, msf = g(0.2 * torch.randn(D, generator=gen)), g(0.2 * torch.randn(D, K, generator=gen))
    mask = g(_mask("channel", D))
    ln = torch.randint(1, N + 1, (B,), generator=gen); ln[0] = N
    pad = g(O.length_mask(ln, N))
    gz, gl = g(torch.randn(B, N, D, generator=gen)), g(torch.randn(B, generator=gen))

    def run():
        leaves = [t.clone().requires_grad_(True) for t in (z, nn_out, sf, msf)]
        zo, lo, _ = Fn.MixtureCouplingFn.apply(leaves[0], leaves[1], leaves[2], leaves[3], None, mask, pad, K, -1.0, 1.0, True, True, True)
        torch.autograd.backward([zo, lo], [gz, gl])
        return [t.grad for t in leaves]
    ref, again = run(), run()
    assert all(torch.equal(x, y) for x, y in zip(ref, again))
    try:
        for mode in (0, 1, 2, 3, 4, 6):
            # (a forced variant whose stage does not fit LDS falls through to the fp64 kernel, whose sums are atomics: values only)
            lib.cnf_set_mixture_bwd_waves(mode)
            a = run()
            for name, x, y in zip(("g_z", "g_nn", "g_sf", "g_msf"), a, ref):
                grad_close(x, y, "%s mode %d" % (name, mode), rel=2e-5)
    finally:
        lib.cnf_set_mixture_bwd_waves(-1)


@pytest.mark.parametrize("compact", [False, True])
@pytest.mark.parametrize("B,N,D,K,pad_last", [(64, 16, 4, 8, False), (33, 17, 6, 4, True), (40, 64, 6, 8, True), (20, 38, 6, 16, False),
                                              (24, 30, 3, 5, True), (16, 21, 2, 8, False), (12, 40, 8, 4, True)])
def test_mixture_backward_streaming_write_back_is_bit_identical(B, N, D, K, pad_last, compact):
    """Large launches of the fp32 mixture backward (one lane per item, > 128 MB of gradient rows) write g_nn with nontemporal stores
    — the reference layout's tokens in address order, zero blocks included, by one loop (cnf_mixture_tok_bwd.hip) — and stage with
    nontemporal loads.  cnf_set_mixture_bwd_big_mb(0) runs that path on small shapes: every gradient, the zero blocks of the
    untransformed channels included, has the bits of the ordinary write-back (same kernel, lanes per item forced to one)."""
    from categoricalnf_amd import _lib, ops
    from categoricalnf_amd.ops import _ptr as P_
    lib = _lib.load()
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(B * 7 + K)
    Pn = 2 + 3 * K
    z = g(torch.randn(B, N, D, generator=gen))
    chess = D == 3                      # (one shape with a chess mask: every channel transformed somewhere, passes are whole tokens)
    if chess and compact:
        pytest.skip("the compact layout is for channel masks")
    mask = g(_mask("chess" if chess else "channel", D))
    m, mr, mc = ops._mask_desc(mask, D, dev)
    act, n_act = ops._act_list(mask, m, mr, mc, D)
    DA = D - D // 2
    width = (DA if compact else D) * Pn
    if compact and (B * N * width) % 4 != 0:
        pytest.skip("compact layout needs B N DA P to be a multiple of 4")
    nn_out = g(0.5 * torch.randn(B, N, width, generator=gen))
    sf, msf = g(0.2 * torch.randn(D, generator=gen)), g(0.2 * torch.randn(D, K, generator=gen))
    ln = torch.randint(1, N + 1, (B,), generator=gen); ln[0] = N
    pad = g(O.length_mask(ln, N)) if pad_last else None
    gz, gl = g(torch.randn(B, N, D, generator=gen)), g(torch.randn(B, generator=gen))
    ws = torch.empty(int(lib.cnf_bwd_workspace_floats(D + D * K)), device=dev)
    fn = lib.cnf_mixture_coupling_compact_bwd_f32 if compact else lib.cnf_mixture_coupling_bwd_f32

    def run():
        g_z = torch.full_like(z, float("nan"))
        g_nn = torch.full_like(nn_out, float("nan"))
        g_sf, g_msf = torch.empty_like(sf), torch.empty_like(msf)
        rc = fn(P_(z), P_(nn_out), P_(sf), P_(msf), P_(m), mr, mc, act, n_act, P_(pad) if pad is not None else None, 1, 1, P_(gz), P_(gl),
                P_(g_z), P_(g_nn), P_(g_sf), P_(g_msf), P_(ws), B, N, D, K, -1.0, 1.0, 1, ops._stream(dev))
        if compact and rc == _lib.CNF_ERR_UNSUPPORTED:
            pytest.skip("the compact-layout backward declines this shape (its stage does not fit with one lane per item)")
        assert rc == 0, lib.cnf_last_error()
        torch.cuda.synchronize()
        return g_z, g_nn, g_sf, g_msf
    try:
        lib.cnf_set_mixture_bwd_waves(2)          # one lane per item, the rolled kernel
        plain = run()
        lib.cnf_set_mixture_bwd_big_mb(0)
        streamed = run()
    finally:
        lib.cnf_set_mixture_bwd_big_mb(-1)
        lib.cnf_set_mixture_bwd_waves(-1)
    assert not torch.isnan(plain[1]).any()
    for name, x, y in zip(("g_z", "g_nn", "g_sf", "g_msf"), streamed, plain):
        assert torch.equal(x, y), name


@pytest.mark.parametrize("compact,K", [(True, 16), (False, 16), (True, 8), (False, 8)])
def test_mixture_backward_large_launch_streams_by_default(compact, K):
    """At a size the streaming write-back takes by itself (B = 16384, N = 64, D = 4: 2.1 M transformed elements, 200-840 MB of gradient
    rows; K = 16 takes the unrolled kernel's streaming build, K = 8 the rolled one's) the default launch gives the gradients of the
    ordinary kernel (rolled, one / two lanes per item at K = 8 / 16, streaming switched off) — g_z and g_nn to fp32 rounding of the same formulas, the two
    parameter gradients to the rounding of their summation order."""
    from categoricalnf_amd import _lib, ops
    from categoricalnf_amd.ops import _ptr as P_
    lib = _lib.load()
    dev = torch.device("cuda:0")
    B, N, D = 16384, 64, 4
    gen = torch.Generator(device=dev).manual_seed(K)
    Pn = 2 + 3 * K
    DA = D - D // 2
    z = torch.randn(B, N, D, generator=gen, device=dev)
    mask = g(_mask("channel", D))
    m, mr, mc = ops._mask_desc(mask, D, dev)
    act, n_act = ops._act_list(mask, m, mr, mc, D)
    nn_out = 0.5 * torch.randn(B, N, (DA if compact else D) * Pn, generator=gen, device=dev)
    sf, msf = 0.2 * torch.randn(D, generator=gen, device=dev), 0.2 * torch.randn(D, K, generator=gen, device=dev)
    gz, gl = torch.randn(B, N, D, generator=gen, device=dev), torch.randn(B, generator=gen, device=dev)
    ws = torch.empty(int(lib.cnf_bwd_workspace_floats(D + D * K)), device=dev)
    fn = lib.cnf_mixture_coupling_compact_bwd_f32 if compact else lib.cnf_mixture_coupling_bwd_f32

    def run():
        g_z = torch.full_like(z, float("nan"))
        g_nn = torch.full_like(nn_out, float("nan"))
        g_sf, g_msf = torch.empty_like(sf), torch.empty_like(msf)
        rc = fn(P_(z), P_(nn_out), P_(sf), P_(msf), P_(m), mr, mc, act, n_act, None, 1, 1, P_(gz), P_(gl),
                P_(g_z), P_(g_nn), P_(g_sf), P_(g_msf), P_(ws), B, N, D, K, -1.0, 1.0, 1, ops._stream(dev))
        assert rc == 0, lib.cnf_last_error()
        torch.cuda.synchronize()
        return g_z, g_nn, g_sf, g_msf
    streamed = run()
    try:
        lib.cnf_set_mixture_bwd_waves(2 if K == 8 else 3)      # (K = 16: one lane per item does not fit LDS with the rolled kernel's sums)
        lib.cnf_set_mixture_bwd_big_mb(1 << 30)
        plain = run()
    finally:
        lib.cnf_set_mixture_bwd_big_mb(-1)
        lib.cnf_set_mixture_bwd_waves(-1)
    assert not torch.isnan(streamed[1]).any() and not torch.isnan(streamed[0]).any()
    if not compact:
        # the untransformed channels' blocks: exact zeros, written by the address-ordered loop
        blocks = streamed[1].view(B, N, D, Pn)
        assert int(torch.count_nonzero(blocks[:, :, : D // 2])) == 0
    if K == 8:
        assert torch.equal(streamed[0], plain[0]) and torch.equal(streamed[1], plain[1])         # the same kernel, another write-back
    else:
        for name, x, y in zip(("g_z", "g_nn"), streamed[:2], plain[:2]):
            err = (x - y).abs().max().item()
            assert err <= 2e-5 * max(1.0, y.abs().max().item()), (name, err)
    for name, x, y in zip(("g_sf", "g_msf"), streamed[2:], plain[2:]):
        grad_close(x, y, name, rel=2e-4)


@pytest.mark.parametrize("D", [1, 2, 3, 4, 6, 8])
def test_lu_weight_assembly_hands_out_the_inverse_the_fused_backward_needs(D):
    """cnf_invconv_lu_weight_inv: W and sum log_s of cnf_invconv_lu_weight bit for bit, and W^-1 = the bits of the inverse launch
    inside cnf_actnorm_invconv_bwd(saved_is_output = 1, weight_inv = NULL) — so the fused pair's backward gives torch.equal
    gradients whether it inverts W itself or takes the inverse from the weight assembly (permutation_layers.py:61-76)."""
    from categoricalnf_amd import _lib, functional as Fn
    from categoricalnf_amd.ops import _ptr as P, _stream
    from categoricalnf_amd.layers.flows.permutation_layers import InvertibleConv
    lib = _lib.load()
    dev = torch.device("cuda:0")
    st = _stream(dev)
    torch.manual_seed(D)
    np.random.seed(D)
    layer = InvertibleConv(D).cuda().train()
    with torch.no_grad():
        for prm in (layer.l, layer.u, layer.log_s):
            prm.add_(0.2 * torch.randn_like(prm))
    w, sldj = layer._build_weight()
    inv = Fn._known_inverse(w)
    assert inv is not None and tuple(inv.shape) == (D, D)
    w0, s0 = torch.empty(D, D, device=dev), torch.empty(1, device=dev)
    assert lib.cnf_invconv_lu_weight(P(layer.p), P(layer.l), P(layer.u), P(layer.log_s), P(layer.sign_s), P(w0), P(s0), D, st) == 0
    assert torch.equal(w0, w.detach()) and torch.equal(s0.reshape(()), sldj.detach())
    ref = torch.inverse(w.detach().double())
    assert float((inv.double() - ref).abs().max()) <= 1e-5 * float(ref.abs().max())
    # the fused pair's backward from the saved OUTPUT: own inverse launch vs the handed-over inverse
    B, N = 33, 20
    z_out, gz, gl = torch.randn(B, N, D, device=dev), torch.randn(B, N, D, device=dev), torch.randn(B, device=dev)
    bias, scales = torch.randn(D, device=dev), 0.1 * torch.randn(D, device=dev)
    outs = []
    for wi in (None, inv):
        hold = Fn._Hold()
        outs.append(Fn._actconv_bwd(z_out, True, bias, scales, w.detach(), None, None, gz, gl, hold, weight_inv=wi))
    torch.cuda.synchronize()
    for a, b in zip(*outs):
        assert torch.isfinite(a).all() and torch.equal(a, b)


@pytest.mark.parametrize("spread", [1.0, 4.0])
def test_fused_pair_backward_from_the_output_with_an_ill_conditioned_weight(spread):
    """the fused training groups keep only the pair's OUTPUT and rebuild its input through W^-1 (fp64 inverse, fp32
    latents), so the error of g_weight / g_scales grows with cond(W).  log_s of the LU parametrisation spread over +-`spread`
    (cond(W) up to e^8 ~ 3000 at 4): the from-output gradients stay within 2e-3 of the from-input ones relative to each
    tensor's largest entry — the documented tolerance of CNF_FUSE_TRAINING (functional.MixtureActConvFn / EncoderActConvFn)."""
    from categoricalnf_amd import functional as Fn, ops
    from categoricalnf_amd.layers.flows.permutation_layers import InvertibleConv
    dev = torch.device("cuda:0")
    D, B, N = 6, 256, 32
    torch.manual_seed(3)
    np.random.seed(3)
    conv = InvertibleConv(D).cuda().train()
    with torch.no_grad():
        conv.log_s.copy_(torch.linspace(-spread, spread, D, device=dev))
        conv.l.add_(0.3 * torch.randn_like(conv.l))
        conv.u.add_(0.3 * torch.randn_like(conv.u))
    w, sldj = conv._build_weight()
    w = w.detach()
    cond = float(torch.linalg.cond(w.double()))
    bias, scales = torch.randn(D, device=dev), 0.3 * torch.randn(D, device=dev)
    z = torch.randn(B, N, D, device=dev)
    z_out, _ = ops.actnorm_invconv(z, bias, scales, w, sldj.detach())
    gz, gl = torch.randn(B, N, D, device=dev), torch.randn(B, device=dev)
    from_in = Fn._actconv_bwd(z, False, bias, scales, w, None, None, gz, gl, Fn._Hold())
    from_out = Fn._actconv_bwd(z_out, True, bias, scales, w, None, None, gz, gl, Fn._Hold(), weight_inv=Fn._known_inverse(conv._build_weight()[0]))
    torch.cuda.synchronize()
    for name, a, b in zip(("g_z", "g_bias", "g_scales", "g_weight", "g_sldj"), from_out, from_in):
        scale = float(b.abs().max()) + 1e-12
        err = float((a - b).abs().max()) / scale
        assert err <= 2e-3, (name, err, cond)
