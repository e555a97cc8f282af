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
