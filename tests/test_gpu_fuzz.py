"""Seeded shape fuzzing of the HIP kernels against the CPU oracle: every kernel family on a few dozen random
(B, N, D, K, C, mask, padding) combinations — odd sizes, single rows, rows shorter than a wave, masks of every kind,
ragged lengths — in both arithmetic modes where the mode changes the kernel.  Shapes are drawn from a fixed seed, so
the list is the same on every run; sizes are kept small so that the oracle finishes in seconds."""
import numpy as np
import pytest
import torch

from categoricalnf_amd import _lib
from oracle import cnf_oracle as O

pytestmark = pytest.mark.gpu
ELEM = dict(rtol=3e-5, atol=3e-5)


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


def _shapes(seed, n, max_b=70, max_n=40, dims=(1, 2, 3, 4, 5, 6, 8)):
    r = np.random.RandomState(seed)
    out = []
    for _ in range(n):
        B = int(r.choice([1, 2, 3, 7, 33, 64, 65, int(r.randint(1, max_b))]))
        N = int(r.choice([1, 2, 5, 16, 17, int(r.randint(1, max_n))]))
        D = int(r.choice(dims))
        out.append((B, N, D, int(r.randint(0, 1 << 30))))
    return out


def _mask_and_pad(kind, B, N, D, gen):
    if kind == "none":
        mask = None
    elif kind == "chess" or D == 1:          # a channel mask needs two channels (the reference switches to chess too)
        mask = O.chess_mask()
    else:
        mask = O.channel_mask(D)
    ln = torch.randint(1, N + 1, (B,), generator=gen)
    ln[0] = N
    return mask, ln, O.length_mask(ln, N)


@pytest.mark.parametrize("B,N,D,seed", _shapes(1, 28))
@pytest.mark.parametrize("mode", [1, 0])
def test_fuzz_affine(B, N, D, seed, mode):
    gen = torch.Generator().manual_seed(seed)
    z = 1.3 * torch.randn(B, N, D, generator=gen)
    nn_out = 0.8 * torch.randn(B, N, 2 * D, generator=gen)
    sf = 0.4 * torch.randn(D, generator=gen) if seed % 3 else None
    kind = ["channel", "chess"][seed % 2]
    mask, ln, pad = _mask_and_pad(kind, B, N, D, gen)
    ldj0 = torch.randn(B, generator=gen)
    lib = _lib.load()
    lib.cnf_set_math_mode(mode)
    try:
        zo, lo = O.affine_coupling(z, nn_out, mask, sf, ldj=ldj0.clone())
        zf, lf = ops().affine_coupling(g(z), g(nn_out), g(sf), g(mask), ldj=g(ldj0))
        close(zf, zo, **ELEM); loglik_close(lf, lo)
        zr, lr = ops().affine_coupling(zf, g(nn_out), g(sf), g(mask), reverse=True, ldj=lf)
        close(zr, z, rtol=1e-4, atol=1e-4); loglik_close(lr, ldj0)
        # fused NLL epilogue with ragged lengths
        zn, lnl, neglog, nll = ops().affine_coupling_nll(g(z), g(nn_out), g(sf), g(mask), ldj=g(ldj0), length=g(ln),
                                                         channel_padding_mask=g(pad))
        assert torch.equal(zn, zf) and torch.equal(lnl, lf)
        loglik_close(nll, O.nll_per_sample(zo, lo, ln.float(), pad))
    finally:
        lib.cnf_set_math_mode(1)


@pytest.mark.parametrize("B,N,D,seed", _shapes(2, 26, max_b=40, max_n=24, dims=(1, 2, 3, 4, 6)))
@pytest.mark.parametrize("mode", [1, 0])
def test_fuzz_mixture(B, N, D, seed, mode):
    gen = torch.Generator().manual_seed(seed)
    K = int([1, 2, 4, 5, 8, 9, 16, 23][seed % 8])
    z = 1.5 * torch.randn(B, N, D, generator=gen)
    nn_out = 0.6 * torch.randn(B, N, D * (2 + 3 * K), generator=gen)
    sf = 0.2 * torch.randn(D, generator=gen) if seed % 4 else None
    msf = 0.2 * torch.randn(D, K, generator=gen) if seed % 5 else None
    kind = ["channel", "chess", "none"][seed % 3]
    mask, ln, pad = _mask_and_pad(kind, B, N, D, gen)
    pad_arg = pad if kind != "none" else None
    kw = dict(num_mixtures=K, reg_max=3.5, reg_factor=2.0, is_training=bool(seed % 2))
    lib = _lib.load()
    lib.cnf_set_math_mode(mode)
    try:
        zo, lo, ro = O.mixture_coupling(z, nn_out, mask, scaling_factor=sf, mixture_scaling_factor=msf,
                                        channel_padding_mask=pad_arg, **kw)
        zf, lf, rf = ops().mixture_coupling(g(z), g(nn_out), g(mask), scaling_factor=g(sf), mixture_scaling_factor=g(msf),
                                            channel_padding_mask=g(pad_arg), **kw)
        close(zf, zo, **ELEM); loglik_close(lf, lo)
        if ro is not None and rf is not None:
            loglik_close(rf, ro)
        zo2, lo2, _ = O.mixture_coupling(zo, nn_out, mask, scaling_factor=sf, mixture_scaling_factor=msf,
                                         channel_padding_mask=pad_arg, reverse=True, **kw)
        zr, lr, _ = ops().mixture_coupling(g(zo), g(nn_out), g(mask), scaling_factor=g(sf), mixture_scaling_factor=g(msf),
                                           channel_padding_mask=g(pad_arg), reverse=True, **kw)
        close(zr, zo2, rtol=1e-4, atol=1e-4); loglik_close(lr, lo2)
    finally:
        lib.cnf_set_math_mode(1)


@pytest.mark.parametrize("B,N,D,seed", _shapes(3, 24))
def test_fuzz_actnorm_invconv_prior(B, N, D, seed):
    gen = torch.Generator().manual_seed(seed)
    z = torch.randn(B, N, D, generator=gen)
    bias, scales = torch.randn(1, 1, D, generator=gen), 0.3 * torch.randn(1, 1, D, generator=gen)
    w = torch.linalg.qr(torch.randn(D, D, generator=gen))[0] * (1.0 + 0.1 * torch.randn(1, generator=gen))
    sldj = torch.slogdet(w)[1]
    _, ln, pad = _mask_and_pad("chess", B, N, D, gen)
    variant = seed % 3                      # 0: nothing, 1: length, 2: length + padding mask
    kw = {} if variant == 0 else ({"length": ln} if variant == 1 else {"length": ln, "channel_padding_mask": pad})
    gkw = {k: g(v) for k, v in kw.items()}
    zin = z * pad if variant == 2 else z
    za, la = O.actnorm(zin, bias, scales, **kw)
    zc, lc = O.invconv(za, w, sldj, ldj=la, **kw)
    a1, l1 = ops().actnorm(g(zin), g(bias), g(scales), **gkw)
    c1, l2 = ops().invconv(a1, g(w), g(sldj), ldj=l1, **gkw)
    close(a1, za, **ELEM); close(c1, zc, **ELEM); loglik_close(l2, lc)
    if D in ops().FUSED_ACTCONV_DIMS:
        f1, lf = ops().actnorm_invconv(g(zin), g(bias), g(scales), g(w), g(sldj), **gkw)
        assert torch.equal(f1, c1) and torch.equal(lf, l2)
    # inverse of both
    winv = torch.inverse(w.double()).float()
    xr, lr = ops().invconv(c1, g(winv), g(sldj), reverse=True, ldj=l2, **gkw)
    zr, lr = ops().actnorm(xr, g(bias), g(scales), reverse=True, ldj=lr, **gkw)
    close(zr, zin, rtol=1e-4, atol=1e-4)
    # the pair's log-det terms are added and taken away again: (0 + a + b) - b - a is zero up to the rounding of its
    # four fp32 additions, i.e. a few ulp of the largest intermediate — that bound, not a flat 1e-3
    ulp = 1.1920929e-07 * torch.maximum(l1.abs(), l2.abs()).cpu()
    assert bool((lr.cpu().abs() <= 4.0 * ulp).all()), (lr.cpu().abs() / ulp.clamp(min=1e-30)).max().item()
    # prior log-prob, NLL, batch sum
    sums = torch.zeros(2, dtype=torch.float64, device="cuda")
    neglog, nll = ops().prior_nll(c1, l2, g(ln), g(pad), sums=sums)
    loglik_close(nll, O.nll_per_sample(zc, lc, ln.float(), pad))
    assert abs(sums[0].item() - nll.double().sum().item()) < 1e-6 * max(1.0, abs(sums[0].item())) and sums[1].item() == B
    close(ops().logistic_log_prob(c1), O.logistic_log_prob(zc), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("B,N,D,seed", _shapes(4, 24, max_b=50, max_n=30, dims=(1, 2, 3, 4, 6, 8)))
def test_fuzz_encoder(B, N, D, seed):
    gen = torch.Generator().manual_seed(seed)
    C = int([2, 3, 5, 16, 27, 51][seed % 6])
    categ = torch.randint(0, C, (B, N), generator=gen)
    table = torch.cat([2.0 * torch.randn(C, D, generator=gen), 0.5 * torch.randn(C, D, generator=gen)], dim=1)
    prior = torch.log_softmax(torch.randn(C, generator=gen), 0)
    u = torch.rand(B * N, 1, D, generator=gen)
    _, ln, pad = _mask_and_pad("chess", B, N, D, gen)
    pad_arg = pad if seed % 2 else None
    beta = [1.0, 0.3][seed % 2]
    eps_o = O.logistic_from_uniform(u)
    eps_g = ops().logistic_from_uniform(g(u))
    close(eps_g, eps_o, rtol=3e-6, atol=3e-6)
    zo, lo, cpo = O.encoder_forward(categ, eps_o, table, prior, beta=beta, channel_padding_mask=pad_arg)
    zg, lg, cpg = ops().encoder_forward(g(categ), g(eps_o), g(table), g(prior), beta=beta, channel_padding_mask=g(pad_arg),
                                        want_class_prob=True)
    close(zg, zo, **ELEM); loglik_close(lg, lo); close(cpg, cpo.reshape(-1), rtol=1e-4, atol=1e-4)
    do, _ = O.encoder_decode(zo, table, prior)
    dg = ops().encoder_decode(g(zo), g(table), g(prior))
    same = (dg.cpu() == do)
    if not bool(same.all()):
        # a differing index is only acceptable on a numerical tie of the two class scores (never seen so far)
        raise AssertionError("decoded categories differ at %d of %d tokens" % (int((~same).sum()), same.numel()))


@pytest.mark.parametrize("B,N,D,seed", _shapes(5, 20, dims=(1, 2, 3, 4, 6, 8)) + [(64, 64, 6, 11), (33, 16, 3, 12), (9, 32, 8, 13), (7, 24, 2, 14)])
@pytest.mark.parametrize("mode", [1, 0])
def test_fuzz_ext_actnorm_and_sigmoid(B, N, D, seed, mode):
    """ExtActNorm (per-element kernel and the grouped-token kernel, which needs N % {1,2,4} == 0 and N/TP >= 8) and
    the sigmoid / logit flow, both directions, both arithmetic modes."""
    gen = torch.Generator().manual_seed(seed)
    z = torch.randn(B, N, D, generator=gen)
    cond = torch.cat([torch.randn(B, N, D, generator=gen), 0.7 * torch.randn(B, N, D, generator=gen)], dim=-1)
    _, ln, pad = _mask_and_pad("chess", B, N, D, gen)
    pad_arg = pad if seed % 2 else None
    ldj0 = torch.randn(B, generator=gen)
    lib = _lib.load()
    lib.cnf_set_math_mode(mode)
    try:
        zo, lo = O.ext_actnorm(z, cond, channel_padding_mask=pad_arg, ldj=ldj0.clone())
        zf, lf = ops().ext_actnorm(g(z), g(cond), channel_padding_mask=g(pad_arg), ldj=g(ldj0.clone()))
        close(zf, zo, **ELEM); loglik_close(lf, lo)
        zo2, lo2 = O.ext_actnorm(zo, cond, reverse=True, channel_padding_mask=pad_arg, ldj=lo.clone())
        zr, lr = ops().ext_actnorm(g(zo), g(cond), reverse=True, channel_padding_mask=g(pad_arg), ldj=g(lo.clone()))
        close(zr, zo2, rtol=1e-4, atol=1e-4); loglik_close(lr, lo2)
        x = torch.randn(B, N, 1, generator=gen) * 3.0
        so, slo = O.sigmoid_flow(x, ldj=ldj0.clone())
        sg, slg = ops().sigmoid_flow(g(x), ldj=g(ldj0.clone()))
        close(sg, so, rtol=1e-5, atol=1e-6); loglik_close(slg, slo)
        u = torch.rand(B, N, 1, generator=gen)
        ro, rlo = O.sigmoid_flow(u, reverse=True, ldj=ldj0.clone())
        rg, rlg = ops().sigmoid_flow(g(u), reverse=True, ldj=g(ldj0.clone()))
        close(rg, ro, rtol=1e-4, atol=1e-4); loglik_close(rlg, rlo)
    finally:
        lib.cnf_set_math_mode(1)


@pytest.mark.parametrize("B,N,D,seed", _shapes(6, 26, max_b=40, max_n=24, dims=(1, 2, 3, 4, 5, 6, 7, 8, 10, 13, 16)))
def test_fuzz_encoder_class_tiled(B, N, D, seed):
    """The class-tiled encoder kernels (forward, decode, backward) on random shapes: every latent width 1..16 (templated and
    generic paths), vocabularies from 2 to a few thousand classes around the chunk / split boundaries (1024, 1025, one
    class in the last split), fewer tokens than a workgroup, padding, beta != 1 — against the oracle and its autograd."""
    from categoricalnf_amd import functional as Fn
    gen = torch.Generator().manual_seed(seed)
    C = int([2, 17, 64, 300, 1023, 1024, 1025, 1537, 2049, 3100][seed % 10])
    if B * N * C > 400000:                       # keep the oracle's [T*C, 1, D] tensors small
        N = max(1, 400000 // (B * C))
    categ = torch.randint(0, C, (B, N), generator=gen)
    categ[0, 0], categ[-1, -1] = 0, C - 1
    table = torch.cat([1.5 * torch.randn(C, D, generator=gen), 0.5 * torch.randn(C, D, generator=gen)], dim=1)
    prior = torch.log_softmax(torch.randn(C, generator=gen), 0)
    eps = O.logistic_from_uniform(torch.rand(B * N, 1, D, generator=gen))
    _, ln, pad = _mask_and_pad("chess", B, N, D, gen)
    pad_arg = pad if seed % 2 else None
    beta = [1.0, 0.6][seed % 2]
    wz, wl = torch.randn(B, N, D, generator=gen), torch.randn(B, generator=gen)
    tc = table.clone().requires_grad_()
    zo, lo, cpo = O.encoder_forward(categ, eps, tc, prior, beta=beta, channel_padding_mask=pad_arg)
    ((zo * wz).sum() + (lo * wl).sum()).backward()
    tg = g(table).requires_grad_()
    zg, lg, cpg = Fn.EncoderForwardFn.apply(tg, g(categ), g(eps), g(prior), g(pad_arg), beta, True, True)
    close(zg, zo, **ELEM); close(lg, lo, rtol=1e-4, atol=1e-4 * max(1.0, float(lo.detach().abs().max())))
    close(cpg, cpo.reshape(-1), rtol=1e-4, atol=2e-4)
    ((zg * g(wz)).sum() + (lg * g(wl)).sum()).backward()
    scale = max(float(tc.grad.abs().max()), 1.0)
    close(tg.grad, tc.grad, rtol=3e-3, atol=3e-4 * scale)
    do, score = O.encoder_decode(zo.detach(), table, prior)
    dg = ops().encoder_decode(g(zo.detach()), g(table), g(prior), tiled=True).cpu()
    if not torch.equal(dg, do):
        # a different index is acceptable only on a tie of the two class scores at fp32 resolution
        s = score.reshape(-1, C)
        gap = (s.gather(1, do.reshape(-1, 1)) - s.gather(1, dg.reshape(-1, 1))).abs().max()
        assert float(gap) < 1e-4, "decoded categories differ beyond a numerical tie (score gap %g)" % float(gap)
    ops().check_flags(torch.device("cuda"), "tiled encoder fuzz")
