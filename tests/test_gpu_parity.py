"""GPU parity tests: the HIP path (through the C ABI, via categoricalnf_amd.ops / the layer modules)
against (a) the golden vectors captured from the real reference and (b) the CPU oracle on seeded
inputs, plus size-independent properties at the full benchmark shape.

Tolerances: fp32 kernels 2e-5 abs/rel per element (1 ulp-level libm differences between ocml and
the reference's SLEEF/MKL), per-sample log-det / log-likelihood 1e-4 relative (BASELINE.json
north_star), integer category indices bit-exact."""
import os

import numpy as np
import pytest
import torch
import torch.nn as nn

from oracle import cnf_oracle as O
from tests.golden_util import load_cases
from categoricalnf_amd import _lib

pytestmark = pytest.mark.gpu

ELEM = dict(rtol=2e-5, atol=2e-5)


def ops():
    from categoricalnf_amd import ops as _ops
    return _ops


def g(t):
    return t.cuda() if isinstance(t, torch.Tensor) else t


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


def test_library_loaded_is_hip():
    from categoricalnf_amd import _lib
    lib = _lib.load()
    assert lib.cnf_abi_version() == 1
    assert torch.cuda.is_available()
    assert "libcnf_hip.so" in open("/proc/self/maps").read()


# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("c", load_cases("affine_coupling"))
def test_affine_golden(c):
    zf, lf = ops().affine_coupling(g(c.z), g(c.nn_out), g(c.scaling_factor), g(c.mask), reverse=False, ldj=g(c.ldj_in))
    close(zf, c.z_fwd, **ELEM); loglik_close(lf, c.ldj_fwd)
    zr, lr = ops().affine_coupling(g(c.z_fwd), g(c.nn_out), g(c.scaling_factor), g(c.mask), reverse=True)
    close(zr, c.z_rev, **ELEM); loglik_close(lr, c.ldj_rev)
    s, t = ops().affine_params(g(c.nn_out), g(c.mask), g(c.scaling_factor))
    close(s, c.s, **ELEM); close(t, c.t, **ELEM)
    s, t = ops().affine_params(g(c.nn_out), g(c.mask), None)
    close(s, c.s_nofac, **ELEM); close(t, c.t_nofac, **ELEM)
    z2, l2 = ops().affine_transform(g(c.z), g(c.s), g(c.t), reverse=False)
    close(z2, c.z_fwd, **ELEM); loglik_close(l2, c.ldj_fwd - c.ldj_in)


@pytest.mark.parametrize("B,N,D,chess", [(37, 64, 6, False), (130, 16, 4, False), (9, 703, 2, False), (50, 21, 2, False),
                                         (33, 7, 3, False), (1000, 1, 4, False), (17, 9, 1, True), (5, 288, 3, False),
                                         (3, 2000, 6, False), (64, 38, 6, False)])
def test_affine_vs_oracle(B, N, D, chess):
    gen = torch.Generator().manual_seed(B * 1000 + N * 10 + D)
    z = torch.randn(B, N, D, generator=gen)
    nn_out = 0.7 * torch.randn(B, N, 2 * D, generator=gen)
    sf = 0.3 * torch.randn(D, generator=gen)
    mask = O.chess_mask() if chess else O.channel_mask(D)
    ldj0 = torch.randn(B, generator=gen)
    for use_sf in (True, False):
        zf_o, lf_o = O.affine_coupling(z, nn_out, mask, sf if use_sf else None, reverse=False, ldj=ldj0)
        zf, lf = ops().affine_coupling(g(z), g(nn_out), g(sf) if use_sf else None, g(mask), reverse=False, ldj=g(ldj0))
        close(zf, zf_o, **ELEM); loglik_close(lf, lf_o)
        zr_o, lr_o = O.affine_coupling(zf_o, nn_out, mask, sf if use_sf else None, reverse=True)
        zr, lr = ops().affine_coupling(g(zf_o), g(nn_out), g(sf) if use_sf else None, g(mask), reverse=True)
        close(zr, zr_o, **ELEM); loglik_close(lr, lr_o)


def test_affine_tiling_knobs_do_not_change_results():
    from categoricalnf_amd import _lib
    lib = _lib.load()
    gen = torch.Generator().manual_seed(5)
    z, nn_out = torch.randn(301, 64, 6, generator=gen), 0.5 * torch.randn(301, 64, 12, generator=gen)
    sf, mask = torch.zeros(6), O.channel_mask(6)
    ref = None
    try:
        lib.cnf_set_math_mode(0)
        for chunks in (64, 128, 192, 256, 512):
            for unroll in (0, 1, 2, 3, 4):
                lib.cnf_set_tile_chunks(chunks); lib.cnf_set_unroll(unroll)
                out = ops().affine_coupling(g(z), g(nn_out), g(sf), g(mask))
                if ref is None:
                    ref = out
                    zo, lo = O.affine_coupling(z, nn_out, mask, sf)
                    close(out[0], zo, **ELEM); loglik_close(out[1], lo)
                assert torch.equal(out[0], ref[0])            # element math independent of the tiling
                close(out[1], ref[1], rtol=1e-6, atol=1e-5)    # sums may associate differently
        # hardware-transcendental math mode: same results to ~1e-6
        lib.cnf_set_math_mode(1)
        fast = ops().affine_coupling(g(z), g(nn_out), g(sf), g(mask))
        close(fast[0], ref[0], rtol=5e-6, atol=5e-6); close(fast[1], ref[1], rtol=1e-5, atol=1e-4)
    finally:
        lib.cnf_set_tile_chunks(128); lib.cnf_set_unroll(2); lib.cnf_set_math_mode(1)


def test_affine_full_size_properties():
    """B=16384, N=64, D=6 (the north-star shape): inverse∘forward round trip and ldj antisymmetry."""
    B, N, D = 16384, 64, 6
    gen = torch.Generator(device="cuda").manual_seed(0)
    z = torch.randn(B, N, D, generator=gen, device="cuda")
    nn_out = 0.5 * torch.randn(B, N, 2 * D, generator=gen, device="cuda")
    sf = torch.zeros(D, device="cuda")
    mask = g(O.channel_mask(D))
    zf, lf = ops().affine_coupling(z, nn_out, sf, mask, reverse=False)
    zr, lr = ops().affine_coupling(zf, nn_out, sf, mask, reverse=True)
    assert torch.equal(lf, -lr)                                   # same sums, opposite sign: exact
    assert (zr - z).abs().max().item() <= 4e-6 * max(1.0, z.abs().max().item())
    assert torch.equal(zf[..., :3], z[..., :3])                   # masked channels untouched bit-for-bit
    # a slice against the oracle + linearity of the log-det in the batch (checksum of checksums)
    zo, lo = O.affine_coupling(z[:64].cpu(), nn_out[:64].cpu(), mask.cpu(), sf.cpu())
    close(zf[:64], zo, **ELEM); loglik_close(lf[:64], lo)
    _, l_half = ops().affine_coupling(z[:8192].contiguous(), nn_out[:8192].contiguous(), sf, mask)
    assert torch.equal(l_half, lf[:8192])


# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("c", load_cases("mixture_coupling"))
def test_mixture_golden(c):
    m = c.meta
    mask, pad = c.get("mask"), c.get("pad")
    ar = m["mask_kind"] == "none"
    kw = dict(num_mixtures=m["K"], scaling_factor=g(c.scaling_factor), mixture_scaling_factor=g(c.mixture_scaling_factor),
              channel_padding_mask=g(pad), reg_max=m["reg_max"], reg_factor=m["reg_factor"], is_training=m["training"])
    zf, lf, reg = ops().mixture_coupling(g(c.z), g(c.nn_out), g(mask), reverse=False, **kw)
    tail = m.get("tail", 1.0) > 1.0
    etol = dict(rtol=1e-4, atol=1e-4) if tail else ELEM
    close(zf, c.z_fwd, **etol); loglik_close(lf, c.ldj_fwd)
    if "reg_ldj" in c:
        loglik_close(reg, c.reg_ldj)
    if "z_rev" in c:
        zr, lr, _ = ops().mixture_coupling(g(c.z_fwd), g(c.get("nn_out_rev", c.nn_out)), g(mask), reverse=True, **kw)
        close(zr, c.z_rev, rtol=1e-4, atol=1e-4); loglik_close(lr, c.ldj_rev)
    if "p_t" in c:
        p = ops().mixture_params(g(c.nn_out), g(mask), m["K"], g(c.scaling_factor), g(c.mixture_scaling_factor))
        for got, key in zip(p, ["p_t", "p_log_s", "p_log_pi", "p_mixt_t", "p_mixt_log_s"]):
            assert got.dtype == torch.float64
            close(got, c[key], rtol=2e-6, atol=2e-6)
        # static run_with_params on the reference's own fp64 parameters
        mk = O.expand_mask(mask, c.z) if mask is not None else None
        z64, l64, reg_el = ops().mixture_transform(g(c.z.double()), *[g(c[k]) for k in ["p_t", "p_log_s", "p_log_pi", "p_mixt_t", "p_mixt_log_s"]],
                                                   reverse=False, reg_max=m["reg_max"], reg_factor=m["reg_factor"], mask=g(mk),
                                                   channel_padding_mask=g(pad) if mask is not None else None, is_training=m["training"])
        zo, lo, ro = O.mixture_transform(c.z.double(), c.p_t, c.p_log_s, c.p_log_pi, c.p_mixt_t, c.p_mixt_log_s, reverse=False,
                                         reg_max=m["reg_max"], reg_factor=m["reg_factor"], mask=mk,
                                         channel_padding_mask=pad if mask is not None else None, is_training=m["training"])
        close(z64, zo, rtol=1e-9, atol=1e-9); close(l64, lo, rtol=1e-9, atol=1e-9); close(reg_el, ro, rtol=1e-9, atol=1e-9)


@pytest.mark.parametrize("B,N,D,K,kind", [(40, 16, 4, 8, "channel"), (7, 50, 6, 16, "channel"), (3, 288, 3, 51, "none"),
                                          (9, 38, 6, 16, "channel"), (4, 703, 2, 8, "channel"), (11, 13, 1, 8, "chess"),
                                          (70, 5, 3, 4, "channel"), (300, 1, 2, 8, "channel")])
def test_mixture_vs_oracle(B, N, D, K, kind):
    gen = torch.Generator().manual_seed(B + 7 * N + 31 * D + K)
    z = 1.5 * torch.randn(B, N, D, generator=gen)
    nn_out = 0.6 * torch.randn(B, N, D * (2 + 3 * K), generator=gen)
    sf, msf = 0.2 * torch.randn(D, generator=gen), 0.2 * torch.randn(D, K, generator=gen)
    mask = None if kind == "none" else (O.chess_mask() if kind == "chess" else O.channel_mask(D))
    ln = torch.randint(max(1, N // 2), N + 1, (B,), generator=gen)
    pad = O.length_mask(ln, N) if kind != "none" else None
    kw = dict(num_mixtures=K, reg_max=3.5, reg_factor=2.0, is_training=True)
    zo, lo, ro = O.mixture_coupling(z, nn_out, mask, scaling_factor=sf, mixture_scaling_factor=msf, channel_padding_mask=pad, **kw)
    zf, lf, rf = ops().mixture_coupling(g(z), g(nn_out), g(mask), scaling_factor=g(sf), mixture_scaling_factor=g(msf),
                                        channel_padding_mask=g(pad), **kw)
    close(zf, zo, **ELEM); loglik_close(lf, lo); loglik_close(rf, ro)
    # inverse (K = 51 exceeds the LDS constant table: exercises the recompute path)
    zo2, lo2, _ = O.mixture_coupling(zo, nn_out, mask, scaling_factor=sf, mixture_scaling_factor=msf,
                                     channel_padding_mask=pad, reverse=True, **kw)
    zr, lr, _ = ops().mixture_coupling(g(zo), g(nn_out), g(mask), scaling_factor=g(sf), mixture_scaling_factor=g(msf),
                                       channel_padding_mask=g(pad), reverse=True, **kw)
    close(zr, zo2, rtol=1e-4, atol=1e-4); loglik_close(lr, lo2)
    # round trip on the transformed, un-padded entries
    keep = (pad if pad is not None else torch.ones(B, N, 1)).expand(-1, -1, D) > 0
    assert ((zr.cpu() - z)[keep]).abs().max() < 5e-4


@pytest.fixture
def exact_math():
    """math mode 0: the fp64 mixture kernel / libm affine kernel (default is 1 = fast)."""
    lib = _lib.load()
    lib.cnf_set_math_mode(0)
    yield
    lib.cnf_set_math_mode(1)


@pytest.mark.parametrize("B,N,D,K,kind", [(40, 16, 4, 8, "channel"), (3, 288, 3, 51, "none"), (11, 13, 1, 8, "chess"),
                                          (9, 38, 6, 16, "channel"), (70, 5, 3, 4, "channel"), (5, 20, 2, 10, "channel")])
def test_mixture_exact_mode_vs_oracle(exact_math, B, N, D, K, kind):
    """The fp64 forward kernel (math mode 0) against the oracle; the default-mode tests above run the fp32 kernel."""
    gen = torch.Generator().manual_seed(B + 7 * N + 31 * D + K)
    z = 1.5 * torch.randn(B, N, D, generator=gen)
    nn_out = 0.6 * torch.randn(B, N, D * (2 + 3 * K), generator=gen)
    sf, msf = 0.2 * torch.randn(D, generator=gen), 0.2 * torch.randn(D, K, generator=gen)
    mask = None if kind == "none" else (O.chess_mask() if kind == "chess" else O.channel_mask(D))
    ln = torch.randint(max(1, N // 2), N + 1, (B,), generator=gen)
    pad = O.length_mask(ln, N) if kind != "none" else None
    kw = dict(num_mixtures=K, reg_max=3.5, reg_factor=2.0, is_training=True)
    zo, lo, ro = O.mixture_coupling(z, nn_out, mask, scaling_factor=sf, mixture_scaling_factor=msf, channel_padding_mask=pad, **kw)
    zf, lf, rf = ops().mixture_coupling(g(z), g(nn_out), g(mask), scaling_factor=g(sf), mixture_scaling_factor=g(msf),
                                        channel_padding_mask=g(pad), **kw)
    close(zf, zo, **ELEM); loglik_close(lf, lo); loglik_close(rf, ro)
    # fp64 inverse (safeguarded Newton and the reference's bisection)
    zo2, lo2, _ = O.mixture_coupling(zo, nn_out, mask, scaling_factor=sf, mixture_scaling_factor=msf,
                                     channel_padding_mask=pad, reverse=True, **kw)
    for mode in (1, 0):
        _lib.load().cnf_set_inverse_mode(mode)
        try:
            zr, lr, _ = ops().mixture_coupling(g(zo), g(nn_out), g(mask), scaling_factor=g(sf), mixture_scaling_factor=g(msf),
                                               channel_padding_mask=g(pad), reverse=True, **kw)
        finally:
            _lib.load().cnf_set_inverse_mode(1)
        close(zr, zo2, rtol=1e-4, atol=1e-4); loglik_close(lr, lo2)


@pytest.mark.parametrize("scale", [1.0, 4.0, 12.0, 40.0])
def test_mixture_fast_vs_exact(scale):
    """fp32 LDS-staged forward kernel vs the fp64 kernel on the same inputs, from the bulk (scale 1) to far tails
    (scale 40: most elements take the kernel's fp64 fallback branch, u or 1-u < 1e-9)."""
    B, N, D, K = 64, 16, 4, 8
    gen = torch.Generator().manual_seed(int(scale * 10))
    z = scale * torch.randn(B, N, D, generator=gen)
    nn_out = 0.6 * torch.randn(B, N, D * (2 + 3 * K), generator=gen)
    sf, msf = 0.2 * torch.randn(D, generator=gen), 0.2 * torch.randn(D, K, generator=gen)
    mask = O.channel_mask(D)
    kw = dict(num_mixtures=K, scaling_factor=g(sf), mixture_scaling_factor=g(msf), reg_max=3.5, reg_factor=2.0, is_training=True)
    lib = _lib.load()
    zf, lf, rf = ops().mixture_coupling(g(z), g(nn_out), g(mask), **kw)
    lib.cnf_set_math_mode(0)
    try:
        ze, le, re_ = ops().mixture_coupling(g(z), g(nn_out), g(mask), **kw)
    finally:
        lib.cnf_set_math_mode(1)
    fin = torch.isfinite(ze)
    assert torch.equal(fin, torch.isfinite(zf))
    close(zf[fin], ze[fin], rtol=1e-5, atol=1e-5)
    close(lf, le, rtol=2e-5, atol=2e-4)
    close(rf, re_, rtol=1e-4, atol=1e-4)
    # and against the oracle.  Beyond |logit| ~ 36 (1 - u < 1e-16) the reference's fp64 `1 - u` is rounding noise of
    # its log-space evaluation (it returns 36.7 or 50.6 depending on the last bit of u), so scale 40 is only
    # compared kernel to kernel above.
    if scale <= 12.0:
        zo, lo, _ = O.mixture_coupling(z, nn_out, mask, num_mixtures=K, scaling_factor=sf, mixture_scaling_factor=msf,
                                       reg_max=3.5, reg_factor=2.0, is_training=True)
        fo = torch.isfinite(zo)
        close(zf.cpu()[fo], zo[fo], rtol=1e-4, atol=1e-4); loglik_close(lf, lo)


def test_mixture_underflow_fallback_matches_logspace():
    """Force the direct PDF sum to underflow (|z| ~ 900 scales from every mean): the element must take
    the log-space branch and still agree with the oracle (rule: rare data-dependent branch gets its own test)."""
    B, N, D, K = 2, 4, 2, 4
    gen = torch.Generator().manual_seed(3)
    z = torch.randn(B, N, D, generator=gen)
    z[0, 0, 1] = 900.0
    z[1, 2, 1] = -850.0
    nn_out = 0.1 * torch.randn(B, N, D * (2 + 3 * K), generator=gen)
    mask = O.channel_mask(D)
    zo, lo, _ = O.mixture_coupling(z, nn_out, mask, K, None, None)
    zf, lf, _ = ops().mixture_coupling(g(z), g(nn_out), g(mask), K)
    assert torch.isfinite(lf).all()
    loglik_close(lf, lo)
    fin = torch.isfinite(zo)
    close(zf.cpu()[fin], zo[fin], **ELEM)


def test_mixture_full_size_properties():
    """configs[1] shape: B=16384, N=16, D=4, K=8 — forward then inverse recovers z; ldj antisymmetric."""
    B, N, D, K = 16384, 16, 4, 8
    gen = torch.Generator(device="cuda").manual_seed(1)
    z = torch.randn(B, N, D, generator=gen, device="cuda")
    nn_out = 0.5 * torch.randn(B, N, D * (2 + 3 * K), generator=gen, device="cuda")
    mask = g(O.channel_mask(D))
    zf, lf, _ = ops().mixture_coupling(z, nn_out, mask, K)
    zr, lr, _ = ops().mixture_coupling(zf, nn_out, mask, K, reverse=True)
    assert (zr - z).abs().max().item() < 2e-4
    assert ((lf + lr).abs() / lf.abs().clamp(min=1.0)).max().item() < 1e-4
    assert torch.equal(zf[..., :2], z[..., :2])
    zo, lo, _ = O.mixture_coupling(z[:32].cpu(), nn_out[:32].cpu(), mask.cpu(), K, None, None)
    close(zf[:32], zo, **ELEM); loglik_close(lf[:32], lo)


# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("c", load_cases("actnorm"))
def test_actnorm_golden(c):
    mode = c.meta["mode"]
    kw = {}
    if "length" in mode:
        kw["length"] = g(c.length)
    if "mask" in mode:
        kw["channel_padding_mask"] = g(c.pad)
    ldj = g(c.ldj_in.clone())
    zf, lf = ops().actnorm(g(c.z), g(c.bias), g(c.scales), reverse=False, ldj=ldj, **kw)
    assert lf.data_ptr() == ldj.data_ptr()           # in-place `ldj +=` like the reference
    close(zf, c.z_fwd, **ELEM); loglik_close(lf, c.ldj_fwd)
    zr, lr = ops().actnorm(g(c.z_fwd), g(c.bias), g(c.scales), reverse=True, **kw)
    close(zr, c.z_rev, **ELEM); loglik_close(lr, c.ldj_rev)
    b, s = ops().actnorm_data_init(g(c.z), g(c.pad) if "mask" in mode else None)
    close(b, c.init_bias, rtol=1e-5, atol=1e-5); close(s, c.init_scales, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("c", load_cases("ext_actnorm"))
def test_ext_actnorm_golden(c):
    pad = g(c.pad) if c.meta["padded"] else None
    zf, lf = ops().ext_actnorm(g(c.z), g(c.nn_out), reverse=False, channel_padding_mask=pad, ldj=g(c.ldj_in.clone()))
    close(zf, c.z_fwd, **ELEM); loglik_close(lf, c.ldj_fwd)
    zr, lr = ops().ext_actnorm(g(c.z_fwd), g(c.nn_out), reverse=True, channel_padding_mask=pad)
    close(zr, c.z_rev, **ELEM); loglik_close(lr, c.ldj_rev)


@pytest.mark.parametrize("c", load_cases("invconv"))
def test_invconv_golden(c):
    from categoricalnf_amd.layers.flows.permutation_layers import InvertibleConv
    mode = c.meta["mode"]
    kw = {}
    if "length" in mode:
        kw["length"] = g(c.length)
    if "mask" in mode:
        kw["channel_padding_mask"] = g(c.pad)
    D = c.meta["D"]
    layer = InvertibleConv(D, LU_decomposed=c.meta["lu"])
    layer.load_state_dict({k[3:]: v for k, v in c.items() if k.startswith("sd_")})     # reference checkpoint keys
    layer.cuda()
    with torch.no_grad():
        for train in (True, False):
            layer.train(train)
            zf, lf = layer(g(c.z), ldj=g(c.ldj_in.clone()), reverse=False, **kw)
            close(zf, c.z_fwd, **ELEM); loglik_close(lf, c.ldj_fwd)
            zr, lr = layer(g(c.z_fwd), reverse=True, **kw)
            close(zr, c.z_rev, **ELEM); loglik_close(lr, c.ldj_rev)
        assert str(torch.device("cuda:0")) in layer.eval_dict or "cuda:0" in layer.eval_dict


@pytest.mark.parametrize("D", [1, 2, 3, 4, 5, 6, 7, 8, 10])
def test_actnorm_invconv_vs_oracle(D):
    B, N = 257, 19
    gen = torch.Generator().manual_seed(D)
    z = torch.randn(B, N, D, generator=gen)
    w = torch.linalg.qr(torch.randn(D, D, generator=gen))[0] + 0.1 * torch.randn(D, D, generator=gen)
    sldj = torch.slogdet(w)[1]
    ln = torch.randint(N // 2, N + 1, (B,), generator=gen)
    pad = O.length_mask(ln, N)
    zo, lo = O.invconv(z, w, sldj, length=ln, channel_padding_mask=pad)
    zf, lf = ops().invconv(g(z), g(w), g(sldj), length=g(ln), channel_padding_mask=g(pad))
    close(zf, zo, **ELEM); loglik_close(lf, lo)
    bias, sc = torch.randn(1, 1, D, generator=gen), 0.3 * torch.randn(1, 1, D, generator=gen)
    zo, lo = O.actnorm(z, bias, sc, channel_padding_mask=pad)           # length from the mask row-sum
    zf, lf = ops().actnorm(g(z), g(bias), g(sc), channel_padding_mask=g(pad))
    close(zf, zo, **ELEM); loglik_close(lf, lo)


# ------------------------------------------------------------------------------------------------
def test_prior_golden():
    for c in load_cases("prior"):
        k = c.meta["kind"]
        if k == "log_prob":
            close(ops().logistic_log_prob(g(c.x)), c.log_prob, rtol=2e-6, atol=2e-6)
        elif k == "sample":
            s = ops().logistic_from_uniform(g(c.u))
            close(s, c.sample, rtol=2e-6, atol=2e-6)
        else:
            sums = torch.zeros(2, dtype=torch.float64, device="cuda")
            neglog, nll = ops().prior_nll(g(c.z), g(c.ldj), g(c.length), g(c.pad), sums=sums)
            loglik_close(neglog, c.neglog); loglik_close(nll, c.nll)
            assert abs(sums[0].item() / sums[1].item() - float(c.nll_mean)) < 1e-5
            assert sums[1].item() == c.z.size(0)


@pytest.mark.parametrize("B,N,D,kind,has_sf", [(33, 64, 6, "channel", True), (5, 13, 1, "chess", True), (300, 7, 3, "channel", False),
                                                (16, 16, 4, "channel", True), (2, 703, 2, "channel", True), (1000, 1, 2, "channel", True),
                                                (4096, 64, 6, "channel", True)])
def test_affine_coupling_nll_fused_vs_oracle_and_split(B, N, D, kind, has_sf):
    """cnf_affine_coupling_nll == the oracle's coupling followed by its NLL assembly, and == the two separate
    kernels (z / ldj bit for bit: same arithmetic; NLL to summation-order rounding)."""
    gen = torch.Generator().manual_seed(B * 31 + N * 7 + D)
    z = torch.randn(B, N, D, generator=gen)
    nn_out = 0.7 * torch.randn(B, N, 2 * D, generator=gen)
    sf = 0.3 * torch.randn(D, generator=gen) if has_sf else None
    mask = O.chess_mask() if kind == "chess" else O.channel_mask(D)
    ldj0 = torch.randn(B, generator=gen)
    ln = torch.randint(max(1, N // 2), N + 1, (B,), generator=gen)
    pad = O.length_mask(ln, N)
    zo, lo = O.affine_coupling(z, nn_out, mask, scaling_factor=sf, ldj=ldj0.clone())
    nll_o = O.nll_per_sample(zo, lo, ln.float(), pad)
    sums = torch.zeros(2, dtype=torch.float64, device="cuda")
    zf, lf, neglog, nll = ops().affine_coupling_nll(g(z), g(nn_out), g(sf), g(mask), ldj=g(ldj0), length=g(ln),
                                                    channel_padding_mask=g(pad), sums=sums)
    close(zf, zo, **ELEM); loglik_close(lf, lo); loglik_close(nll, nll_o)
    z2, l2 = ops().affine_coupling(g(z), g(nn_out), g(sf), g(mask), ldj=g(ldj0))
    neglog2, nll2 = ops().prior_nll(z2, l2, g(ln), g(pad))
    assert torch.equal(zf, z2) and torch.equal(lf, l2)
    close(neglog, neglog2.cpu(), rtol=2e-6, atol=1e-4); close(nll, nll2.cpu(), rtol=2e-6, atol=2e-5)
    assert sums[1].item() == B and abs(sums[0].item() - nll.double().sum().item()) < 1e-6 * max(1.0, abs(sums[0].item()))
    # batch sum inside the kernel: fixed-point integer atomics -> exact to 2^-32 per row and bit-reproducible
    acc1 = torch.zeros(ops().NLL_ACC_SLOTS, dtype=torch.int64, device="cuda")
    acc2 = torch.zeros(ops().NLL_ACC_SLOTS, dtype=torch.int64, device="cuda")
    for acc in (acc1, acc2):
        za, la, _, na = ops().affine_coupling_nll(g(z), g(nn_out), g(sf), g(mask), ldj=g(ldj0), length=g(ln),
                                                  channel_padding_mask=g(pad), acc=acc)
    assert torch.equal(acc1, acc2) and torch.equal(za, zf) and torch.equal(na, nll)
    s_acc = ops().nll_acc_read(acc1, B)
    want = nll.double().sum().item()
    assert s_acc[1].item() == B and abs(s_acc[0].item() - want) <= B * 2.0 ** -32 + 1e-12 * abs(want)
    # without padding / length (defaults: every token counts, length = N)
    zf, lf, neglog, nll = ops().affine_coupling_nll(g(z), g(nn_out), g(sf), g(mask))
    zo, lo = O.affine_coupling(z, nn_out, mask, scaling_factor=sf)
    loglik_close(nll, O.nll_per_sample(zo, lo, torch.full((B,), float(N)), None))


def test_encoder_full_size_properties():
    """B=16384, N=64, D=6, C=16 (6.3 M latents): batch-slicing invariance (bit-exact), encode -> decode recovers every
    category when the class means are well separated, log-det finite, and agreement with the oracle on a slice."""
    B, N, D, C = 16384, 64, 6, 16
    gen = torch.Generator().manual_seed(11)
    categ = torch.randint(0, C, (B, N), generator=gen)
    table = torch.cat([12.0 * torch.randn(C, D, generator=gen), 0.3 * torch.randn(C, D, generator=gen)], dim=1)
    prior = torch.log_softmax(torch.randn(C, generator=gen), 0)
    u = torch.rand(B * N, 1, D, generator=gen)
    eps = ops().logistic_from_uniform(g(u))
    z, ldj, _ = ops().encoder_forward(g(categ), eps, g(table), g(prior))
    assert torch.isfinite(z).all() and torch.isfinite(ldj).all()
    dec = ops().encoder_decode(z, g(table), g(prior))
    assert (dec.cpu() == categ).float().mean().item() > 0.9999
    zs, ls, _ = ops().encoder_forward(g(categ[:8192]), eps[:8192 * N], g(table), g(prior))
    assert torch.equal(zs, z[:8192]) and torch.equal(ls, ldj[:8192])
    zo, lo, _ = O.encoder_forward(categ[:64], O.logistic_from_uniform(u[:64 * N]), table, prior)
    close(z[:64], zo, **ELEM); loglik_close(ldj[:64], lo)
    assert torch.equal(dec[:64].cpu(), O.encoder_decode(zo, table, prior)[0])


@pytest.mark.parametrize("c", load_cases("encoder"))
def test_encoder_golden(c):
    m = c.meta
    eps = ops().logistic_from_uniform(g(c.u))
    pad = g(c.pad) if m["padded"] else None
    z, ldj, cpl = ops().encoder_forward(g(c.categ), eps, g(c.table), g(c.category_prior), beta=m["beta"],
                                        channel_padding_mask=pad, want_class_prob=True)
    close(z, c.z, **ELEM); loglik_close(ldj, c.ldj)
    dec = ops().encoder_decode(g(c.z), g(c.table), g(c.category_prior))
    assert torch.equal(dec.cpu(), c.decoded)                       # integer indices: bit-exact
    dec = ops().encoder_decode(g(c.z_probe), g(c.table), g(c.category_prior))
    assert torch.equal(dec.cpu(), c.decoded_probe)


@pytest.mark.parametrize("c", load_cases("encoder"))
def test_encoder_module_golden(c):
    """The drop-in module with the reference's state_dict and the reference's noise draw."""
    from categoricalnf_amd.layers.categorical_encoding.linear_encoding import LinearCategoricalEncoding
    m = c.meta
    enc = LinearCategoricalEncoding(num_dimensions=m["D"], flow_config={"num_flows": 0}, vocab_size=m["C"])
    enc.load_state_dict({k[3:]: v for k, v in c.items() if k.startswith("sd_")})
    enc.cuda().train(m["training"])
    kw = dict(channel_padding_mask=g(c.pad)) if m["padded"] else {}
    with torch.no_grad():
        z, ldj, det = enc(g(c.categ), reverse=False, beta=m["beta"], noise=g(c.u), **kw)
        dec, ldj_r, _ = enc(g(c.z), reverse=True)
    close(z, c.z, **ELEM); loglik_close(ldj, c.ldj)
    assert torch.equal(dec.cpu(), c.decoded) and float(ldj_r.abs().sum()) == 0.0
    if m["training"]:
        assert set(det) == {"avg_token_prob", "avg_token_bpd", "z_min", "z_max", "z_std"}
        close(det["avg_token_prob"], c.detail_avg_token_prob, rtol=1e-4, atol=1e-5)
        close(det["z_std"], c.detail_z_std, rtol=1e-4, atol=1e-5)
    else:
        assert det == {}


def test_encoder_cpu_generator_matches_reference_seed():
    """CNF_NOISE=cpu mode: torch.manual_seed reproduces the reference's own noise (linear_encoding.py:76)."""
    from categoricalnf_amd.layers.categorical_encoding.linear_encoding import LinearCategoricalEncoding
    c = load_cases("encoder")[0]
    m = c.meta
    enc = LinearCategoricalEncoding(num_dimensions=m["D"], flow_config={"num_flows": 0}, vocab_size=m["C"])
    enc.load_state_dict({k[3:]: v for k, v in c.items() if k.startswith("sd_")})
    enc.cuda().train(m["training"])
    enc.noise_generator = "cpu"
    torch.manual_seed(500 + 0)
    with torch.no_grad():
        z, ldj, _ = enc(g(c.categ), reverse=False, beta=m["beta"])
    close(z, c.z, **ELEM); loglik_close(ldj, c.ldj)


def test_sigmoid_and_dequant_golden():
    for c in load_cases("sigmoid"):
        if not c.meta["reverse_layer"]:
            a, la = ops().sigmoid_flow(g(c.z), reverse=False)
            b, lb = ops().sigmoid_flow(g(c.u), reverse=True)
        else:
            a, la = ops().sigmoid_flow(g(c.u), reverse=True)
            b, lb = ops().sigmoid_flow(g(c.z), reverse=False)
        close(a, c.out_fwd, **ELEM); loglik_close(la, c.ldj_fwd); close(b, c.out_rev, **ELEM); loglik_close(lb, c.ldj_rev)

    from categoricalnf_amd.layers.categorical_encoding.variational_dequantization import VariationalDequantization
    c = load_cases("dequant")[0]
    m = c.meta
    hidden, emb = m["hidden"], m["emb"]

    class Net(nn.Module):
        def __init__(self, c_out):
            super().__init__()
            self.inp = nn.Linear(1, hidden)
            self.main = nn.Sequential(nn.Linear(hidden + emb, hidden), nn.ReLU(), nn.Linear(hidden, c_out))

        def forward(self, x, ext_input, **kw):
            return self.main(torch.cat([self.inp(x), ext_input], dim=-1))

    vd = VariationalDequantization(vocab_size=m["C"], flow_config={"num_flows": m["num_flows"], "model_func": lambda c_out: Net(c_out),
                                                                    "block_type": "Linear"}, default_embed_layer_dims=emb)
    vd.load_state_dict({k[3:]: v for k, v in c.items() if k.startswith("sd_")})
    vd.cuda()
    with torch.no_grad():
        z, ldj = vd(g(c.categ), reverse=False, noise=g(c.u))
        rec, _ = vd(z, reverse=True)
    close(z, c.z, rtol=1e-4, atol=1e-4); loglik_close(ldj, c.ldj)
    assert torch.equal(rec.cpu(), c.categ)                          # inverse∘forward bit-exact on integer indices
    assert torch.equal(rec.cpu(), c.decoded)


# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("c", load_cases("flow_stack"))
def test_flow_stack_config0(c):
    """BASELINE configs[0]: |S|=16, d_latent=2, 4 affine couplings, batch 256 — whole-model per-sample
    log-likelihood (<= 1e-4 relative), bits/dim (+-0.01) and decoded indices vs the reference."""
    from categoricalnf_amd.layers.flows.flow_model import FlowModel
    from categoricalnf_amd.layers.flows.activation_normalization import ActNormFlow
    from categoricalnf_amd.layers.flows.permutation_layers import InvertibleConv
    from categoricalnf_amd.layers.flows.coupling_layer import CouplingLayer
    from categoricalnf_amd.layers.flows.distributions import LogisticDistribution
    from categoricalnf_amd.layers.categorical_encoding.linear_encoding import LinearCategoricalEncoding
    m = c.meta
    D, hidden = m["D"], m["hidden"]
    mk = lambda c_out: nn.Sequential(nn.Linear(D, hidden), nn.GELU(), nn.Linear(hidden, c_out))
    layers = [LinearCategoricalEncoding(num_dimensions=D, flow_config={"num_flows": 0}, vocab_size=m["C"])]
    for _ in range(m["flows"]):
        layers += [ActNormFlow(D), InvertibleConv(D), CouplingLayer(D, CouplingLayer.create_channel_mask(D), mk)]
    model = FlowModel(layers)
    model.load_state_dict({k[3:]: v for k, v in c.items() if k.startswith("sd_")})
    model.cuda().eval()
    ln = torch.full((m["B"],), m["N"], dtype=torch.long, device="cuda")
    with torch.no_grad():
        z, ldj = model(g(c.categ), reverse=False, length=ln, noise=g(c.u))
        neglog, nll = ops().prior_nll(z, ldj, ln)
        dec, _ = model(g(c.z), reverse=True, length=ln)
    close(z, c.z, rtol=1e-4, atol=1e-4)
    loglik_close(ldj, c.ldj)
    loglik_close(nll, c.nll)
    bpd = float(np.log2(np.exp(1)) * nll.mean().item())
    assert abs(bpd - float(c.bpd)) < 0.01
    assert torch.equal(dec.cpu(), c.decoded)
    with torch.no_grad():
        dec_r, _ = model.reverse(g(c.z))                     # the convenience method = forward(reverse=True); no padding here
    assert torch.equal(dec_r.cpu(), c.decoded)
    # FlowModel.nll: the last affine coupling and the NLL assembly as one kernel — same latents, log-det and NLL
    sums = torch.zeros(2, dtype=torch.float64, device="cuda")
    z2, ldj2, nll2 = model.nll(g(c.categ), length=ln, noise=g(c.u), sums=sums)
    assert torch.equal(z2, z) and torch.equal(ldj2, ldj)
    close(nll2, nll, rtol=2e-6, atol=2e-5); loglik_close(nll2, c.nll)
    assert sums[1].item() == m["B"] and abs(sums[0].item() - nll2.double().sum().item()) < 1e-6 * abs(sums[0].item()) + 1e-9


def test_nan_raises_reference_assertion():
    from categoricalnf_amd.layers.flows.flow_model import FlowModel
    from categoricalnf_amd.layers.flows.activation_normalization import ActNormFlow
    model = FlowModel([ActNormFlow(3, data_init=False)]).cuda()
    z = torch.randn(4, 5, 3, device="cuda")
    z[1, 2, 0] = float("nan")
    with pytest.raises(AssertionError):
        with torch.no_grad():
            model(z)
    with torch.no_grad():
        model(torch.randn(4, 5, 3, device="cuda"))       # flag word was cleared


def test_cpu_tensor_is_rejected():
    from categoricalnf_amd import ops as o
    with pytest.raises(o.HipOnlyError):
        o.affine_coupling(torch.randn(2, 3, 2), torch.randn(2, 3, 4), None, None)


# ------------------------------------------------------------------------------------------------
def _set_model(meta):
    from categoricalnf_amd.experiments.set_modeling import FlowSetModeling, SetShufflingDataset
    params = {"set_size": meta["set_size"], "coupling_hidden_layers": meta["transformer_layers"],
              "coupling_hidden_size": meta["hidden"], "coupling_num_flows": meta["flows"], "coupling_mask_ratio": 0.5,
              "coupling_num_mixtures": meta["K"],
              "categ_encoding": {"use_dequantization": False, "use_variational": False, "use_decoder": False,
                                 "num_dimensions": meta["D"], "flow_config": {"num_flows": 0, "hidden_layers": 2, "hidden_size": 128},
                                 "decoder_config": {"num_layers": 1, "hidden_size": 64}}}
    return FlowSetModeling(params, SetShufflingDataset), SetShufflingDataset


@pytest.mark.parametrize("compact", [False, True])
@pytest.mark.parametrize("golden", ["set_shuffling_model.npz", "set_shuffling_trained.npz"])
def test_set_shuffling_trained_model_bits_per_dim(golden, compact):
    """A FlowSetModeling trained WITH THE REFERENCE for 6000 CPU iterations (CNF_TRAIN_ITERS=6000, 3.59 bpd; oracle/gen_set_shuffling_golden.py)
    and one trained by this package's driver on an MI355X for 50000 iterations (2.94 bpd, sharp mixtures) and evaluated
    by the REFERENCE on the CPU (oracle/gen_set_shuffling_trained_golden.py): same weights on the HIP path must give
    the reference's per-sample log-likelihood (1e-4 relative), decoded indices (bit-exact) and validation bits/dim
    (+-0.01) — BASELINE.json north_star."""
    import json
    import os
    from tests.golden_util import GOLDEN_DIR
    data = np.load(os.path.join(GOLDEN_DIR, golden))
    meta = json.loads(bytes(data["meta"]).decode())
    model, dataset = _set_model(meta)
    model.load_state_dict({k[3:]: torch.from_numpy(np.array(data[k])) for k in data.files if k.startswith("sd_")})
    model.cuda().eval()
    if compact:
        # the same reference-trained weights with every mixture coupling on the compact parameter layout (the last Linear of its
        # Transformer sub-network applies the transformed channels' rows only): the reference's numbers all the same
        on = [layer.enable_compact_params() for layer in model.flow_layers if hasattr(layer, "enable_compact_params")]
        assert on and all(on)
    S = meta["set_size"]
    x = torch.from_numpy(data["x256"]).cuda()
    ln = torch.full((x.size(0),), S, dtype=torch.long, device="cuda")
    with torch.no_grad():
        z, ldj = model(x, reverse=False, length=ln, beta=1, noise=torch.from_numpy(data["u256"]).cuda())
        _, nll = ops().prior_nll(z, ldj, ln)
        dec, _ = model(torch.from_numpy(data["z256"]).cuda(), reverse=True, length=ln)
    close(z, torch.from_numpy(data["z256"]), rtol=2e-4, atol=2e-4)
    loglik_close(ldj, torch.from_numpy(data["ldj256"]))
    loglik_close(nll, torch.from_numpy(data["nll256"]))
    assert torch.equal(dec.cpu(), torch.from_numpy(data["dec256"]))
    # full deterministic validation set (32768 permutations, seed 123), device-generated noise
    val = torch.from_numpy(dataset(S, train=False, val=True).shuffle_set).long().cuda()
    total = torch.zeros(2, dtype=torch.float64, device="cuda")
    sums = torch.zeros(2, dtype=torch.float64, device="cuda")
    torch.manual_seed(0)
    with torch.no_grad():
        for i in range(0, val.size(0), 4096):
            xb = val[i:i + 4096]
            lb = torch.full((xb.size(0),), S, dtype=torch.long, device="cuda")
            zb, lj = model(xb, reverse=False, length=lb, beta=1)
            ops().prior_nll(zb, lj, lb, sums=sums)
            total += sums
    from categoricalnf_amd.distributed import allreduce_nll
    mean_nll, bpd = allreduce_nll(total)
    assert abs(bpd - meta["val_bpd"]) < 0.01, (bpd, meta["val_bpd"])
    assert bpd > dataset.optimum_bpd(S) - 1e-3
    # FlowModel.nll on a flow that ends in a mixture coupling: the separate prior kernel, same numbers as the oracle
    x64 = torch.from_numpy(data["x256"][:64]).long().cuda()
    l64 = torch.full((64,), S, dtype=torch.long, device="cuda")
    u64 = torch.from_numpy(data["u256"][:64 * S]).cuda()
    zq, lq, nq = model.nll(x64, length=l64, noise=u64, beta=1)
    loglik_close(nq, torch.from_numpy(data["nll256"][:64]))


# ------------------------------------------------------------------------------------------------
class _Stub(nn.Module):
    """coupling sub-network stand-in that returns a fixed tensor (as in oracle/gen_golden.py)"""

    def __init__(self):
        super().__init__()
        self.out = None

    def forward(self, *args, **kwargs):
        return self.out


@pytest.mark.parametrize("c", load_cases("encoder_linear_flows"))
def test_linear_flow_encoder_golden(c):
    """num_flows > 0: the class flows are ExtActNorm + 1x1 conv + affine coupling kernels composed over [T*C,1,D]."""
    from categoricalnf_amd.layers.categorical_encoding.linear_encoding import LinearCategoricalEncoding
    m = c.meta
    enc = LinearCategoricalEncoding(num_dimensions=m["D"], flow_config={"num_flows": m["flows"], "hidden_layers": 1, "hidden_size": m["hidden"]},
                                    vocab_size=m["C"], default_embed_layer_dims=m["embed"])
    enc.load_state_dict({k[3:]: v for k, v in c.items() if k.startswith("sd_")})
    assert enc.info() == m["infos"]
    enc.cuda().train(m["training"])
    kw = dict(channel_padding_mask=g(c.pad)) if m["padded"] else {}
    with torch.no_grad():
        z, ldj, _ = enc(g(c.categ), reverse=False, beta=1, noise=g(c.u), **kw)
        dec, _, _ = enc(g(c.z), reverse=True)
    close(z, c.z, rtol=1e-4, atol=1e-4); loglik_close(ldj, c.ldj)
    assert torch.equal(dec.cpu(), c.decoded)


@pytest.mark.parametrize("c", load_cases("node_edge_coupling"))
def test_node_edge_coupling_golden(c):
    from categoricalnf_amd.experiments.graph_node_edge_coupling import NodeEdgeCoupling, NodeEdgeFlowWrapper
    from categoricalnf_amd.layers.flows.coupling_layer import CouplingLayer
    from categoricalnf_amd.layers.flows.activation_normalization import ActNormFlow
    m = c.meta
    layer = NodeEdgeCoupling(c_in_nodes=m["Dn"], c_in_edges=m["De"], mask_nodes=CouplingLayer.create_channel_mask(m["Dn"]),
                             mask_edges=CouplingLayer.create_channel_mask(m["De"]), num_mixtures_nodes=m["Kn"],
                             num_mixtures_edges=m["Ke"], model_func=lambda c_out_nodes, c_out_edges: _Stub(),
                             regularizer_max=3.5, regularizer_factor=2)
    layer.load_state_dict({k[3:]: v for k, v in c.items() if k.startswith("sd_")})
    assert layer.info() == m["info"]
    layer.cuda().train(m["training"])
    layer.nn.out = (g(c.nn_nodes), g(c.nn_edges))
    kw = dict(length=g(c.length), channel_padding_mask=g(c.pad), mask_valid=g(c.mask_valid))
    with torch.no_grad():
        zn, ze, ldj, det = layer(g(c.z_nodes), g(c.z_edges), reverse=False, **kw)
        zn_r, ze_r, ldj_r, det_r = layer(g(c.z_nodes_fwd), g(c.z_edges_fwd), reverse=True, **kw)
    close(zn, c.z_nodes_fwd, **ELEM); close(ze, c.z_edges_fwd, **ELEM); loglik_close(ldj, c.ldj_fwd)
    loglik_close(det["regularizer_nodes_ldj"], c.reg_nodes); loglik_close(det["regularizer_edges_ldj"], c.reg_edges)
    close(zn_r, c.z_nodes_rev, rtol=1e-4, atol=1e-4); close(ze_r, c.z_edges_rev, rtol=1e-4, atol=1e-4)
    loglik_close(ldj_r, c.ldj_rev)
    assert "regularizer_nodes_ldj" not in det_r
    # the reference's two-call static path (get_mixt_params + run_with_params) gives the same nodes result
    from categoricalnf_amd.layers.flows.mixture_cdf_layer import MixtureCDFCoupling
    mask_n = layer.mask_nodes[None, :1, :]
    params = MixtureCDFCoupling.get_mixt_params(g(c.nn_nodes) * g(c.pad), mask_n, m["Kn"], layer.scaling_factor_nodes.data,
                                                layer.mixture_scaling_factor_nodes.data)
    z64, l64, reg64 = MixtureCDFCoupling.run_with_params(g(c.z_nodes).double(), *params, reverse=False, is_training=m["training"],
                                                         reg_max=3.5, reg_factor=2, mask=mask_n, channel_padding_mask=g(c.pad),
                                                         return_reg_ldj=True)
    close((z64.float() * g(c.pad)), c.z_nodes_fwd, **ELEM)
    wrap = NodeEdgeFlowWrapper(ActNormFlow(m["Dn"], data_init=False), ActNormFlow(m["De"], data_init=False))
    wrap.load_state_dict({k[4:]: v for k, v in c.items() if k.startswith("wsd_")})
    wrap.cuda()
    with torch.no_grad():
        wn, we, wl = wrap(g(c.z_nodes), g(c.z_edges), ldj=g(c.wrap_ldj_in.clone()), reverse=False, **kw)
    close(wn, c.wrap_nodes, **ELEM); close(we, c.wrap_edges, **ELEM); loglik_close(wl, c.wrap_ldj)


def test_data_dependent_init_driver_golden():
    """FlowModel.initialize_data_dependent: ActNorm -> 1x1 conv -> ActNorm on three ragged batches."""
    from categoricalnf_amd.layers.flows.flow_model import FlowModel
    from categoricalnf_amd.layers.flows.activation_normalization import ActNormFlow
    from categoricalnf_amd.layers.flows.permutation_layers import InvertibleConv
    c = load_cases("data_init")[0]
    D = c.meta["D"]
    model = FlowModel([ActNormFlow(D), InvertibleConv(D), ActNormFlow(D)])
    model.load_state_dict({k[4:]: v for k, v in c.items() if k.startswith("sd0_")})
    model.cuda()
    batches = [(g(c["z%d" % i]), {"length": g(c["length%d" % i]), "channel_padding_mask": g(c["pad%d" % i])})
               for i in range(len(c.meta["batches"]))]
    model.initialize_data_dependent(batches)
    got = model.state_dict()
    for k in ("flow_layers.0.bias", "flow_layers.0.scales", "flow_layers.2.bias", "flow_layers.2.scales"):
        close(got[k], c["sd_" + k], rtol=1e-4, atol=1e-4)
        assert tuple(got[k].shape) == (1, 1, D)


def test_autoregressive_module_golden():
    from categoricalnf_amd.layers.flows.autoregressive_coupling import AutoregressiveMixtureCDFCoupling
    c = [x for x in load_cases("mixture_coupling") if x.meta["mask_kind"] == "none"][0]
    m = c.meta
    layer = AutoregressiveMixtureCDFCoupling(c_in=m["D"], model_func=lambda c_out: _Stub(), num_mixtures=m["K"])
    layer.scaling_factor.data, layer.mixture_scaling_factor.data = c.scaling_factor.clone(), c.mixture_scaling_factor.clone()
    layer.cuda()
    layer.nn.out = g(c.nn_out)
    ldj0 = torch.randn(m["B"], device="cuda")
    with torch.no_grad():
        z, ldj = layer(g(c.z), ldj=ldj0)
    close(z, c.z_fwd, **ELEM); loglik_close(ldj - ldj0, c.ldj_fwd)
    with pytest.raises(NotImplementedError):
        layer(g(c.z), reverse=True)


# ------------------------------------------------------------------------------------------------
# Backward kernels against the reference's autograd (tests/golden/grads.npz)
GRAD = dict(rtol=2e-4, atol=2e-4)


# Error budget of the gradient checks.  grads.npz carries every gradient twice: from the reference's own fp32 autograd
# (`g_*`, `gp_*`) and from a float64 run of the same reference modules (`g64_*`, `gp64_*`; oracle/gen_golden.py:gen_grads).
# Their difference is the rounding noise of the REFERENCE's fp32 gradients (1e-7 ... 5e-6 absolute on gradients of size
# 1 ... 30).  A HIP gradient must lie within GRAD_K times that noise of the float64 gradient, plus GRAD_FLOOR of the
# gradient's scale for entries whose reference noise happens to be zero (hardware exp / log / rcp are good to ~1e-7
# relative, sums run in another order).  Measured on an MI355X: the HIP gradients deviate from the float64 ones by 0.2 ... 5.6
# times the reference's own noise (25 x where that noise happens to be 1e-8), at most 0.3 of this bound
# (profiles/r03_grad_budget.txt; CNF_GRAD_REPORT=1 prints the table).  Where the reference has no float64 run (the 1x1 convolution's inverse casts its
# weight to float) the flat tolerance stays.
GRAD_K = 8.0
GRAD_FLOOR = 5e-7


def grad_close(actual, c, key, flat=None):
    import os
    k64 = key.replace("gp_", "gp64_", 1) if key.startswith("gp_") else key.replace("g_", "g64_", 1)
    a = actual.detach().double().cpu()
    if k64 not in c:
        close(actual, c[key], **(flat or GRAD))
        return
    g32, g64 = c[key].double(), c[k64].double()
    assert a.shape == g64.shape, (a.shape, g64.shape)
    noise, scale = (g32 - g64).abs().max().item(), max(g64.abs().max().item(), 1e-30)
    bound = GRAD_K * noise + GRAD_FLOOR * scale
    dev = (a - g64).abs().max().item()
    if os.environ.get("CNF_GRAD_REPORT"):
        print("grad %-44s dev %.2e  ref noise %.2e  scale %.2e  dev/bound %.3f  dev/noise %.1f" % (
            c.meta["layer"] + ":" + key, dev, noise, scale, dev / bound, dev / max(noise, 1e-30)))
    assert dev <= bound, "%s: |g - g64| = %.3g exceeds %.0f x reference fp32 noise (%.3g) + %.0e x scale (%.3g)" % (
        key, dev, GRAD_K, noise, GRAD_FLOOR, scale)


def _leaf(t):
    return t.cuda().clone().requires_grad_(True)


def _grad_cases(layer):
    return [c for c in load_cases("grads") if c.meta["layer"] == layer]


@pytest.mark.parametrize("c", _grad_cases("affine"))
def test_affine_backward(c):
    from categoricalnf_amd.layers.flows.coupling_layer import CouplingLayer
    m = c.meta
    layer = CouplingLayer(c_in=m["D"], mask=c.mask, model_func=lambda c_out: _Stub()).cuda()
    layer.scaling_factor.data = g(c.scaling_factor.clone())
    z, nn_out, ldj = _leaf(c.z), _leaf(c.nn_out), _leaf(c.ldj)
    layer.nn.out = nn_out
    zo, lo = layer(z, ldj=ldj, reverse=m["reverse"])
    ((zo * g(c.wz)).sum() + (lo * g(c.wl)).sum()).backward()
    grad_close(z.grad, c, "g_z"); grad_close(nn_out.grad, c, "g_nn"); grad_close(ldj.grad, c, "g_ldj")
    grad_close(layer.scaling_factor.grad, c, "g_sf")


@pytest.mark.parametrize("c", _grad_cases("actnorm"))
def test_actnorm_backward(c):
    from categoricalnf_amd.layers.flows.activation_normalization import ActNormFlow
    m = c.meta
    layer = ActNormFlow(m["D"]).cuda()
    layer.bias.data, layer.scales.data = g(c.bias.clone()), g(c.scales.clone())
    z, ldj = _leaf(c.z), _leaf(c.ldj)
    kw = {}
    if "length" in m["mode"]:
        kw["length"] = g(c.length)
    if "mask" in m["mode"]:
        kw["channel_padding_mask"] = g(c.pad)
    zo, lo = layer(z, ldj=ldj * 1.0, reverse=m["reverse"], **kw)
    ((zo * g(c.wz)).sum() + (lo * g(c.wl)).sum()).backward()
    grad_close(z.grad, c, "g_z"); grad_close(ldj.grad, c, "g_ldj")
    grad_close(layer.bias.grad, c, "g_bias"); grad_close(layer.scales.grad, c, "g_scales")


@pytest.mark.parametrize("c", _grad_cases("ext_actnorm"))
def test_ext_actnorm_backward(c):
    from categoricalnf_amd.layers.flows.activation_normalization import ExtActNormFlow
    m = c.meta
    net = _Stub()
    layer = ExtActNormFlow(m["D"], net=net)
    z, nn_out, ldj = _leaf(c.z), _leaf(c.nn_out), _leaf(c.ldj)
    net.out = nn_out
    kw = dict(channel_padding_mask=g(c.pad)) if m["padded"] else {}
    zo, lo = layer(z, ldj * 1.0, ext_input=z, reverse=m["reverse"], **kw)
    ((zo * g(c.wz)).sum() + (lo * g(c.wl)).sum()).backward()
    grad_close(z.grad, c, "g_z"); grad_close(nn_out.grad, c, "g_nn"); grad_close(ldj.grad, c, "g_ldj")


@pytest.mark.parametrize("c", _grad_cases("invconv"))
def test_invconv_backward(c):
    from categoricalnf_amd.layers.flows.permutation_layers import InvertibleConv
    m = c.meta
    layer = InvertibleConv(m["D"], LU_decomposed=m["lu"])
    layer.load_state_dict({k[3:]: v for k, v in c.items() if k.startswith("sd_")})
    layer.cuda().train()
    x, ldj = _leaf(c.x), _leaf(c.ldj)
    kw = {}
    if "length" in m["mode"]:
        kw["length"] = g(c.length)
    if "mask" in m["mode"]:
        kw["channel_padding_mask"] = g(c.pad)
    zo, lo = layer(x, ldj=ldj, reverse=m["reverse"], **kw)
    ((zo * g(c.wz)).sum() + (lo * g(c.wl)).sum()).backward()
    grad_close(x.grad, c, "g_x"); grad_close(ldj.grad, c, "g_ldj")
    for name, p in layer.named_parameters():
        grad_close(p.grad, c, "gp_" + name, flat=dict(rtol=1e-3, atol=1e-3))


def test_prior_and_sigmoid_backward():
    from categoricalnf_amd.layers.flows.distributions import LogisticDistribution
    from categoricalnf_amd.layers.flows.sigmoid_layer import SigmoidFlow
    from categoricalnf_amd import functional as Fn
    c = _grad_cases("log_prob")[0]
    x = _leaf(c.x)
    (LogisticDistribution().log_prob(x) * g(c.w)).sum().backward()
    grad_close(x.grad, c, "g_x")
    c = _grad_cases("nll")[0]
    z, ldj = _leaf(c.z), _leaf(c.ldj)
    nll = Fn.PriorNllFn.apply(z, ldj, g(c.length), g(c.pad))
    (nll * g(c.wl)).sum().backward()
    grad_close(z.grad, c, "g_z"); grad_close(ldj.grad, c, "g_ldj")
    for c in _grad_cases("sigmoid"):
        z, ldj = _leaf(c.z), _leaf(c.ldj)
        zo, lo = SigmoidFlow()(z, ldj=ldj, reverse=c.meta["reverse"])
        ((zo * g(c.wz)).sum() + (lo * g(c.wl)).sum()).backward()
        grad_close(z.grad, c, "g_z"); grad_close(ldj.grad, c, "g_ldj")


@pytest.mark.parametrize("c", _grad_cases("mixture"))
def test_mixture_backward(c):
    from categoricalnf_amd.layers.flows.mixture_cdf_layer import MixtureCDFCoupling
    from categoricalnf_amd.layers.flows.autoregressive_coupling import AutoregressiveMixtureCDFCoupling
    m = c.meta
    if m["mask_kind"] == "none":
        layer = AutoregressiveMixtureCDFCoupling(c_in=m["D"], model_func=lambda c_out: _Stub(), num_mixtures=m["K"])
    else:
        layer = MixtureCDFCoupling(c_in=m["D"], mask=c.mask, model_func=lambda c_out: _Stub(), num_mixtures=m["K"],
                                   regularizer_max=m["reg_max"], regularizer_factor=m["reg_factor"])
    layer.cuda().train(m["training"])
    layer.scaling_factor.data, layer.mixture_scaling_factor.data = g(c.scaling_factor.clone()), g(c.mixture_scaling_factor.clone())
    z, nn_out = _leaf(c.z), _leaf(c.nn_out)
    layer.nn.out = nn_out
    res = layer(z, reverse=False, **(dict(channel_padding_mask=g(c.pad)) if m["padded"] else {}))
    ((res[0] * g(c.wz)).sum() + (res[1] * g(c.wl)).sum()).backward()
    grad_close(z.grad, c, "g_z"); grad_close(nn_out.grad, c, "g_nn")
    grad_close(layer.scaling_factor.grad, c, "g_sf"); grad_close(layer.mixture_scaling_factor.grad, c, "g_msf")
    with pytest.raises(NotImplementedError):
        layer(z, reverse=True)


@pytest.mark.parametrize("c", _grad_cases("encoder"))
def test_encoder_backward(c):
    from categoricalnf_amd.layers.categorical_encoding.linear_encoding import LinearCategoricalEncoding
    m = c.meta
    enc = LinearCategoricalEncoding(num_dimensions=m["D"], flow_config={"num_flows": 0}, vocab_size=m["C"], default_embed_layer_dims=8)
    enc.load_state_dict({k[3:]: v for k, v in c.items() if k.startswith("sd_")})
    enc.cuda().eval()
    kw = dict(channel_padding_mask=g(c.pad)) if m["padded"] else {}
    zo, lo, _ = enc(g(c.categ), reverse=False, beta=m["beta"], noise=g(c.u), **kw)
    ((zo * g(c.wz)).sum() + (lo * g(c.wl)).sum()).backward()
    for name, p in enc.named_parameters():
        grad_close(p.grad, c, "gp_" + name)


def test_training_steps_reduce_nll_on_set_shuffling():
    """End to end: a few optimiser steps of the set-shuffling flow (encoder + ActNorm + 1x1 conv + mixture coupling
    with a Transformer sub-network), every flow layer differentiated by the HIP backward kernels."""
    from categoricalnf_amd import functional as Fn
    torch.manual_seed(0)
    np.random.seed(0)
    model, dataset = _set_model(dict(set_size=16, transformer_layers=1, hidden=32, flows=2, K=8, D=4))
    model.cuda().train()
    rng = np.random.RandomState(1)
    draw = lambda n: torch.from_numpy(np.stack([rng.permutation(16) for _ in range(n)])).long().cuda()
    ln = torch.full((128,), 16, dtype=torch.long, device="cuda")
    model.initialize_data_dependent([(draw(128), {"length": ln}) for _ in range(4)])
    opt = torch.optim.Adam(model.parameters(), lr=2e-3)
    losses = []
    for it in range(60):
        z, ldj = model(draw(128), reverse=False, length=ln, beta=1)
        loss = Fn.PriorNllFn.apply(z, ldj, ln, None).mean()
        opt.zero_grad()
        loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 0.25)
        opt.step()
        losses.append(float(loss.detach()))
    assert all(np.isfinite(losses))
    assert np.mean(losses[-10:]) < np.mean(losses[:10]) - 0.1, (losses[:3], losses[-3:])
    assert all(p.grad is not None for p in model.parameters() if p.requires_grad)


@pytest.mark.parametrize("c", [x for x in _grad_cases("mixture") if x.meta["mask_kind"] != "none"])
def test_mixture_static_api_backward(c):
    """The reference's own call pattern (get_mixt_params + run_with_params, as NodeEdgeCoupling uses it) is
    differentiable end to end and gives the reference's gradients."""
    from categoricalnf_amd.layers.flows.mixture_cdf_layer import MixtureCDFCoupling
    m = c.meta
    z, nn_out = _leaf(c.z), _leaf(c.nn_out)
    sf, msf = _leaf(c.scaling_factor), _leaf(c.mixture_scaling_factor)
    mask = O.expand_mask(c.mask, c.z).cuda()
    pad = g(c.pad) if m["padded"] else None
    p = MixtureCDFCoupling.get_mixt_params(nn_out, mask, m["K"], sf, msf)
    z64, l64, _ = MixtureCDFCoupling.run_with_params(z.double(), *p, reverse=False, reg_max=m["reg_max"], reg_factor=m["reg_factor"],
                                                     mask=mask, channel_padding_mask=pad if pad is not None else torch.ones_like(z),
                                                     is_training=m["training"], return_reg_ldj=True)
    zo = z64.float() * (pad if pad is not None else 1.0)
    ((zo * g(c.wz)).sum() + (l64.float() * g(c.wl)).sum()).backward()
    grad_close(z.grad, c, "g_z"); grad_close(nn_out.grad, c, "g_nn")
    grad_close(sf.grad, c, "g_sf"); grad_close(msf.grad, c, "g_msf")


@pytest.mark.parametrize("c", _grad_cases("affine"))
def test_affine_static_api_backward(c):
    from categoricalnf_amd.layers.flows.coupling_layer import CouplingLayer
    m = c.meta
    z, nn_out, sf = _leaf(c.z), _leaf(c.nn_out), _leaf(c.scaling_factor)
    mask = O.expand_mask(c.mask, c.z).cuda()
    s, t = CouplingLayer.get_coup_params(nn_out, mask, scaling_factor=sf)
    zo, lo = CouplingLayer.run_with_params(z, s, t, reverse=m["reverse"])
    ((zo * g(c.wz)).sum() + ((g(c.ldj) + lo) * g(c.wl)).sum()).backward()
    grad_close(z.grad, c, "g_z"); grad_close(nn_out.grad, c, "g_nn"); grad_close(sf.grad, c, "g_sf")


@pytest.mark.parametrize("D", [1, 2, 3, 4, 5, 6, 8])
def test_fused_actnorm_invconv_equals_the_two_layers(D):
    """cnf_actnorm_invconv == cnf_actnorm followed by cnf_invconv (and the pair backwards), bit for bit."""
    B, N = 129, 23
    gen = torch.Generator().manual_seed(100 + D)
    z = torch.randn(B, N, D, generator=gen).cuda()
    bias, sc = torch.randn(1, 1, D, generator=gen).cuda(), (0.3 * torch.randn(1, 1, D, generator=gen)).cuda()
    w = (torch.linalg.qr(torch.randn(D, D, generator=gen))[0] + 0.1 * torch.randn(D, D, generator=gen)).cuda()
    w_inv = torch.inverse(w.double()).float()
    sldj = torch.slogdet(w)[1]
    ln = torch.randint(N // 2, N + 1, (B,), generator=gen).cuda()
    pad = g(O.length_mask(ln.cpu(), N))
    ldj0 = torch.randn(B, generator=gen).cuda()
    for kw in (dict(length=ln, channel_padding_mask=pad), dict(), dict(channel_padding_mask=pad), dict(length=ln)):
        z1, l1 = ops().actnorm(z, bias, sc, ldj=ldj0.clone(), **kw)
        z1, l1 = ops().invconv(z1, w, sldj, ldj=l1, **kw)
        zf, lf = ops().actnorm_invconv(z, bias, sc, w, sldj, ldj=ldj0.clone(), **kw)
        assert torch.equal(zf, z1) and torch.equal(lf, l1)
        z2, l2 = ops().invconv(z1, w_inv, sldj, reverse=True, ldj=ldj0.clone(), **kw)
        z2, l2 = ops().actnorm(z2, bias, sc, reverse=True, ldj=l2, **kw)
        zr, lr = ops().actnorm_invconv(z1, bias, sc, w_inv, sldj, reverse=True, ldj=ldj0.clone(), **kw)
        assert torch.equal(zr, z2) and torch.equal(lr, l2)


def test_flow_model_layer_fusion_is_unobservable():
    from categoricalnf_amd import ops as o
    c = load_cases("flow_stack")[1]
    m = c.meta
    from categoricalnf_amd.layers.flows.flow_model import FlowModel
    from categoricalnf_amd.layers.flows.activation_normalization import ActNormFlow
    from categoricalnf_amd.layers.flows.permutation_layers import InvertibleConv
    from categoricalnf_amd.layers.flows.coupling_layer import CouplingLayer
    from categoricalnf_amd.layers.categorical_encoding.linear_encoding import LinearCategoricalEncoding
    D, hidden = m["D"], m["hidden"]
    mk = lambda c_out: nn.Sequential(nn.Linear(D, hidden), nn.GELU(), nn.Linear(hidden, c_out))
    layers = [LinearCategoricalEncoding(num_dimensions=D, flow_config={"num_flows": 0}, vocab_size=m["C"])]
    for _ in range(m["flows"]):
        layers += [ActNormFlow(D), InvertibleConv(D), CouplingLayer(D, CouplingLayer.create_channel_mask(D), mk)]
    model = FlowModel(layers)
    model.load_state_dict({k[3:]: v for k, v in c.items() if k.startswith("sd_")})
    model.cuda().eval()
    ln = torch.full((m["B"],), m["N"], dtype=torch.long, device="cuda")
    outs = {}
    try:
        for fuse in (True, False):
            o.FUSE_LAYERS = fuse
            with torch.no_grad():
                z, ldj = model(g(c.categ), reverse=False, length=ln, noise=g(c.u))
                dec, _ = model(g(c.z), reverse=True, length=ln)
            outs[fuse] = (z, ldj, dec)
    finally:
        o.FUSE_LAYERS = True
    assert torch.equal(outs[True][0], outs[False][0]) and torch.equal(outs[True][1], outs[False][1])
    assert torch.equal(outs[True][2], outs[False][2])


def test_empty_and_single_element_batches():
    """Edge shapes: B = 0 (nothing launched), B = 1, N = 1, D = 1."""
    z0 = torch.zeros(0, 5, 4, device="cuda")
    zf, lf = ops().affine_coupling(z0, torch.zeros(0, 5, 8, device="cuda"), torch.zeros(4, device="cuda"), g(O.channel_mask(4)))
    assert zf.shape == (0, 5, 4) and lf.shape == (0,)
    zf, lf, _ = ops().mixture_coupling(z0, torch.zeros(0, 5, 4 * 26, device="cuda"), g(O.channel_mask(4)), 8)
    assert zf.shape == (0, 5, 4)
    za, la = ops().actnorm(z0, torch.zeros(1, 1, 4, device="cuda"), torch.zeros(1, 1, 4, device="cuda"))
    assert za.shape == (0, 5, 4) and la.shape == (0,)
    gen = torch.Generator().manual_seed(9)
    for (B, N, D) in [(1, 1, 1), (1, 7, 1), (1, 1, 6), (2, 1, 3)]:
        z, nn_out = torch.randn(B, N, D, generator=gen), torch.randn(B, N, 2 * D, generator=gen)
        mask = O.channel_mask(D) if D > 1 else O.chess_mask()
        zo, lo = O.affine_coupling(z, nn_out, mask, None)
        zf, lf = ops().affine_coupling(g(z), g(nn_out), None, g(mask))
        close(zf, zo, **ELEM); loglik_close(lf, lo)
        K = 4
        nn_m = 0.5 * torch.randn(B, N, D * (2 + 3 * K), generator=gen)
        zo, lo, _ = O.mixture_coupling(z, nn_m, mask, K, None, None)
        zf, lf, _ = ops().mixture_coupling(g(z), g(nn_m), g(mask), K)
        close(zf, zo, **ELEM); loglik_close(lf, lo)


def test_hip_graph_replay_matches_eager():
    """A whole flow pass captured in a HIP graph replays the same kernels: identical outputs, no per-launch host cost."""
    from categoricalnf_amd.graphs import GraphedFlow
    torch.manual_seed(3)
    model, _ = _set_model(dict(set_size=16, transformer_layers=1, hidden=32, flows=2, K=8, D=4))
    model.cuda().eval()
    rng = np.random.RandomState(5)
    draw = lambda n: torch.from_numpy(np.stack([rng.permutation(16) for _ in range(n)])).long().cuda()
    B = 64
    ln = torch.full((B,), 16, dtype=torch.long, device="cuda")
    x1, x2 = draw(B), draw(B)
    u = torch.rand(B * 16, 1, 4, device="cuda")
    fwd = GraphedFlow(model, x1, reverse=False, length=ln, noise=u)
    for x in (x1, x2):
        z_g, ldj_g = [t.clone() for t in fwd(x)]
        with torch.no_grad():
            z_e, ldj_e = model(x, reverse=False, length=ln, noise=u)
        assert torch.equal(z_g, z_e) and torch.equal(ldj_g, ldj_e)
    inv = GraphedFlow(model, z_e, reverse=True, length=ln)
    dec_g = inv(z_e)[0].clone()
    with torch.no_grad():
        dec_e, _ = model(z_e, reverse=True, length=ln)
    assert torch.equal(dec_g, dec_e)


@pytest.mark.parametrize("c", load_cases("graph_node_flow"))
def test_graph_colouring_flow_golden(c):
    """BASELINE configs[2]: node-based GraphCNF on synthetic graphs (6..10 nodes; 10..20 nodes D=2 K=8 = tiny_3 sizes;
    25..50 nodes D=6 K=16 = large_3 sizes), 3 colours, RGCN-attention coupling sub-network, CDF regulariser — latents,
    log-det and decoded colours vs the reference, plus its own reversibility / permutation-equivariance checks with its
    tolerances."""
    from tests.test_host_cpu import _graph_model
    model = _graph_model(c.meta)
    model.load_state_dict({k[3:]: v for k, v in c.items() if k.startswith("sd_")})
    model.cuda().eval()
    with torch.no_grad():
        z, ldj = model(g(c.categ), adjacency=g(c.adjacency), reverse=False, length=g(c.length), noise=g(c.u))
        dec, _ = model(g(c.z), adjacency=g(c.adjacency), reverse=True, length=g(c.length))
    close(z, c.z, rtol=2e-4, atol=2e-4); loglik_close(ldj, c.ldj)
    assert torch.equal(dec.cpu(), c.decoded)
    assert c.meta["rev_ok"] and model.test_reversibility(g(c.categ), g(c.adjacency), g(c.length))
    assert c.meta["perm_ok"] and model.test_permutation(g(c.categ), g(c.adjacency), g(c.length))


@pytest.mark.parametrize("c", load_cases("language_model"))
def test_language_model_flow_golden(c):
    """configs[3] end to end: linear-flow encoder (ExtActNorm, 1x1 conv, affine coupling on a LinearNet) or mixture
    encoder, then ActNorm / 1x1 conv / autoregressive mixture-CDF couplings (K = 5 and the published K = 51) on the
    LSTM sub-network, variable lengths — latents and log-det vs the reference run with the same injected noise."""
    from tests.test_host_cpu import _language_model
    model = _language_model(c.meta)
    model.load_state_dict({k[3:]: v for k, v in c.items() if k.startswith("sd_")})
    model.cuda().eval()
    with torch.no_grad():
        z, ldj = model(g(c.tokens), reverse=False, length=g(c.length), noise=g(c.u))
    close(z, c.z, rtol=5e-4, atol=5e-4); loglik_close(ldj, c.ldj)
    # per-sample log-likelihood within 1e-4 relative (north_star)
    pad = (torch.arange(c.meta["T"])[None, :] < c.length[:, None]).float().unsqueeze(-1)
    nll_ref = O.nll_per_sample(c.z, c.ldj, c.length.float(), pad)
    _, nll = ops().prior_nll(z, ldj, g(c.length), g(pad))
    close(nll, nll_ref, rtol=1e-4, atol=1e-4)


def test_two_rank_data_parallel_gradients_match_single_process():
    """§8e/8f-1: one process per GPU, batch shards, DDP gradient all-reduce around the HIP backward kernels == the
    whole batch in one process (2 ranks sharing cuda:0 over gloo here; RCCL on a multi-GPU node via --backend nccl)."""
    import socket, subprocess, sys
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(root, "tools", "ddp_check.py"),
                        "--backend", "gloo", "--share-device"], capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert r.returncode == 0 and "DDP_CHECK OK" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])


def test_bench_self_launches_n_ranks():
    """a plain `python bench.py --gpus 2` (no torchrun environment) must start 2 ranks itself and
    report n_gpus == 2 (both ranks on cuda:0 over gloo here; RCCL with --backend nccl on a multi-GPU node)."""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--share-device", "--backend", "gloo",
                        "--steps", "5", "--warmup", "2", "--batch", "2048", "--prewarm-seconds", "0.2"],
                       capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and len(out["per_rank_elems_per_s"]) == 2 and out["allreduce_latency_us"] > 0
    assert out["value"] > 0 and out["roofline"]["kernel_ms"] > 0 and np.isfinite(out["mean_nll"])


def test_graphed_train_step_tracks_its_eager_twin():
    """graphs.GraphedTrainStep (back in round 3 with its root cause found): 400 replays of a captured training step of the
    set-modelling flow — forward, the HIP backward kernels, clipping, RAdam — stay as close to an eagerly trained copy fed
    the same data as a second eager copy does (tools/graph_train_soak.py; the acceptance run is 3000 replays,
    profiles/r03_graph_train_soak.txt), and the same run WITHOUT capture_safe_linear leaves it (the memset-node fault)."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "tools", "graph_train_soak.py"), "--steps", "400", "--check_every", "100", "--flows", "4", "--hidden", "128"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0 and "SOAK OK" in r.stdout, (r.stdout[-3000:], r.stderr[-2000:])
    r = subprocess.run(cmd + ["--plain_linear"], capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode != 0 and "SOAK FAILED" in r.stdout, (r.stdout[-3000:], r.stderr[-2000:])
    # and GraphedTrainStep refuses a step that still holds a memset node instead of replaying it wrongly
    from categoricalnf_amd.graphs import GraphedTrainStep
    lin = torch.nn.Linear(256, 256).cuda()
    x, w = torch.randn(1024, 256, device="cuda"), torch.randn(1024, 256, device="cuda")

    def with_a_two_pass_sum():                       # dY.sum(0) over 1024 rows: reduce_kernel behind a memset of its semaphores
        y = torch.tanh(x @ lin.weight.t())
        return torch.autograd.grad((y * w).sum(), [lin.weight])[0].sum(0) + (y * w).sum(0)

    def staged():
        from categoricalnf_amd.graphs import _column_sums
        y = torch.tanh(x @ lin.weight.t())
        return _column_sums(y * w)
    ok = GraphedTrainStep(staged, torch.device("cuda", 0))
    assert ok.nodes is not None and ok.nodes.get("memset", 0) == 0 and ok.nodes.get("kernel", 0) > 0, ok.nodes
    with pytest.raises(RuntimeError, match="memset node"):
        GraphedTrainStep(with_a_two_pass_sum, torch.device("cuda", 0))


def test_rccl_backend_initialises_and_reduces_on_this_box():
    """RCCL itself (backend "nccl"), as far as a 1-GPU box allows: a one-rank process group on cuda:0 — communicator
    set-up, the all-reduce of the (sum NLL, count) pair this library's jobs perform, a barrier — in a process of its own
    (several ranks need several GPUs: RCCL refuses two ranks on one device; the N-rank paths run over gloo above)."""
    import subprocess, sys
    code = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29677", rank=0, world_size=1)
torch.cuda.set_device(0)
from categoricalnf_amd.distributed import allreduce_nll
t = torch.tensor([12.5, 4.0], dtype=torch.float64, device="cuda")
dist.all_reduce(t)
dist.barrier()
mean, bpd = allreduce_nll(t.clone())
assert t.tolist() == [12.5, 4.0] and abs(mean - 3.125) < 1e-12, (t, mean)
print("RCCL OK", dist.get_backend(), torch.cuda.nccl.version())
dist.destroy_process_group()
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, "-c", code, root], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "RCCL OK nccl" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])


def test_bench_rehearses_eight_ranks_on_one_device():
    """The driver's first real 8-GPU run must not be the first 8-rank run: `python bench.py --gpus 8` as 8 processes
    sharing cuda:0 over gloo — rendezvous, per-rank rates, the all-reduce of the batch sums as the closing barrier —
    and the refusal paths (a WORLD_SIZE that contradicts --gpus; 8 devices asked for where 1 is visible)."""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--share-device", "--backend", "gloo",
                        "--steps", "5", "--warmup", "2", "--batch", "1024", "--prewarm-seconds", "0.1"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and len(out["per_rank_elems_per_s"]) == 8 and out["allreduce_latency_us"] > 0
    assert out["scaling"] == "weak" and out["value"] > 0 and np.isfinite(out["mean_nll"])
    assert abs(out["value"] - 8 * 1024 * 64 * 6 * 5 / (out["ms_per_step"] * 5e-3)) < 1e-6 * out["value"]
    if torch.cuda.device_count() < 8:
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1"],
                           capture_output=True, text=True, timeout=300, env=env, cwd=root)
        assert r.returncode != 0 and "HIP device(s) visible" in (r.stdout + r.stderr)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, timeout=300, cwd=root,
                       env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"))
    assert r.returncode != 0 and "does not match WORLD_SIZE" in (r.stdout + r.stderr)


def test_scale_sweep_script_rehearses_the_drivers_scaling_run():
    """tools/scale_sweep.sh — bench.py at N = 1, 2, 8 ranks back to back, one JSON line each, the efficiency table — with all
    ranks on cuda:0 over gloo: the N-rank code path of the driver's SCALE run (rank start-up, per-rank kernel times and
    rates gathered to rank 0, the all-reduce as the closing barrier), not the interconnect."""
    import json, subprocess, tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    with tempfile.TemporaryDirectory() as out:
        r = subprocess.run(["bash", os.path.join(root, "tools", "scale_sweep.sh"), out, "1", "2", "8"], capture_output=True, text=True, timeout=1500, cwd=root,
                           env=dict(env, SHARE_DEVICE="1", BENCH_FLAGS="--steps 5 --warmup 2 --batch 1024 --prewarm-seconds 0.1 --no-cpu-baseline --no-mixture"))
        assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
        rows = [json.loads(l) for l in open(os.path.join(out, "scale.jsonl"))]
        table = json.load(open(os.path.join(out, "scale.json")))          # the table for programs
    assert [x["n_gpus"] for x in rows] == [1, 2, 8], r.stdout[-2000:]
    assert [x["n_gpus"] for x in table["rows"]] == [1, 2, 8] and table["scaling"] == "weak"
    assert all(x["share_device"] for x in table["rows"]) and table["rows"][1]["backend"] == "gloo" and table["rows"][0]["efficiency"] == 1.0
    for x in rows:
        assert len(x["per_rank_elems_per_s"]) == x["n_gpus"] == len(x["roofline"]["per_rank_kernel_ms"])
        assert all(k > 0 for k in x["roofline"]["per_rank_kernel_ms"])
        assert (x["allreduce_latency_us"] is None) == (x["n_gpus"] == 1)
    assert "efficiency" in r.stdout and len([l for l in r.stdout.splitlines() if l.strip() and l.split()[0] in ("1", "2", "8")]) == 3, r.stdout


def test_actnorm_data_init_statistics_meet_across_ranks():
    """distributed.sync_data_init(): ranks holding different shards of the initialisation batch all-reduce ActNorm's
    per-channel sums and end up with the whole batch's bias / scales; 2 and 3 ranks on cuda:0."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for world, port in ((2, "29641"), (3, "29642")):
        r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
                            "--master-addr", "127.0.0.1", "--master-port", port, os.path.join(root, "tools", "init_sync_check.py"),
                            "--backend", "gloo", "--share-device"], capture_output=True, text=True, timeout=600, cwd=root,
                           env=dict(os.environ, OMP_NUM_THREADS="1"))
        assert r.returncode == 0 and r.stdout.count("INIT_SYNC OK") == world, (r.stdout[-2000:], r.stderr[-2000:])


def test_dispatch_bound_kernel_timing_matches_event_brackets():
    """cnf_prof_arm / cnf_prof_collect (bench.py's roofline clock): the dispatch-bound duration of a big launch is
    positive, below a marker-bracketed measurement of the same launch and within 2x of it."""
    import ctypes
    lib = _lib.load()
    B, N, D = 16384, 64, 6
    z, nn_out = torch.randn(B, N, D, device="cuda"), torch.randn(B, N, 2 * D, device="cuda")
    mask = torch.tensor([[1., 1., 1., 0., 0., 0.]], device="cuda")
    zo, lo = torch.empty_like(z), torch.empty(B, device="cuda")
    k = ops().affine_coupling_launch(z, nn_out, torch.zeros(D, device="cuda"), mask, zo, lo)
    for _ in range(20):
        k()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record(); k(); b.record()
    lib.cnf_prof_arm(3)
    for _ in range(5):
        k()
    buf = (ctypes.c_float * 8)()
    n = lib.cnf_prof_collect(buf, 8)
    torch.cuda.synchronize()
    assert n == 3
    bracket = a.elapsed_time(b)
    for i in range(n):
        assert 0.005 < buf[i] < 0.2, list(buf)
    assert min(buf[:n]) <= bracket * 1.05
    assert lib.cnf_prof_collect(buf, 8) == 0


def test_set_modelling_driver_trains_checkpoints_and_reloads(tmp_path):
    """The host template (run_set_modeling): a short run on set summation lowers the validation bits/dim well below
    the uniform 4 bpd, writes a reference-format checkpoint, and --only_eval from that checkpoint reproduces the
    stored validation NLL."""
    from categoricalnf_amd.experiments import run_set_modeling as R
    small = ["--dataset", "summation", "--coupling_hidden_size", "32", "--coupling_hidden_layers", "1", "--coupling_num_flows", "2",
             "--checkpoint_path", str(tmp_path), "--print_freq", "1000000"]
    out = R.main(small + ["--max_iterations", "400", "--eval_freq", "400", "--batch_size", "128", "--learning_rate", "2e-3"])
    assert np.isfinite(out["val_bpd"]) and out["val_bpd"] < 3.6, out
    assert out["best_file"] and os.path.isfile(out["best_file"])
    again = R.main(small + ["--only_eval"])
    assert abs(again["val_bpd"] - out["val_bpd"]) < 2e-3, (again, out)
    # param_config.pik (general/train.py:428-432): evaluation needs the directory only, the model is rebuilt from it
    assert os.path.isfile(os.path.join(str(tmp_path), R.PARAM_CONFIG_FILE))
    assert R.load_args(out["best_file"]).coupling_hidden_size == 32
    bare = R.main(["--checkpoint_path", str(tmp_path), "--only_eval", "--load_best_model"])
    assert abs(bare["val_bpd"] - out["val_bpd"]) < 2e-3, (bare, out)



def test_set_modelling_driver_with_the_captured_training_step(tmp_path):
    """run_set_modeling --graph_step (round 3): the training step replayed from a HIP graph learns like the eager loop
    (validation bits/dim well below the uniform 4 bpd after 400 steps), its checkpoint — written while the weights moved
    under replays, so the eval-mode caches of the 1x1 convolutions must have been dropped — evaluates to the stored
    validation NLL in a fresh eager process state, and a resumed run carries on from it."""
    from categoricalnf_amd.experiments import run_set_modeling as R
    small = ["--dataset", "summation", "--coupling_hidden_size", "32", "--coupling_hidden_layers", "1", "--coupling_num_flows", "2",
             "--checkpoint_path", str(tmp_path), "--print_freq", "1000000", "--batch_size", "128", "--learning_rate", "2e-3"]
    out = R.main(small + ["--max_iterations", "400", "--eval_freq", "200", "--save_freq", "400", "--graph_step"])
    assert np.isfinite(out["val_bpd"]) and out["val_bpd"] < 3.6, out
    again = R.main(small + ["--only_eval"])
    assert abs(again["val_bpd"] - out["val_bpd"]) < 2e-3, (again, out)
    more = R.main(small + ["--max_iterations", "500", "--eval_freq", "100", "--graph_step"])
    assert np.isfinite(more["val_bpd"]) and more["val_bpd"] < out["val_bpd"] + 0.05, (more, out)


def test_backward_with_non_contiguous_upstream_gradients():
    """Both upstream gradients of a layer arrive non-contiguous (a transposed view and the expanded gradient of a
    mean): each is copied for the kernel and both copies must stay alive until the launch (they once could be handed
    the same block).  Affine coupling and the static (s, t) split API against the oracle's autograd."""
    from categoricalnf_amd import functional as Fn
    gen = torch.Generator().manual_seed(21)
    B, N, D = 37, 11, 6
    z = torch.randn(B, N, D, generator=gen)
    nn_out = 0.5 * torch.randn(B, N, 2 * D, generator=gen)
    sf = 0.2 * torch.randn(D, generator=gen)
    w = torch.randn(B, D, N, generator=gen)
    mask = O.channel_mask(D)

    def loss_of(zo, lo):
        return (zo.transpose(1, 2) * (w.to(zo.device))).sum() + 3.0 * lo.mean()
    zc, nc, sc = (t.clone().requires_grad_() for t in (z, nn_out, sf))
    loss_of(*O.affine_coupling(zc, nc, mask, sc)).backward()
    zg, ng, sg = (g(t).requires_grad_() for t in (z, nn_out, sf))
    loss_of(*Fn.AffineCouplingFn.apply(zg, ng, sg, None, g(mask), False)).backward()
    close(zg.grad, zc.grad, **GRAD); close(ng.grad, nc.grad, **GRAD); close(sg.grad, sc.grad, **GRAD)
    # static API: s and t come back separately, their gradients are both views of one transposed tensor
    nc2, sc2 = nn_out.clone().requires_grad_(), sf.clone().requires_grad_()
    s_o, t_o = O.affine_params(nc2, O.expand_mask(mask, z), sc2)
    ws_, wt_ = torch.randn(B, D, N, generator=gen), torch.randn(B, D, N, generator=gen)
    ((s_o.transpose(1, 2) * ws_).sum() + (t_o.transpose(1, 2) * wt_).sum()).backward()
    ng2, sg2 = g(nn_out).requires_grad_(), g(sf).requires_grad_()
    s_g, t_g = Fn.AffineParamsFn.apply(ng2, sg2, g(mask))
    ((s_g.transpose(1, 2) * g(ws_)).sum() + (t_g.transpose(1, 2) * g(wt_)).sum()).backward()
    close(ng2.grad, nc2.grad, **GRAD); close(sg2.grad, sc2.grad, **GRAD)


def test_c_abi_without_torch(tmp_path):
    """The boundary is a plain C ABI: tests/abi/abi_roundtrip.cpp (hipMalloc'ed buffers, no torch, no Python) is
    compiled against include/cnf_hip.h, linked to libcnf_hip.so and run in its own process; it checks the fused
    coupling + NLL kernel, the batch sum and the inverse against a scalar fp64 loop, the mixture coupling (both parameter layouts,
    forward, inverse and the fp32 backward: compact rows bit for bit, zero blocks, a central difference) and the encoder
    forward + decode."""
    import shutil, subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib_dir = os.path.join(root, "categoricalnf_amd", "lib")
    exe = str(tmp_path / "abi_roundtrip")
    build = subprocess.run([hipcc, "--offload-arch=gfx950", "-O2", os.path.join(root, "tests", "abi", "abi_roundtrip.cpp"),
                            "-I", os.path.join(root, "include"), "-L", lib_dir, "-lcnf_hip", "-Wl,-rpath," + lib_dir, "-o", exe],
                           capture_output=True, text=True, timeout=600)
    assert build.returncode == 0, build.stderr[-2000:]
    run = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert run.returncode == 0 and "ABI_C OK" in run.stdout, (run.stdout[-1000:], run.stderr[-1000:])


@pytest.mark.parametrize("fixture", ["graph_cnf", "graph_cnf_zinc"])
def test_molecule_graph_cnf_three_stage_flow_golden(fixture):
    """BASELINE configs[4]: the three-stage GraphCNF (nodes / edge attributes / virtual edges) assembled on the HIP
    layers against the REFERENCE's assembly of its own layers, sub-network outputs injected on both sides
    (oracle/gen_golden.py::gen_graph_cnf): latents after every stage, the log-det of every layer, the final per-sample
    log-likelihood terms within 1e-4 relative, and the sampling pass's decoded node types and adjacency bit-exact.
    `graph_cnf_zinc` is the configuration at its real sizes (experiments/molecule_generation/README.md:19-30,
    zinc250k.py:150-164): 38 nodes, 703 node pairs, D = 6 / 2, K = 16 / 8, 9 node and 3 edge types, 4 + 6 + 6 coupling
    layers, graphs of different sizes."""
    from tests.test_host_cpu import _graph_cnf_model
    c = load_cases(fixture)[0]
    model = _graph_cnf_model(c).cuda()
    for layer in list(model.step1_flows) + list(model.step2_flows) + list(model.step3_flows):
        if hasattr(layer, "nn"):
            v = layer.nn.value
            layer.nn.value = tuple(t.cuda() for t in v) if isinstance(v, tuple) else v.cuda()
    stage = {}
    hooks = [model.step1_flows[-1].register_forward_hook(lambda m, i, o: stage.__setitem__("s1", o)),
             model.step2_flows[-1].register_forward_hook(lambda m, i, o: stage.__setitem__("s2", o)),
             model.step3_flows[-1].register_forward_hook(lambda m, i, o: stage.__setitem__("s3", o))]
    with torch.no_grad():
        z, ldj, per_layer = model(g(c.nodes), adjacency=g(c.adjacency), reverse=False, get_ldj_per_layer=True, length=g(c.length),
                                  noise=(g(c.u_nodes), g(c.u_attr), g(c.u_virtual)))
    for h in hooks:
        h.remove()
    close(stage["s1"][0], c.s1_z, **ELEM)
    close(stage["s2"][0], c.s2_z_nodes, **ELEM); close(stage["s2"][1], c.s2_z_edges, **ELEM)
    close(stage["s3"][0], c.s3_z_nodes, **ELEM); close(stage["s3"][1], c.s3_z_edges, **ELEM)
    close(z, c.z, **ELEM)
    assert len(per_layer) == c.layer_ldj.shape[0]
    for got, ref in zip(per_layer, c.layer_ldj):
        if torch.isnan(ref).all():
            assert isinstance(got, dict) and len(got) == 0          # an encoder in eval mode reports nothing
            continue
        val = got if isinstance(got, torch.Tensor) else (got["ldj"] if "ldj" in got else list(got.values())[0])
        loglik_close(val, ref)
    loglik_close(ldj, c.ldj)
    with torch.no_grad():
        (nodes, adjacency), ldj_rev = model(g(c.z), reverse=True, length=g(c.length), edge_latents=g(c.edge_latents))
    assert torch.equal(nodes.cpu(), c.dec_nodes) and torch.equal(adjacency.cpu(), c.dec_adjacency)
    loglik_close(ldj_rev, c.ldj_rev)


def test_molecule_graph_cnf_trains_and_samples_end_to_end_at_zinc_sizes():
    """configs[4] executed END TO END on the device at its real sizes (38 nodes, 703 pairs, D = 6 / 2, K = 16 / 8, 9 node
    types): the three-stage GraphCNF on the HIP layers with the reference's own sub-network architectures (RGCN, Edge-GNN:
    layers/networks/edge_gnn.py, pinned by tests/golden/edge_gnn.npz) — data-dependent init, 30 training
    steps through every backward kernel of the path (loss falls), evaluation, one sampling pass (tools/molecule_train_probe.py)."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "molecule_train_probe.py"), "--steps", "30", "--batch", "16",
                        "--flows", "2,2,2", "--hidden_nodes", "64", "--hidden_edges", "32", "--layers", "2", "--graphs", "256"],
                       capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0 and "MOLECULE PROBE OK" in r.stdout, (r.stdout[-3000:], r.stderr[-3000:])


@pytest.mark.parametrize("B,N,D,C", [(9, 16, 4, 16), (5, 33, 3, 51), (3, 20, 10, 700), (2, 12, 6, 3000), (4, 7, 1, 2), (70, 5, 2, 1200)])
def test_class_tiled_encoder_kernels(B, N, D, C):
    """cnf_encoder_forward_tiled / cnf_encoder_decode_tiled (vocabularies beyond the LDS-resident class table):
    equal to the LDS-resident kernels where both apply, equal to the oracle at 700 / 1200 / 3000 classes — latents,
    per-sample log-det (1e-4 relative), class posterior, decoded indices bit-exact."""
    gen = torch.Generator().manual_seed(B * 31 + C)
    cat = torch.randint(0, C, (B, N), generator=gen)
    table = torch.randn(C, 2 * D, generator=gen)
    prior = torch.log_softmax(torch.randn(C, generator=gen), 0)
    eps = O.logistic_from_uniform(torch.rand(B * N, 1, D, generator=gen))
    ln = torch.randint(max(1, N // 2), N + 1, (B,), generator=gen)
    pad = O.length_mask(ln, N)
    ldj0 = torch.randn(B, generator=gen)
    zt, lt, ct = ops().encoder_forward(g(cat), g(eps), g(table), g(prior), beta=1.5, channel_padding_mask=g(pad), ldj=g(ldj0),
                                       want_class_prob=True, tiled=True)
    zo, lo, co = O.encoder_forward(cat, eps, table, prior, beta=1.5, channel_padding_mask=pad)
    close(zt, zo, **ELEM); loglik_close(lt, lo + ldj0)
    close(ct, co.reshape(-1), rtol=1e-4, atol=1e-4)
    dt = ops().encoder_decode(zt, g(table), g(prior), tiled=True)
    assert torch.equal(dt.cpu(), O.encoder_decode(zt.cpu(), table, prior)[0])
    if ops().encoder_fused_supported(C, D):
        zf, lf, cf = ops().encoder_forward(g(cat), g(eps), g(table), g(prior), beta=1.5, channel_padding_mask=g(pad), ldj=g(ldj0),
                                           want_class_prob=True, tiled=False)
        close(zt, zf, rtol=1e-6, atol=1e-6); close(lt, lf, rtol=1e-5, atol=1e-4); close(ct, cf, rtol=1e-5, atol=1e-5)
        assert torch.equal(dt, ops().encoder_decode(zt, g(table), g(prior), tiled=False))
    else:
        # the automatic choice is the tiled kernel
        z2, l2, _ = ops().encoder_forward(g(cat), g(eps), g(table), g(prior), beta=1.5, channel_padding_mask=g(pad), ldj=g(ldj0))
        assert torch.equal(z2, zt) and torch.equal(l2, lt)
    ops().check_flags(torch.device("cuda"), "tiled encoder")


def test_large_vocabulary_encoder_module_runs_on_the_tiled_kernels():
    """LinearCategoricalEncoding with 2000 classes: evaluation passes go through the class-tiled kernels (no [T*C,1,D]
    tensor), decode(encode(x)) agrees with the composed path's posterior arg-max, an out-of-range index raises."""
    from categoricalnf_amd.layers.categorical_encoding.linear_encoding import LinearCategoricalEncoding
    torch.manual_seed(3)
    enc = LinearCategoricalEncoding(num_dimensions=6, flow_config={"num_flows": 0}, vocab_size=2000).cuda().eval()
    for p in enc.parameters():                       # the conditioner's last layer starts at zero: make classes differ
        p.data.normal_(0.0, 0.5)
    assert enc._is_mixture_model() and not ops().encoder_fused_supported(2000, 6)
    x = torch.randint(0, 2000, (4, 24), device="cuda")
    u = torch.rand(4 * 24, 1, 6, device="cuda")
    with torch.no_grad():
        z, ldj, _ = enc(x, noise=u)
        zc, ldjc, _ = enc._forward_composed(x, 1, None, u)
        dec, _, _ = enc(z, reverse=True)
        scores = enc._all_class_scores(z.reshape(-1, 1, 6))
    close(z, zc, **ELEM); loglik_close(ldj, ldjc)
    best = scores.max(dim=-1).values
    picked = scores.gather(1, dec.reshape(-1, 1)).squeeze(1)
    assert (dec.reshape(-1) == scores.argmax(dim=-1)).float().mean() > 0.98
    assert float((best - picked).max()) < 1e-3       # any disagreement is a tie at fp32 resolution
    with pytest.raises(AssertionError):
        bad = x.clone(); bad[0, 0] = 2000
        with torch.no_grad():
            enc(bad, noise=u)
        ops().check_flags(torch.device("cuda"), "range")


@pytest.mark.parametrize("cpl", [1, 2, 4])
def test_stream_probe_moves_what_it_says(cpl):
    """bench.py's measured-ceiling kernel: out = a + b_even * b_odd over n elements, ragged last block."""
    n = 4 * (256 * 7 + 33)
    a, b = torch.randn(n, device="cuda"), torch.randn(2 * n, device="cuda")
    out = torch.zeros(n, device="cuda")
    ops()._launch(a.device, "cnf_stream_probe", a.data_ptr(), b.data_ptr(), out.data_ptr(), n, cpl, ops()._stream(a.device))
    assert torch.equal(out, a + b[0::2] * b[1::2])


def test_graph_colouring_generation_and_validity_on_device():
    """task.py:170-215 on the drop-in: latents from the prior -> flow backwards -> decoded colours -> validity counted on
    the device; the count equals the reference's per-graph numpy rule (restated here) on the same samples, and the
    per-node NLL helper equals the oracle's assembly."""
    from tests.test_host_cpu import _graph_model
    from categoricalnf_amd.experiments.graph_coloring import sample_colorings, generation_validity, flow_nll
    from categoricalnf_amd.experiments.graph_coloring_data import GraphColoringDataset, coloring_validity
    from categoricalnf_amd.layers.flows.distributions import LogisticDistribution
    c = load_cases("graph_node_flow")[0]
    model = _graph_model(c.meta)
    model.load_state_dict({k[3:]: v for k, v in c.items() if k.startswith("sd_")})
    model.cuda().eval()
    prior = LogisticDistribution(mu=0.0, sigma=1.0)
    adj, ln = g(c.adjacency), g(c.length)
    u = torch.rand(adj.shape[0], adj.shape[1], model.embed_dim, device="cuda")
    nodes = sample_colorings(model, prior, adj, ln, noise=u)
    assert nodes.dtype == torch.int64 and nodes.shape == adj.shape[:2] and int(nodes.min()) >= 0 and int(nodes.max()) < 3
    valid = coloring_validity(nodes, adj, ln)
    n_, a_, l_ = nodes.cpu().numpy(), c.adjacency.numpy(), c.length.numpy()
    ref = [bool(np.all((n_[i, :l_[i]] + 1)[:, None] != a_[i, :l_[i], :l_[i]] * (n_[i, :l_[i]] + 1)[None, :])) for i in range(len(l_))]
    assert valid.cpu().tolist() == ref
    torch.manual_seed(0)
    out = generation_validity(model, prior, [(None, adj, ln), (None, adj[:3], ln[:3])], GraphColoringDataset)
    assert out["num_graphs"] == adj.shape[0] + 3 and 0.0 <= out["valid_ratio"] <= 1.0
    with torch.no_grad():
        nll, per_layer = flow_nll(model, prior, g(c.categ), adj, ln)
    assert nll.shape == (adj.shape[0],) and torch.isfinite(nll).all() and len(per_layer) == len(model.flow_layers)


def _encoder_grad_inputs(B, N, D, C, seed):
    gen = torch.Generator().manual_seed(seed)
    cat = torch.randint(0, C, (B, N), generator=gen)
    table = 0.7 * torch.randn(C, 2 * D, generator=gen)
    prior = torch.log_softmax(torch.randn(C, generator=gen), 0)
    eps = O.logistic_from_uniform(torch.rand(B * N, 1, D, generator=gen))
    ln = torch.randint(max(1, N // 2), N + 1, (B,), generator=gen)
    pad = O.length_mask(ln, N)
    wz, wl = torch.randn(B, N, D, generator=gen), torch.randn(B, generator=gen)
    return cat, table, prior, eps, pad, wz, wl


@pytest.mark.parametrize("B,N,D,C", [(6, 9, 6, 300), (3, 40, 4, 1500), (5, 7, 3, 16), (2, 300, 10, 120), (4, 5, 1, 2)])
def test_class_tiled_encoder_backward(B, N, D, C):
    """cnf_encoder_forward_bwd_cpl / _tiled (the pair kernel or the token-lane + class-lane passes, by shape): d loss / d class
    table against autograd through the oracle on the CPU, bit-reproducible, whichever forward kernel produced the latents."""
    from categoricalnf_amd import functional as Fn
    cat, table, prior, eps, pad, wz, wl = _encoder_grad_inputs(B, N, D, C, seed=B * 7 + C)
    tc = table.clone().requires_grad_()
    zo, lo, _ = O.encoder_forward(cat, eps, tc, prior, beta=1.3, channel_padding_mask=pad)
    ((zo * wz).sum() + (lo * wl).sum()).backward()

    def run(tiled):
        tg = g(table).requires_grad_()
        z, ldj, _ = Fn.EncoderForwardFn.apply(tg, g(cat), g(eps), g(prior), g(pad), 1.3, False, tiled)
        ((z * g(wz)).sum() + (ldj * g(wl)).sum()).backward()
        return tg.grad
    gt = run(True)
    scale = float(tc.grad.abs().max())
    close(gt, tc.grad, rtol=2e-3, atol=2e-4 * max(scale, 1.0))
    assert torch.equal(gt, run(True))                                   # fixed summation order
    # the forward on the LDS-resident kernel where it applies (tiled = False / None): the same backward, reproducible too
    if ops().encoder_fused_supported(C, D):
        close(gt, run(False), rtol=1e-3, atol=1e-4 * max(scale, 1.0))
        assert torch.equal(run(None), run(None))
    # only one of the two upstream gradients
    tg = g(table).requires_grad_()
    z, ldj, _ = Fn.EncoderForwardFn.apply(tg, g(cat), g(eps), g(prior), None, 1.0, False, True)
    (ldj * g(wl)).sum().backward()
    tc2 = table.clone().requires_grad_()
    (O.encoder_forward(cat, eps, tc2, prior)[1] * wl).sum().backward()
    close(tg.grad, tc2.grad, rtol=2e-3, atol=2e-4 * max(float(tc2.grad.abs().max()), 1.0))


def test_large_vocabulary_encoder_trains_on_the_tiled_kernels():
    """2000 classes, D = 6: the module's parameter gradients through the tiled forward + backward equal those of the
    composed path (layer kernels over the expanded [T*C, 1, D] tensor, the reference's own route)."""
    from categoricalnf_amd.layers.categorical_encoding.linear_encoding import LinearCategoricalEncoding
    torch.manual_seed(5)
    enc = LinearCategoricalEncoding(num_dimensions=6, flow_config={"num_flows": 0}, vocab_size=2000).cuda().train()
    for p in enc.parameters():
        p.data.normal_(0.0, 0.3)
    x = torch.randint(0, 2000, (3, 20), device="cuda")
    u = torch.rand(3 * 20, 1, 6, device="cuda")
    wz, wl = torch.randn(3, 20, 6, device="cuda"), torch.randn(3, device="cuda")
    z, ldj, det = enc(x, noise=u, beta=0.8)
    assert set(det) == {"avg_token_prob", "avg_token_bpd", "z_min", "z_max", "z_std"}
    ((z * wz).sum() + (ldj * wl).sum()).backward()
    got = {n: p.grad.clone() for n, p in enc.named_parameters()}
    enc.zero_grad()
    zc, ldjc, _ = enc._forward_composed(x, 0.8, None, u)
    ((zc * wz).sum() + (ldjc * wl).sum()).backward()
    close(z, zc, **ELEM); loglik_close(ldj, ldjc)
    for n, p in enc.named_parameters():
        close(got[n], p.grad, rtol=3e-3, atol=3e-4 * max(float(p.grad.abs().max()), 1.0))


def test_class_split_encoder_results_do_not_depend_on_the_batch():
    """Above 1024 classes the class range is split over workgroups; the number of splits is a function of C only, so
    a sample's latents, log-det, decoded class and table gradient contribution are the same whatever batch it sits in."""
    B, N, D, C = 12, 19, 4, 2500
    cat, table, prior, eps, pad, _, _ = _encoder_grad_inputs(B, N, D, C, seed=77)
    z, ldj, cpl = ops().encoder_forward(g(cat), g(eps), g(table), g(prior), channel_padding_mask=g(pad), want_class_prob=True)
    h = 5
    zh, ldjh, cplh = ops().encoder_forward(g(cat[:h]), g(eps[:h * N]), g(table), g(prior), channel_padding_mask=g(pad[:h]),
                                           want_class_prob=True)
    assert torch.equal(zh, z[:h]) and torch.equal(ldjh, ldj[:h]) and torch.equal(cplh, cpl[:h * N])
    dec = ops().encoder_decode(z, g(table), g(prior))
    assert torch.equal(ops().encoder_decode(z[:h], g(table), g(prior)), dec[:h])
    assert torch.equal(dec.cpu(), O.encoder_decode(z.cpu(), table, prior)[0])


@pytest.mark.parametrize("c", load_cases("encoder_large_vocab"))
def test_encoder_large_vocab_golden(c):
    """The REFERENCE's encoder at 300 / 1100 / 1300 classes (tests/golden/encoder_large_vocab.npz): the drop-in module on
    the kernels its vocabulary selects — 300 classes at D = 6 still fit the LDS-resident forward (47 KB of class table,
    the faster kernel there since round 3), 1100 / 1300 run on the class-tiled kernels with class splits — latents,
    log-det (1e-4 relative), decoded classes bit-exact, and the parameter gradients of the reference's autograd (the
    class-tiled backward at every size)."""
    from categoricalnf_amd.layers.categorical_encoding.linear_encoding import LinearCategoricalEncoding
    m = c.meta
    assert ops().encoder_prefers_tiled_forward(m["C"], m["D"]) == (m["C"] > 1000)
    enc = LinearCategoricalEncoding(num_dimensions=m["D"], flow_config={"num_flows": 0}, vocab_size=m["C"], default_embed_layer_dims=8)
    enc.load_state_dict({k[3:]: v for k, v in c.items() if k.startswith("sd_")})
    enc.cuda().eval()
    kw = dict(channel_padding_mask=g(c.pad)) if m["padded"] else {}
    z, ldj, _ = enc(g(c.categ), reverse=False, beta=m["beta"], noise=g(c.u), **kw)
    close(z, c.z, **ELEM); loglik_close(ldj, c.ldj)
    ((z * g(c.wz)).sum() + (ldj * g(c.wl)).sum()).backward()
    for name, p in enc.named_parameters():
        ref = c["gp_" + name]
        close(p.grad, ref, rtol=2e-3, atol=2e-4 * max(float(ref.abs().max()), 1.0))
    with torch.no_grad():
        dec, _, _ = enc(g(c.z), reverse=True)
        dec_probe, _, _ = enc(g(c.z_probe), reverse=True)
    assert torch.equal(dec.cpu(), c.decoded) and torch.equal(dec_probe.cpu(), c.decoded_probe)
    # the kernels directly on the reference's class table
    zt, lt, _ = ops().encoder_forward(g(c.categ), ops().logistic_from_uniform(g(c.u)), g(c.table), g(c.category_prior),
                                      beta=m["beta"], channel_padding_mask=kw.get("channel_padding_mask"))
    close(zt, c.z, **ELEM); loglik_close(lt, c.ldj)


def test_graph_colouring_driver_trains_and_samples(tmp_path):
    """The host template for the second experiment (run_graph_coloring): synthetic planted-colouring graphs in the
    reference's file format, bucketed batches, beta schedule, per-node NLL; a short run lowers the bits per node, the
    permutation / reversibility checks of the reference pass at start-up, sampled colourings are scored for validity, and
    --only_eval from the checkpoint reproduces the validation figure."""
    from categoricalnf_amd.experiments import run_graph_coloring as R
    common = ["--dataset", "tiny_3", "--data_root", str(tmp_path / "data"), "--generate_data", "--num_graphs", "3000",
              "--coupling_hidden_size", "32", "--coupling_hidden_layers", "2", "--coupling_num_flows", "2",
              "--checkpoint_path", str(tmp_path / "ck"), "--print_freq", "1000000", "--eval_batch_size", "256"]
    assert abs(R.beta_at(R.parse(common), 5000) - 1.5) < 1e-9          # parameter_scheduler.py:120-121 at one step size
    out = R.main(common + ["--max_iterations", "300", "--eval_freq", "300", "--batch_size", "128", "--learning_rate", "2e-3"])
    assert np.isfinite(out["val_bpd"]) and out["val_bpd"] < np.log2(3) - 0.05, out       # below the uniform 1.585 bits
    assert 0.0 <= out["val_valid_ratio"] <= 1.0 and out["best_file"] and os.path.isfile(out["best_file"])
    again = R.main(common + ["--only_eval"])
    # the encoder draws fresh noise per evaluation: 300 validation graphs give the figure to a few hundredths of a bit
    assert abs(again["val_bpd"] - out["val_bpd"]) < 0.06, (again, out)


def test_graph_colouring_driver_with_the_captured_training_step(tmp_path):
    """run_graph_coloring --graph_step (round 4): full-width batches, beta and the learning rate in device scalars, the step
    replayed from a HIP graph without a memset node (RelationGraphAttention takes its attention logits by contraction) — it
    learns like the eager loop and its checkpoint evaluates to the same figure in eager mode."""
    from categoricalnf_amd.experiments import run_graph_coloring as R
    common = ["--dataset", "tiny_3", "--data_root", str(tmp_path / "data"), "--generate_data", "--num_graphs", "3000",
              "--coupling_hidden_size", "32", "--coupling_hidden_layers", "2", "--coupling_num_flows", "2",
              "--checkpoint_path", str(tmp_path / "ck"), "--print_freq", "1000000", "--eval_batch_size", "256"]
    out = R.main(common + ["--max_iterations", "300", "--eval_freq", "300", "--batch_size", "128", "--learning_rate", "2e-3", "--graph_step"])
    assert np.isfinite(out["val_bpd"]) and out["val_bpd"] < np.log2(3) - 0.05, out
    again = R.main(common + ["--only_eval"])
    assert abs(again["val_bpd"] - out["val_bpd"]) < 0.06, (again, out)


def test_language_modelling_driver_with_the_captured_training_step(tmp_path):
    """run_language_modeling --graph_step (round 4): the LSTM on PyTorch's native path (MIOpen's RNN aborts under a stream
    capture), beta and the learning rate in device scalars; the replayed step beats the context-free optimum like the eager
    loop does, and the checkpoint reproduces the figure in eager mode."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    common = ["--vocab_size", "9", "--source_alpha", "0.3", "--max_seq_len", "32", "--batch_size", "64", "--num_val", "256",
              "--coupling_hidden_size", "128", "--coupling_hidden_layers", "1", "--coupling_num_mixtures", "9",
              "--encoding_dim", "3", "--variable_length", "--checkpoint_path", str(tmp_path / "lm")]
    # in a process of its own: the mode switches torch.backends.cudnn off for the process
    code = ("import sys, json; sys.path.insert(0, %r); from categoricalnf_amd.experiments import run_language_modeling as R; "
            "out = R.main(sys.argv[1:]); print('RESULT ' + json.dumps({k: v for k, v in out.items() if isinstance(v, (int, float, str))}))" % root)
    r = subprocess.run([sys.executable, "-c", code] + common + ["--max_iterations", "1500", "--eval_freq", "500", "--print_freq", "500",
                                                                 "--learning_rate", "2e-3", "--graph_step"],
                       capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0 and "RESULT " in r.stdout, (r.stdout[-3000:], r.stderr[-3000:])
    assert "hipGraph nodes" in r.stdout and "'memset'" not in r.stdout.split("hipGraph nodes")[1].split("\n")[0], r.stdout[-2000:]
    import json
    out = json.loads(r.stdout.split("RESULT ")[1].splitlines()[0])
    assert out["entropy_rate"] < out["val_bpc"] < out["unigram_entropy"], out
    from categoricalnf_amd.experiments import run_language_modeling as R
    again = R.main(common + ["--only_eval"])
    assert abs(again["val_bpc"] - out["val_bpc"]) < 0.08, (again, out)


@pytest.mark.parametrize("c", [c for c in load_cases("encoder")][:4])
def test_encoder_with_beta_in_a_device_scalar(c):
    """functional.EncoderForwardDevBetaFn (a captured step's beta schedule): the same latents, log-det and class-table gradient as
    EncoderForwardFn given the number, for a beta other than the one the kernels run at."""
    from categoricalnf_amd import functional as Fn
    m = c.meta
    pad = g(c.pad) if m["padded"] else None
    B, N = c.categ.shape
    gen = torch.Generator().manual_seed(5)
    wz, wl = g(torch.randn(B, N, c.table.shape[1] // 2, generator=gen)), g(torch.randn(B, generator=gen))
    res = []
    for beta in (1.0, 1.7):
        t1, t2 = g(c.table).clone().requires_grad_(True), g(c.table).clone().requires_grad_(True)
        z1, l1, _ = Fn.EncoderForwardFn.apply(t1, g(c.categ), g(c.u), g(c.category_prior), pad, beta, True, None, 1e-4)
        z2, l2, _ = Fn.EncoderForwardDevBetaFn.apply(t2, g(c.categ), g(c.u), g(c.category_prior), pad, torch.tensor(beta, device="cuda"), 1e-4)
        ((z1 * wz).sum() + (l1 * wl).sum()).backward()
        ((z2 * wz).sum() + (l2 * wl).sum()).backward()
        assert torch.equal(z1, z2)
        close(l2, l1, rtol=2e-5, atol=2e-4)
        scale = max(float(t1.grad.abs().max()), 1.0)
        assert float((t1.grad - t2.grad).abs().max()) <= 2e-5 * scale, (beta, float((t1.grad - t2.grad).abs().max()), scale)


def test_language_modelling_driver_trains_towards_the_source_entropy(tmp_path):
    """§8 f-3/f-4: the language-modelling host loop (autoregressive mixture coupling + LSTM sub-network, configs[3]'s layer
    stack) on the synthetic Markov source: a small flow trained for 1500 iterations on variable-length sentences beats the
    best context-free model, its checkpoint reloads, and --only_eval reproduces the validation figure."""
    from categoricalnf_amd.experiments import run_language_modeling as R
    common = ["--vocab_size", "9", "--source_alpha", "0.3", "--max_seq_len", "32", "--batch_size", "64", "--num_val", "256",
              "--coupling_hidden_size", "128", "--coupling_hidden_layers", "1", "--coupling_num_mixtures", "9",
              "--encoding_dim", "3", "--variable_length", "--checkpoint_path", str(tmp_path / "lm")]
    out = R.main(common + ["--max_iterations", "1500", "--eval_freq", "500", "--print_freq", "500", "--learning_rate", "2e-3"])
    assert out["entropy_rate"] < out["val_bpc"] < out["unigram_entropy"], out
    assert out["best_file"] and os.path.isfile(out["best_file"])
    again = R.main(common + ["--only_eval"])
    assert abs(again["val_bpc"] - out["val_bpc"]) < 0.08, (again, out)


def test_language_modelling_driver_two_ranks(tmp_path):
    """§8e on the third host loop: 2 ranks (sharing cuda:0 over gloo here; RCCL with the default backend on a multi-GPU
    node) train under DDP on their own sentences, evaluate shards of the held-out set with one all-reduce, and only rank 0
    prints and writes the checkpoint."""
    import re, socket, subprocess, sys
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), PYTHONPATH=root)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), "-m", "categoricalnf_amd.experiments.run_language_modeling",
                        "--share_device", "--vocab_size", "9", "--source_alpha", "0.3", "--max_seq_len", "32", "--batch_size", "64",
                        "--num_val", "250", "--coupling_hidden_size", "64", "--coupling_hidden_layers", "1",
                        "--coupling_num_mixtures", "9", "--variable_length", "--max_iterations", "60", "--eval_freq", "30",
                        "--print_freq", "30", "--checkpoint_path", str(tmp_path / "lm2")],
                       capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    finals = re.findall(r"final: validation ([0-9.]+) bits per character", r.stdout)
    assert len(finals) == 1 and 1.9 < float(finals[0]) < 4.0, r.stdout[-1500:]
    assert len(re.findall(r"iteration +30 \| validation", r.stdout)) == 1
    assert any(f.endswith(".tar") for f in os.listdir(tmp_path / "lm2"))


def test_set_modelling_driver_flat_optimizer_matches_per_tensor_training(tmp_path):
    """--flat_optimizer (RAdam, clipping and zero_grad on one flat buffer) trains to the same validation figure as the
    per-tensor optimiser on the same data stream; its checkpoint reloads for evaluation."""
    from categoricalnf_amd.experiments import run_set_modeling as R
    common = ["--dataset", "shuffling", "--set_size", "8", "--batch_size", "128", "--coupling_num_flows", "2",
              "--coupling_hidden_size", "32", "--coupling_hidden_layers", "1", "--max_iterations", "60", "--eval_freq", "60",
              "--print_freq", "60", "--save_freq", "60"]
    a = R.main(common + ["--checkpoint_path", str(tmp_path / "per_tensor")])
    b = R.main(common + ["--checkpoint_path", str(tmp_path / "flat"), "--flat_optimizer"])
    assert abs(a["val_bpd"] - b["val_bpd"]) < 0.02, (a, b)           # encoder noise differs between two evaluations
    c = R.main(common + ["--checkpoint_path", str(tmp_path / "flat"), "--only_eval"])
    assert abs(c["val_bpd"] - b["val_bpd"]) < 0.02, (b, c)
    blob = torch.load(sorted((tmp_path / "flat").glob("*.tar"))[-1], weights_only=False)
    assert "optimizer_state_dict" not in blob and "scheduler_state_dict" in blob


def test_graph_colouring_driver_two_ranks(tmp_path):
    """§8e on the second host loop: 2 ranks (sharing cuda:0 over gloo here) train under DDP on their own bucketed batches,
    deal the evaluation batches out between them and all-reduce (sum of NLL, graphs, valid colourings) once."""
    import re, socket, subprocess, sys
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), PYTHONPATH=root)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), "-m", "categoricalnf_amd.experiments.run_graph_coloring",
                        "--share_device", "--dataset", "tiny_3", "--data_root", str(tmp_path / "data"), "--generate_data",
                        "--num_graphs", "2000", "--coupling_hidden_size", "32", "--coupling_hidden_layers", "2",
                        "--coupling_num_flows", "2", "--checkpoint_path", str(tmp_path / "ck"), "--print_freq", "40",
                        "--eval_batch_size", "128", "--max_iterations", "80", "--eval_freq", "80", "--batch_size", "128",
                        "--learning_rate", "2e-3"], capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    finals = re.findall(r"final: validation ([0-9.]+) bits per node / ([0-9.]+) % valid, test ([0-9.]+)", r.stdout)
    assert len(finals) == 1 and 0.3 < float(finals[0][0]) < 1.7 and 0.0 <= float(finals[0][1]) <= 100.0, r.stdout[-1500:]
    assert any(f.endswith(".tar") for f in os.listdir(tmp_path / "ck"))


def test_edge_gnn_golden_on_the_device():
    """The in-package Edge-GNN (configs[4]'s stage-2 / 3 sub-network; dense masked attention on PyTorch-ROCm) on the GPU against the
    reference's outputs at the Zinc250k graph sizes (tests/golden/edge_gnn.npz)."""
    from tests.test_host_cpu import edge_gnn_from_golden, run_edge_gnn_case
    cases = load_cases("edge_gnn")
    for c in cases:
        on, oe = run_edge_gnn_case(edge_gnn_from_golden(cases, c, "cuda"), c, "cuda")
        scale_n, scale_e = c.out_nodes.abs().max().item(), c.out_edges.abs().max().item()
        assert (on.cpu() - c.out_nodes).abs().max().item() <= 5e-5 * max(scale_n, 1.0), c.meta
        assert (oe.cpu() - c.out_edges).abs().max().item() <= 5e-5 * max(scale_e, 1.0), c.meta


def test_split_k_linear_has_the_gradients_of_nn_linear():
    """graph_layers.SplitKLinear (the Linear layers of the Edge-GNN that run on the pair list, 45 000 rows a batch): same
    output bits as nn.Linear, gradients of input / weight / bias to fp32 re-association of the sum over rows, same
    state_dict; short inputs and no-grad calls take F.linear itself."""
    from categoricalnf_amd.layers.networks.graph_layers import SplitKLinear
    torch.manual_seed(3)
    for rows, i_f, o_f in ((64 * 703, 128, 128), (16 * 703, 128, 4), (8192, 96, 52)):
        ref = torch.nn.Linear(i_f, o_f).cuda()
        lin = SplitKLinear(i_f, o_f).cuda()
        lin.load_state_dict(ref.state_dict())
        x = torch.randn(rows // 703 if rows % 703 == 0 else 1, 703 if rows % 703 == 0 else rows, i_f, device="cuda")
        xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
        ya, yb = ref(xa), lin(xb)
        assert torch.equal(ya, yb)
        go = torch.randn_like(ya)
        ya.backward(go); yb.backward(go)
        assert torch.equal(xa.grad, xb.grad)
        for pa, pb in zip(ref.parameters(), lin.parameters()):
            scale = pa.grad.abs().max().item()
            assert (pa.grad - pb.grad).abs().max().item() <= 2e-5 * scale
        assert list(lin.state_dict()) == list(ref.state_dict())
    with torch.no_grad():
        assert torch.equal(lin(x), ref(x))


@pytest.mark.parametrize("c", load_cases("encoder"))
def test_fused_encoder_entry_points_against_the_oracle_on_the_golden_cases(c):
    """The three fused entry points of round 3 directly against the oracle on the reference's golden encoder cases (they are
    also tested bit for bit against the chains of kernels they replace, tests/test_gpu_encoder_kernels.py):
    cnf_encoder_forward_sampled (sampler + encoder) against the golden latents / log-det; cnf_encoder_forward_actconv against
    O.encoder_forward -> O.actnorm -> O.invconv; cnf_encoder_decode_actconv against O.invconv / O.actnorm reversed ->
    O.encoder_decode (linear_encoding.py:59-133, 184-196; activation_normalization.py:24-48; permutation_layers.py:106-136)."""
    m = c.meta
    B, N = c.categ.shape
    D, C = c.table.shape[1] // 2, c.table.shape[0]
    pad = c.pad if m["padded"] else None
    # (1) the sampler fused into the forward
    z, ldj, _ = ops().encoder_forward(g(c.categ), g(c.u), g(c.table), g(c.category_prior), beta=m["beta"],
                                      channel_padding_mask=g(pad) if pad is not None else None, uniform_squeeze=1e-4)
    close(z, c.z, **ELEM); loglik_close(ldj, c.ldj)
    # (2) + ActNorm + 1x1 convolution behind it
    gen = torch.Generator().manual_seed(17 + C + D)
    bias, scales = torch.randn(1, 1, D, generator=gen), 0.2 * torch.randn(1, 1, D, generator=gen)
    w = torch.linalg.qr(torch.randn(D, D, generator=gen))[0].contiguous() + 0.05 * torch.randn(D, D, generator=gen)
    sldj = torch.slogdet(w)[1]
    ldj0 = torch.randn(B, generator=gen)
    length = pad.reshape(B, N).sum(1) if pad is not None else None
    eps = O.logistic_from_uniform(c.u.reshape(B * N, 1, D))
    zo, lo, _ = O.encoder_forward(c.categ, eps, c.table, c.category_prior, beta=m["beta"], channel_padding_mask=pad)
    za, la = O.actnorm(zo, bias, scales, length=length, channel_padding_mask=pad, ldj=ldj0 + lo)
    zc, lc = O.invconv(za, w, sldj, length=length, channel_padding_mask=pad, ldj=la)
    zf, lf = ops().encoder_forward_actconv(g(c.categ), g(c.u), g(c.table), g(c.category_prior), g(bias), g(scales), g(w), g(sldj), beta=m["beta"],
                                           channel_padding_mask=g(pad) if pad is not None else None, length=g(length) if length is not None else None,
                                           ldj=g(ldj0), uniform_squeeze=1e-4)
    close(zf, zc, rtol=5e-5, atol=5e-5); loglik_close(lf, lc)
    # (3) the sampling direction: inverse convolution, inverse ActNorm, arg-max decode
    w_inv = torch.inverse(w.double()).float()
    probe = zc + 0.3 * torch.randn(B, N, D, generator=gen) * (pad if pad is not None else 1.0)
    xi, li = O.invconv(probe, w, sldj, reverse=True, length=length, channel_padding_mask=pad, ldj=ldj0.clone())     # inverts w itself
    xa, la2 = O.actnorm(xi, bias, scales, reverse=True, length=length, channel_padding_mask=pad, ldj=li)
    dec_o, score = O.encoder_decode(xa, c.table, c.category_prior)
    dec, ld = ops().encoder_decode_actconv(g(probe), g(bias), g(scales), g(w_inv), g(sldj), g(c.table), g(c.category_prior),
                                           channel_padding_mask=g(pad) if pad is not None else None, length=g(length) if length is not None else None, ldj=g(ldj0))
    loglik_close(ld, la2)
    differ = dec.cpu() != dec_o
    if differ.any():        # only where the oracle's two best classes are a rounding error apart
        top2 = score.topk(min(2, C), dim=-1).values
        gap = (top2[:, 0] - top2[:, -1]).reshape(B, N)
        assert C > 1 and float(gap[differ].max()) < 1e-4, "decoded classes differ where the scores are %g apart" % float(gap[differ].max())


@pytest.mark.parametrize("B,N,D", [(64, 16, 4), (33, 17, 6), (129, 38, 3), (7, 5, 5), (256, 64, 8), (16, 9, 2), (5, 3, 1), (9, 11, 7)])
def test_standalone_actnorm_and_invconv_forward_kernels_equal_the_fused_pair(B, N, D):
    """cnf_actnorm / cnf_invconv run the fused pair's token-owner kernel with the other layer compiled out (D in {1..6, 8}; their own
    older kernels, bit-identical, were removed in round 6 with the switch that selected them; D = 7 takes the generic kernels):
    ActNorm then convolution == the fused pair, bit for bit, in both directions, with padding and lengths, and against the oracle."""
    gen = torch.Generator().manual_seed(B * 7 + D)
    z = torch.randn(B, N, D, generator=gen)
    bias, scales = torch.randn(1, 1, D, generator=gen), 0.3 * torch.randn(1, 1, D, generator=gen)
    w = torch.linalg.qr(torch.randn(D, D, generator=gen))[0].contiguous() + 0.05 * torch.randn(D, D, generator=gen)
    sldj = torch.slogdet(w)[1]
    ln = torch.randint(1, N + 1, (B,), generator=gen); ln[0] = N
    pad = O.length_mask(ln, N)
    ldj0 = torch.randn(B, generator=gen)
    if D != 7:
        for kw in ({}, {"channel_padding_mask": g(pad)}, {"length": g(ln.float())}, {"channel_padding_mask": g(pad), "length": g(ln.float())}):
            a, la = ops().actnorm(g(z), g(bias), g(scales), ldj=g(ldj0).clone(), **kw)
            c, lc = ops().invconv(a, g(w), g(sldj), ldj=la, **kw)
            f, lf = ops().actnorm_invconv(g(z), g(bias), g(scales), g(w), g(sldj), ldj=g(ldj0).clone(), **kw)
            assert torch.equal(c, f) and torch.equal(lc, lf), kw.keys()
    # against the oracle once (the older kernels are golden-tested; this pins the new default directly as well)
    zo, lo = O.actnorm(z, bias, scales, channel_padding_mask=pad, ldj=ldj0.clone())
    za, la = ops().actnorm(g(z), g(bias), g(scales), channel_padding_mask=g(pad), ldj=g(ldj0).clone())
    close(za, zo, **ELEM); loglik_close(la, lo)
    zo, lo = O.invconv(z, w, sldj, channel_padding_mask=pad, ldj=ldj0.clone())
    zc, lc = ops().invconv(g(z), g(w), g(sldj), channel_padding_mask=g(pad), ldj=g(ldj0).clone())
    close(zc, zo, **ELEM); loglik_close(lc, lo)


def test_deferred_reductions_give_the_bits_of_the_immediate_ones():
    """cnf_bwd_defer_begin / cnf_bwd_defer_flush: the streaming backward entry points queue their closing parameter-gradient
    reductions, the flush runs them as one launch — same kernel body, same order: torch.equal with the immediate reductions,
    for every entry point that takes part (general/train.py:144-155: one backward pass, many layers)."""
    from categoricalnf_amd import _lib
    from categoricalnf_amd.ops import _ptr as P, _stream
    lib = _lib.load()
    dev = torch.device("cuda:0")
    st = _stream(dev)
    B, N, D = 512, 40, 6
    gen = torch.Generator(device=dev).manual_seed(5)
    rn = lambda *s, k=1.0: k * torch.randn(*s, generator=gen, device=dev)
    z, nn, gz, gl = rn(B, N, D), rn(B, N, 2 * D, k=0.5), rn(B, N, D), rn(B)
    sf, bias, scales = rn(D, k=0.1), rn(D), rn(D, k=0.1)
    w = torch.linalg.qr(torch.randn(D, D))[0].to(dev).contiguous()
    ln = torch.randint(N // 2, N + 1, (B,), generator=gen, device=dev).float()
    from categoricalnf_amd.layers.flows.coupling_layer import CouplingLayer
    mask = CouplingLayer.create_channel_mask(D).to(dev).contiguous()
    nws = int(lib.cnf_bwd_workspace_floats(D * D + 2 * D + 2))

    def run(deferred):
        outs = {k: torch.full((n,), float("nan"), device=dev) for k, n in (("sf", D), ("b", D), ("s", D), ("w", D * D), ("sl", 1), ("par", D * D + 1 + 2 * D), ("par2", D * D + 1 + 2 * D))}
        o1, o2 = torch.empty(B, N, D, device=dev), torch.empty(B, N, 2 * D, device=dev)
        ws = [torch.empty(nws, device=dev) for _ in range(5)]
        if deferred:
            lib.cnf_bwd_defer_begin()
        rcs = [lib.cnf_affine_coupling_bwd(P(z), P(nn), P(sf), P(mask), 1, D, P(gz), P(gl), P(o1), P(o2), P(outs["sf"]), P(ws[0]), B, N, D, 0, st),
               lib.cnf_actnorm_bwd(P(z), P(bias), P(scales), None, P(ln), P(gz), P(gl), P(o1), P(outs["b"]), P(outs["s"]), P(ws[1]), B, N, D, 0, st),
               lib.cnf_invconv_bwd(P(z), P(w), None, P(ln), P(gz), P(gl), P(o1), P(outs["w"]), P(outs["sl"]), P(ws[2]), B, N, D, 0, st),
               lib.cnf_actnorm_invconv_bwd(P(z), 0, P(bias), P(scales), P(w), None, None, P(ln), P(gz), P(gl), P(o1), P(outs["par"]), P(ws[3]), B, N, D, st),
               lib.cnf_actnorm_invconv_bwd(P(z), 1, P(bias), P(scales), P(w), None, None, P(ln), P(gz), P(gl), P(o1), P(outs["par2"]), P(ws[4]), B, N, D, st)]
        if deferred:
            rcs.append(lib.cnf_bwd_defer_flush(st))
        torch.cuda.synchronize()
        assert all(rc == 0 for rc in rcs), lib.cnf_last_error()
        return outs
    now, later = run(False), run(True)
    for k in now:
        assert torch.isfinite(later[k]).all(), k
        assert torch.equal(now[k], later[k]), k
    # after the flush the library is back to immediate reductions
    again = run(False)
    assert all(torch.equal(now[k], again[k]) for k in now)


@pytest.mark.parametrize("B,N,D", [(256, 64, 6), (64, 16, 4), (40, 16, 2), (33, 20, 3), (16, 38, 6), (7, 5, 8), (130, 64, 6), (9, 11, 5), (4, 703, 2)])
@pytest.mark.parametrize("with_sf", [True, False])
def test_affine_coupling_actnorm_conv_fusion_is_bit_identical_to_the_chain(B, N, D, with_sf):
    """cnf_affine_coupling_actconv: affine coupling + ActNorm + 1x1 convolution of the next flow step in one kernel (forward), the
    coupling's inverse + the inverted convolution and ActNorm of its own step (reverse) — z and log-det torch.equal to
    cnf_affine_coupling followed by cnf_actnorm_invconv, with padding and lengths, and against the oracle's three layers.
    Shapes outside the fused kernel (D = 5; N * D odd; rows too long for a wave tile) take the two kernels: same results."""
    gen = torch.Generator().manual_seed(B * 11 + N + D)
    z = torch.randn(B, N, D, generator=gen)
    nn_out = 0.5 * torch.randn(B, N, 2 * D, generator=gen)
    sf = 0.1 * torch.randn(D, generator=gen) if with_sf else None
    mask = O.channel_mask(D)
    bias, scales = torch.randn(1, 1, D, generator=gen), 0.3 * torch.randn(1, 1, D, generator=gen)
    w = torch.linalg.qr(torch.randn(D, D, generator=gen))[0].contiguous() + 0.05 * torch.randn(D, D, generator=gen)
    sldj = torch.slogdet(w)[1]
    w_inv = torch.inverse(w.double()).float().contiguous()
    ln = torch.randint(1, N + 1, (B,), generator=gen); ln[0] = N
    pad = O.length_mask(ln, N)
    ldj0 = torch.randn(B, generator=gen)
    for kw in ({}, {"channel_padding_mask": g(pad)}, {"length": g(ln.float())}, {"channel_padding_mask": g(pad), "length": g(ln.float())}):
        for reverse, weight in ((False, w), (True, w_inv)):
            zc, lc = ops().affine_coupling(g(z), g(nn_out), g(sf), g(mask), reverse=reverse, ldj=g(ldj0))
            zp, lp = ops().actnorm_invconv(zc, g(bias), g(scales), g(weight), g(sldj), reverse=reverse, ldj=lc, **kw)
            zf, lf = ops().affine_coupling_actconv(g(z), g(nn_out), g(sf), g(mask), g(bias), g(scales), g(weight), g(sldj), reverse=reverse,
                                                   ldj=g(ldj0), **kw)
            assert torch.equal(zf, zp) and torch.equal(lf, lp), (kw.keys(), reverse)
    zo, lo = O.affine_coupling(z, nn_out, mask, sf, ldj=ldj0)
    zo, lo = O.actnorm(zo, bias, scales, channel_padding_mask=pad, ldj=lo)
    zo, lo = O.invconv(zo, w, sldj, channel_padding_mask=pad, ldj=lo)
    zf, lf = ops().affine_coupling_actconv(g(z), g(nn_out), g(sf), g(mask), g(bias), g(scales), g(w), g(sldj), channel_padding_mask=g(pad), ldj=g(ldj0))
    close(zf, zo, **ELEM); loglik_close(lf, lo)
    ops().check_flags(torch.device("cuda"), "affine actconv fusion")


def test_flow_model_fuses_affine_couplings_with_the_next_steps_pair():
    """FlowModel (no grad): [ActNorm, conv, affine coupling] x 4 — every coupling but the last runs fused with the next step's
    ActNorm + convolution (forward), every coupling with its own step's inverted pair (reverse); same bits as the layer-by-layer
    pass, and the sampling direction closes the round trip."""
    from categoricalnf_amd.layers.flows.activation_normalization import ActNormFlow
    from categoricalnf_amd.layers.flows.coupling_layer import CouplingLayer
    from categoricalnf_amd.layers.flows.flow_model import FlowModel
    from categoricalnf_amd.layers.flows.permutation_layers import InvertibleConv
    D, B, N = 6, 128, 16
    torch.manual_seed(3)
    mask = CouplingLayer.create_channel_mask(D)
    mf = lambda c_out: nn.Sequential(nn.Linear(D, 32), nn.GELU(), nn.Linear(32, c_out))
    layers = []
    for _ in range(4):
        layers += [ActNormFlow(D), InvertibleConv(D), CouplingLayer(D, mask, mf)]
    flow = FlowModel(layers).cuda()
    with torch.no_grad():
        for p in flow.parameters():
            p.add_(0.05 * torch.randn_like(p))
    z = torch.randn(B, N, D, device="cuda")
    calls = []
    o = ops()
    real = o._launch

    def spy(dev, name, *a, **kw):
        calls.append(name)
        return real(dev, name, *a, **kw)
    o._launch = spy
    try:
        with torch.no_grad():
            zf, lf = flow(z)
            zr, lr = flow(zf, reverse=True)
            fused = list(calls)
            o.FUSE_LAYERS = False
            z1, l1 = flow(z)
            zr1, lr1 = flow(z1, reverse=True)
    finally:
        o._launch = real
        o.FUSE_LAYERS = True
    assert fused.count("cnf_affine_coupling_actconv") == 3 + 4, fused
    assert torch.equal(zf, z1) and torch.equal(lf, l1) and torch.equal(zr, zr1) and torch.equal(lr, lr1)
    close(zr, z, rtol=1e-3, atol=1e-3)
