"""Pin oracle/cnf_oracle.py to the golden vectors captured from the real reference.

Integer / index outputs must be identical; floating-point outputs are expected bit-equal because the
oracle uses the same torch CPU ops in the same order — asserted with a 1e-6 guard band to stay
robust against thread-count dependent reduction order."""
import os

import numpy as np
import pytest
import torch

from oracle import cnf_oracle as O
from tests.golden_util import load_cases

TOL = dict(rtol=1e-6, atol=1e-6)


def close(a, b, **kw):
    kw = {**TOL, **kw}
    torch.testing.assert_close(a, b, **kw)


@pytest.mark.parametrize("c", load_cases("affine_coupling"))
def test_affine(c):
    zf, lf = O.affine_coupling(c.z, c.nn_out, c.mask, c.scaling_factor, reverse=False, ldj=c.ldj_in)
    close(zf, c.z_fwd); close(lf, c.ldj_fwd)
    zr, lr = O.affine_coupling(c.z_fwd, c.nn_out, c.mask, c.scaling_factor, reverse=True)
    close(zr, c.z_rev); close(lr, c.ldj_rev)
    m = O.expand_mask(c.mask, c.z)
    s, t = O.affine_params(c.nn_out, m, c.scaling_factor)
    close(s, c.s); close(t, c.t)
    s, t = O.affine_params(c.nn_out, m, None)
    close(s, c.s_nofac); close(t, c.t_nofac)


@pytest.mark.parametrize("c", load_cases("mixture_coupling"))
def test_mixture(c):
    m = c.meta
    mask = c.get("mask")
    pad = c.get("pad")
    kw = dict(num_mixtures=m["K"], scaling_factor=c.scaling_factor, mixture_scaling_factor=c.mixture_scaling_factor,
              channel_padding_mask=pad, reg_max=m["reg_max"], reg_factor=m["reg_factor"], is_training=m["training"])
    zf, lf, reg = O.mixture_coupling(c.z, c.nn_out, mask, reverse=False, **kw)
    close(zf, c.z_fwd); close(lf, c.ldj_fwd)
    if "reg_ldj" in c:
        close(reg, c.reg_ldj)
    nn_rev = c.get("nn_out_rev", c.nn_out)
    zr, lr, _ = O.mixture_coupling(c.z_fwd, nn_rev, mask, reverse=True, **kw)
    close(zr, c.z_rev, atol=1e-5); close(lr, c.ldj_rev, atol=1e-5)
    if "p_t" in c:
        p = O.mixture_params(c.nn_out, O.expand_mask(mask, c.z) if mask is not None else None, m["K"],
                             c.scaling_factor, c.mixture_scaling_factor)
        for got, key in zip(p, ["p_t", "p_log_s", "p_log_pi", "p_mixt_t", "p_mixt_log_s"]):
            assert got.dtype == torch.float64
            close(got, c[key], rtol=0, atol=0)
    if m.get("demo"):
        # the reference's own known-answer demo: round trip <= 1.2e-7, ldj error 0
        assert (c.z - zr).abs().max() <= 2e-7
        assert (lf + lr).abs().max() <= 1e-6


@pytest.mark.parametrize("c", load_cases("actnorm"))
def test_actnorm(c):
    mode = c.meta["mode"]
    kw = {}
    if "length" in mode:
        kw["length"] = c.length
    if "mask" in mode:
        kw["channel_padding_mask"] = c.pad
    zf, lf = O.actnorm(c.z, c.bias, c.scales, reverse=False, ldj=c.ldj_in, **kw)
    close(zf, c.z_fwd); close(lf, c.ldj_fwd)
    zr, lr = O.actnorm(c.z_fwd, c.bias, c.scales, reverse=True, **kw)
    close(zr, c.z_rev); close(lr, c.ldj_rev)
    b, s = O.actnorm_data_init(c.z, c.pad.expand(-1, -1, c.z.size(2)) if "mask" in mode else None)
    close(b, c.init_bias); close(s, c.init_scales)


@pytest.mark.parametrize("c", load_cases("ext_actnorm"))
def test_ext_actnorm(c):
    pad = c.pad if c.meta["padded"] else None
    zf, lf = O.ext_actnorm(c.z, c.nn_out, reverse=False, channel_padding_mask=pad, ldj=c.ldj_in)
    close(zf, c.z_fwd); close(lf, c.ldj_fwd)
    zr, lr = O.ext_actnorm(c.z_fwd, c.nn_out, reverse=True, channel_padding_mask=pad)
    close(zr, c.z_rev); close(lr, c.ldj_rev)


@pytest.mark.parametrize("c", load_cases("invconv"))
def test_invconv(c):
    mode = c.meta["mode"]
    kw = {}
    if "length" in mode:
        kw["length"] = c.length
    if "mask" in mode:
        kw["channel_padding_mask"] = c.pad
    if c.meta["lu"]:
        w, sldj = O.invconv_weight_lu(c.sd_p, c.sd_l, c.sd_u, c.sd_log_s, c.sd_sign_s)
    else:
        w, sldj = c.sd_weight, torch.slogdet(c.sd_weight)[1]
    close(w, c.weight); close(sldj, c.sldj)
    zf, lf = O.invconv(c.z, w, sldj, reverse=False, ldj=c.ldj_in, **kw)
    close(zf, c.z_fwd); close(lf, c.ldj_fwd); close(zf, c.z_fwd_eval); close(lf, c.ldj_fwd_eval)
    zr, lr = O.invconv(c.z_fwd, w, sldj, reverse=True, **kw)
    close(zr, c.z_rev); close(lr, c.ldj_rev); close(zr, c.z_rev_eval)
    close(torch.inverse(w.double()).float(), c.inv_weight)


def test_prior():
    for c in load_cases("prior"):
        k = c.meta["kind"]
        if k == "log_prob":
            assert abs(c.meta["sigma"] - O.LOGISTIC_SIGMA) < 1e-15
            close(O.logistic_log_prob(c.x), c.log_prob, rtol=0, atol=0)
        elif k == "sample":
            s = O.logistic_from_uniform(c.u)
            close(s, c.sample, rtol=0, atol=0)
            close(O.logistic_log_prob(s), c.log_prob, rtol=0, atol=0)
        else:
            nll = O.nll_per_sample(c.z, c.ldj, c.length, c.pad)
            close(nll, c.nll)
            assert abs(O.bits_per_dim(float(nll.mean())) - float(c.bpd)) < 1e-6


@pytest.mark.parametrize("c", load_cases("encoder"))
def test_encoder(c):
    m = c.meta
    eps = O.logistic_from_uniform(c.u)
    pad = c.pad if m["padded"] else None
    z, ldj, cpl = O.encoder_forward(c.categ, eps, c.table, c.category_prior, beta=m["beta"], channel_padding_mask=pad)
    close(z, c.z); close(ldj, c.ldj, atol=2e-6)
    if m["training"]:
        w = (pad.reshape(-1) if pad is not None else torch.ones_like(cpl))
        close((cpl.exp() * w).sum() / w.sum(), c.detail_avg_token_prob)
    dec, _ = O.encoder_decode(c.z, c.table, c.category_prior)
    assert torch.equal(dec, c.decoded)
    dec, _ = O.encoder_decode(c.z_probe, c.table, c.category_prior)
    assert torch.equal(dec, c.decoded_probe)


def test_sigmoid():
    for c in load_cases("sigmoid"):
        if not c.meta["reverse_layer"]:
            a, la = O.sigmoid_flow(c.z, reverse=False)
            b, lb = O.sigmoid_flow(c.u, reverse=True)
        else:  # XOR of the layer's own flag (sigmoid_layer.py:29)
            a, la = O.sigmoid_flow(c.u, reverse=True)
            b, lb = O.sigmoid_flow(c.z, reverse=False)
        close(a, c.out_fwd); close(la, c.ldj_fwd); close(b, c.out_rev); close(lb, c.ldj_rev)


@pytest.mark.parametrize("c", load_cases("encoder_large_vocab"))
def test_encoder_large_vocab(c):
    """300 / 1100 / 1300 classes (the drop-in's class-tiled kernels, without and with class splits): the oracle against
    the reference's latents, log-det, decoded classes and — through autograd — the gradient of the class table."""
    m = c.meta
    eps = O.logistic_from_uniform(c.u)
    pad = c.pad if m["padded"] else None
    z, ldj, _ = O.encoder_forward(c.categ, eps, c.table, c.category_prior, beta=m["beta"], channel_padding_mask=pad)
    close(z, c.z); close(ldj, c.ldj, atol=5e-6)
    assert torch.equal(O.encoder_decode(c.z, c.table, c.category_prior)[0], c.decoded)
    assert torch.equal(O.encoder_decode(c.z_probe, c.table, c.category_prior)[0], c.decoded_probe)


@pytest.mark.skipif(not os.path.isdir("/root/reference/layers"), reason="reference checkout only exists in the build container")
@pytest.mark.parametrize("seed", [0, 1])
def test_oracle_equals_the_live_reference_on_random_cases(seed):
    """oracle/fuzz_vs_reference.py: the reference's own layer objects (imported from the checkout, sub-networks replaced by
    injected outputs) against the oracle on seeded random shapes, masks, mixture counts, paddings, regulariser settings and
    train / eval modes — the pin beyond the fixed golden cases.  Runs in a subprocess: the reference's `layers` package
    must not meet this package's aliases."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "oracle", "fuzz_vs_reference.py"), "100", str(seed)],
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd="/tmp", timeout=600,
                         env=dict(os.environ, PYTHONDONTWRITEBYTECODE="1", MPLBACKEND="Agg"))
    import re
    done = re.search(r"FUZZ OK (\d+)", out.stdout)
    assert out.returncode == 0 and done and int(done.group(1)) >= 800, (out.stdout[-800:], out.stderr[-1500:])
    assert "gradients (affine + mixture)" in out.stdout
