"""Compact parameter layout of the masked mixture-CDF coupling (cnf_mixture_coupling_compact*, round 6).

The reference's sub-network emits parameter blocks for all D channels and get_mixt_params multiplies the untransformed channels'
blocks by the zero mask (mixture_cdf_layer.py:65-78, 163-171).  The compact entry points take the transformed channels' blocks
only, [B, N, n_act * (2 + 3K)].  Parity: the reference's goldens and the oracle with the injected nn_out sliced to the transformed
blocks; bit-identity with the reference-layout kernels (same arithmetic, only a token's span address differs); gradients incl. the
sliced last Linear's; the module switch (MixtureCDFCoupling(compact_params=True)) with the reference's parameter names."""
import copy

import pytest
import torch
import torch.nn as nn

from categoricalnf_amd import _lib
from oracle import cnf_oracle as O
from tests.golden_util import load_cases

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
    a, r = actual.detach().double().cpu(), ref.detach().double().cpu()
    assert a.shape == r.shape
    worst = ((a - r).abs() / r.abs().clamp(min=floor)).max().item() if a.numel() else 0.0
    assert worst <= rel, "relative deviation %.3g exceeds %.1g" % (worst, rel)


def compact_of(nn_out, mask, K):
    """the transformed channels' blocks of a reference-layout nn_out [B,N,D*P] under a channel mask [1,D]"""
    P = 2 + 3 * K
    m = mask.reshape(-1)
    act = [i for i in range(m.numel()) if m[i].item() == 0.0]
    B, N = nn_out.shape[:2]
    return nn_out.reshape(B, N, m.numel(), P)[:, :, act[0]:act[0] + len(act)].reshape(B, N, len(act) * P).contiguous(), act


def _case(B, N, D, K, kind, seed, padded=True):
    gen = torch.Generator().manual_seed(seed)
    z = 1.5 * torch.randn(B, N, D, generator=gen)
    nn_out = 0.6 * torch.randn(B, N, D * (2 + 3 * K), generator=gen)
    sf, msf = 0.2 * torch.randn(D, generator=gen), 0.2 * torch.randn(D, K, generator=gen)
    mask = 1.0 - O.channel_mask(D) if kind == "channel_inv" else O.channel_mask(D)
    ln = torch.randint(max(1, N // 2), N + 1, (B,), generator=gen)
    ln[0] = N
    pad = O.length_mask(ln, N) if padded else None
    return z, nn_out, sf, msf, mask, ln, pad


def _in_mode(mode, fn):
    lib = _lib.load()
    lib.cnf_set_math_mode(mode)
    try:
        return fn()
    finally:
        lib.cnf_set_math_mode(1)


@pytest.mark.parametrize("c", [c for c in load_cases("mixture_coupling") if c.meta["mask_kind"] == "channel"])
def test_reference_goldens_on_the_compact_layout(c):
    """the reference's own outputs (tests/golden/mixture_coupling.npz) with its nn_out sliced to the transformed blocks"""
    m = c.meta
    mask, pad = c.get("mask"), c.get("pad")
    kw = dict(num_mixtures=m["K"], scaling_factor=g(c.scaling_factor), mixture_scaling_factor=g(c.mixture_scaling_factor),
              channel_padding_mask=g(pad), reg_max=m["reg_max"], reg_factor=m["reg_factor"], is_training=m["training"])
    nn_c, _ = compact_of(c.nn_out, mask, m["K"])
    assert nn_c.numel() < c.nn_out.numel()
    zf, lf, reg = ops().mixture_coupling(g(c.z), g(nn_c), g(mask), reverse=False, **kw)
    etol = dict(rtol=1e-4, atol=1e-4) if m.get("tail", 1.0) > 1.0 else ELEM
    close(zf, c.z_fwd, **etol); loglik_close(lf, c.ldj_fwd)
    if "reg_ldj" in c:
        loglik_close(reg, c.reg_ldj)
    if "z_rev" in c:
        nn_r, _ = compact_of(c.get("nn_out_rev", c.nn_out), mask, m["K"])
        zr, lr, _ = ops().mixture_coupling(g(c.z_fwd), g(nn_r), g(mask), reverse=True, **kw)
        close(zr, c.z_rev, rtol=1e-4, atol=1e-4); loglik_close(lr, c.ldj_rev)
    ops().check_flags(torch.device("cuda"), "compact goldens")


SHAPES = [(40, 16, 4, 8, "channel"), (24, 38, 6, 16, "channel"), (6, 703, 2, 8, "channel"), (16, 50, 6, 16, "channel_inv"),
          (70, 5, 3, 4, "channel"), (300, 1, 2, 8, "channel"), (5, 97, 5, 9, "channel"), (2, 1500, 3, 5, "channel_inv"),
          (33, 20, 2, 8, "channel"), (1, 2048, 4, 8, "channel"), (130, 64, 6, 8, "channel"), (9, 31, 6, 51, "channel")]


@pytest.mark.parametrize("mode", [1, 0])
@pytest.mark.parametrize("B,N,D,K,kind", SHAPES)
def test_compact_kernels_give_the_bits_of_the_reference_layout_and_match_the_oracle(mode, B, N, D, K, kind):
    z, nn_out, sf, msf, mask, ln, pad = _case(B, N, D, K, kind, B + 7 * N + 31 * D + K)
    nn_c, act = compact_of(nn_out, mask, K)
    kw = dict(num_mixtures=K, reg_max=3.5, reg_factor=2.0, is_training=True)
    gk = dict(scaling_factor=g(sf), mixture_scaling_factor=g(msf), channel_padding_mask=g(pad), **kw)
    ldj0 = torch.randn(B, generator=torch.Generator().manual_seed(5))
    zo, lo, ro = O.mixture_coupling(z, nn_out, mask, scaling_factor=sf, mixture_scaling_factor=msf, channel_padding_mask=pad, **kw)
    full = _in_mode(mode, lambda: ops().mixture_coupling(g(z), g(nn_out), g(mask), ldj=g(ldj0), **gk))
    comp = _in_mode(mode, lambda: ops().mixture_coupling(g(z), g(nn_c), g(mask), ldj=g(ldj0), **gk))
    close(comp[0], zo, **ELEM); loglik_close(comp[1], lo + ldj0); loglik_close(comp[2], ro)
    assert torch.equal(comp[0], full[0]) and torch.equal(comp[1], full[1])
    close(comp[2], full[2], rtol=1e-6, atol=1e-6)             # the regulariser sum is a float atomic in both layouts
    # inverse (Newton; fp64 polish in mode 0)
    full_r = _in_mode(mode, lambda: ops().mixture_coupling(g(zo), g(nn_out), g(mask), reverse=True, **gk))
    comp_r = _in_mode(mode, lambda: ops().mixture_coupling(g(zo), g(nn_c), g(mask), reverse=True, **gk))
    assert torch.equal(comp_r[0], full_r[0]) and torch.equal(comp_r[1], full_r[1])
    zo2, lo2, _ = O.mixture_coupling(zo, nn_out, mask, scaling_factor=sf, mixture_scaling_factor=msf, channel_padding_mask=pad,
                                     reverse=True, **kw)
    close(comp_r[0], zo2, rtol=1e-4, atol=1e-4); loglik_close(comp_r[1], lo2)
    ops().check_flags(torch.device("cuda"), "compact kernels")
    torch.cuda.synchronize()
    for w in ops()._mix_ws.values():
        assert int(w.count_nonzero().item()) == 0


@pytest.mark.parametrize("B,N,D,K,kind", SHAPES)
def test_compact_nll_and_actconv_variants(B, N, D, K, kind):
    z, nn_out, sf, msf, mask, ln, pad = _case(B, N, D, K, kind, 17 + B + N + K)
    nn_c, act = compact_of(nn_out, mask, K)
    gk = dict(scaling_factor=g(sf), mixture_scaling_factor=g(msf), channel_padding_mask=g(pad))
    ldj0 = torch.randn(B, generator=torch.Generator().manual_seed(6))
    acc_f = torch.zeros(ops().NLL_ACC_SLOTS, dtype=torch.int64, device="cuda")
    acc_c = torch.zeros_like(acc_f)
    full = ops().mixture_coupling_nll(g(z), g(nn_out), g(mask), K, ldj=g(ldj0), length=g(ln), acc=acc_f, **gk)
    comp = ops().mixture_coupling_nll(g(z), g(nn_c), g(mask), K, ldj=g(ldj0), length=g(ln), acc=acc_c, **gk)
    for a, b in zip(comp, full):
        if a is not None:
            assert torch.equal(a, b)
    assert torch.equal(acc_c, acc_f)
    zo, lo, _ = O.mixture_coupling(z, nn_out, mask, K, sf, msf, channel_padding_mask=pad)
    loglik_close(comp[4], O.nll_per_sample(zo, lo + ldj0, ln.float(), pad))
    if D in (2, 3, 4, 6):
        gen = torch.Generator().manual_seed(1)
        bias, sc = torch.randn(1, 1, D, generator=gen), 0.2 * torch.randn(1, 1, D, generator=gen)
        w = torch.linalg.qr(torch.randn(D, D, generator=gen))[0].contiguous()
        sldj = torch.slogdet(w)[1]
        full = ops().mixture_coupling_actconv(g(z), g(nn_out), g(mask), K, g(bias), g(sc), g(w), g(sldj), ldj=g(ldj0), length=g(ln), **gk)
        comp = ops().mixture_coupling_actconv(g(z), g(nn_c), g(mask), K, g(bias), g(sc), g(w), g(sldj), ldj=g(ldj0), length=g(ln), **gk)
        assert torch.equal(comp[0], full[0]) and torch.equal(comp[1], full[1])
        chain = ops().mixture_coupling(g(z), g(nn_c), g(mask), K, ldj=g(ldj0), **gk)
        chain = ops().actnorm_invconv(chain[0], g(bias), g(sc), g(w), g(sldj), length=g(ln), channel_padding_mask=g(pad), ldj=chain[1])
        assert torch.equal(comp[0], chain[0]) and torch.equal(comp[1], chain[1])
    ops().check_flags(torch.device("cuda"), "compact variants")


def _grads(z, nn_out, sf, msf, mask, pad, K, gz, gl, reg):
    from categoricalnf_amd import functional as Fn
    zz, nn_ = g(z).requires_grad_(True), g(nn_out).requires_grad_(True)
    sf_, msf_ = g(sf).requires_grad_(True), g(msf).requires_grad_(True)
    zo, lo, _ = Fn.MixtureCouplingFn.apply(zz, nn_, sf_, msf_, None, g(mask), g(pad), K, reg[0], reg[1], True, True, True)
    torch.autograd.backward([zo, lo], [g(gz), g(gl)])
    return [t.grad.detach().cpu() for t in (zz, nn_, sf_, msf_)]


@pytest.mark.parametrize("B,N,D,K,kind", SHAPES)
def test_compact_backward_equals_the_reference_layouts(B, N, D, K, kind):
    """cnf_mixture_coupling_compact_bwd_f32: g_z, g_sf, g_msf bit-equal to the reference-layout kernel's (same passes, same
    order of the parameter-gradient sums), g_nn_compact == the transformed blocks of its g_nn (whose other blocks are zeros)."""
    z, nn_out, sf, msf, mask, ln, pad = _case(B, N, D, K, kind, 555 + B + N)
    nn_c, act = compact_of(nn_out, mask, K)
    gen = torch.Generator().manual_seed(B + K)
    gz, gl = torch.randn(B, N, D, generator=gen), torch.randn(B, generator=gen)
    full = _grads(z, nn_out, sf, msf, mask, pad, K, gz, gl, (3.5, 2.0))
    comp = _grads(z, nn_c, sf, msf, mask, pad, K, gz, gl, (3.5, 2.0))
    P = 2 + 3 * K
    g_full = full[1].reshape(B, N, D, P)
    assert torch.equal(comp[1].reshape(B, N, len(act), P), g_full[:, :, act[0]:act[0] + len(act)])
    rest = [d for d in range(D) if d not in act]
    assert g_full[:, :, rest].abs().max().item() == 0.0
    assert torch.equal(comp[0], full[0])
    for name, a, b in zip(("g_sf", "g_msf"), comp[2:], full[2:]):
        assert torch.equal(a, b), name
    # twice the same bits (no atomics in the reduction)
    again = _grads(z, nn_c, sf, msf, mask, pad, K, gz, gl, (3.5, 2.0))
    for a, b in zip(comp, again):
        assert torch.equal(a, b)


def test_compact_backward_with_tail_elements():
    """latents out to 18 sigma take the fix-up launch (fp64 arithmetic on one element by a whole wave): compact == reference layout"""
    B, N, D, K = 48, 16, 4, 8
    z, nn_out, sf, msf, mask, ln, pad = _case(B, N, D, K, "channel", 31337)
    z = z * 12.0
    nn_c, act = compact_of(nn_out, mask, K)
    gen = torch.Generator().manual_seed(9)
    gz, gl = torch.randn(B, N, D, generator=gen), torch.randn(B, generator=gen)
    full = _grads(z, nn_out, sf, msf, mask, pad, K, gz, gl, (-1.0, 1.0))
    comp = _grads(z, nn_c, sf, msf, mask, pad, K, gz, gl, (-1.0, 1.0))
    P = 2 + 3 * K
    assert torch.equal(comp[1].reshape(B, N, len(act), P), full[1].reshape(B, N, D, P)[:, :, act[0]:act[0] + len(act)])
    for a, b in zip((comp[0], comp[2], comp[3]), (full[0], full[2], full[3])):
        assert torch.equal(a, b)


def test_declined_shapes_and_modes_go_through_the_reference_layout():
    """bisection inverse (inverse mode 0), the round-1 kernels (cnf_set_mixture_kernel(1)) and the fp64 backward (math mode 0) do
    not read the compact layout: CNF_ERR_UNSUPPORTED from the compact entry point, then the expanded tensor through the plain one"""
    B, N, D, K = 12, 20, 4, 8
    z, nn_out, sf, msf, mask, ln, pad = _case(B, N, D, K, "channel", 3)
    nn_c, act = compact_of(nn_out, mask, K)
    lib = _lib.load()
    gk = dict(scaling_factor=g(sf), mixture_scaling_factor=g(msf), channel_padding_mask=g(pad))
    lib.cnf_set_inverse_mode(0)
    try:
        full = ops().mixture_coupling(g(z), g(nn_out), g(mask), K, reverse=True, **gk)
        comp = ops().mixture_coupling(g(z), g(nn_c), g(mask), K, reverse=True, **gk)
    finally:
        lib.cnf_set_inverse_mode(1)
    assert torch.equal(comp[0], full[0]) and torch.equal(comp[1], full[1])
    lib.cnf_set_mixture_kernel(1)
    try:
        full = ops().mixture_coupling(g(z), g(nn_out), g(mask), K, **gk)
        comp = ops().mixture_coupling(g(z), g(nn_c), g(mask), K, **gk)
        full_n = ops().mixture_coupling_nll(g(z), g(nn_out), g(mask), K, length=g(ln), **gk)
        comp_n = ops().mixture_coupling_nll(g(z), g(nn_c), g(mask), K, length=g(ln), **gk)
    finally:
        lib.cnf_set_mixture_kernel(0)
    assert torch.equal(comp[0], full[0]) and torch.equal(comp[1], full[1])
    assert torch.equal(comp_n[4], full_n[4])
    gen = torch.Generator().manual_seed(4)
    gz, gl = torch.randn(B, N, D, generator=gen), torch.randn(B, generator=gen)
    full_g = _in_mode(0, lambda: _grads(z, nn_out, sf, msf, mask, pad, K, gz, gl, (3.5, 2.0)))
    comp_g = _in_mode(0, lambda: _grads(z, nn_c, sf, msf, mask, pad, K, gz, gl, (3.5, 2.0)))
    P = 2 + 3 * K
    assert torch.equal(comp_g[1].reshape(B, N, len(act), P), full_g[1].reshape(B, N, D, P)[:, :, act[0]:act[0] + len(act)])
    for a, b in zip((comp_g[0], comp_g[2], comp_g[3]), (full_g[0], full_g[2], full_g[3])):
        assert torch.equal(a, b)
    # a wrong size is an error, not a guess
    with pytest.raises(ValueError):
        ops().mixture_coupling(g(z), g(nn_c[..., :-1].contiguous()), g(mask), K, **gk)
    ops().check_flags(torch.device("cuda"), "declined shapes")


class _Net(nn.Module):
    """a coupling sub-network in the shape of the reference's (MLP in, a mixing layer over the set, LayerNorm + MLP out)"""

    def __init__(self, c_in, c_out, hidden=64):
        super().__init__()
        self.inp = nn.Sequential(nn.Linear(c_in, hidden), nn.GELU(), nn.Linear(hidden, hidden))
        self.out = nn.Sequential(nn.LayerNorm(hidden), nn.Linear(hidden, hidden), nn.GELU(), nn.Linear(hidden, c_out))

    def forward(self, x, **kwargs):
        h = self.inp(x)
        h = h + h.mean(dim=1, keepdim=True)
        return self.out(h)


@pytest.mark.parametrize("D,K", [(6, 8), (4, 8), (3, 4)])
def test_module_switch_keeps_names_results_and_gradients(D, K):
    """MixtureCDFCoupling(compact_params=True): same state_dict keys as the reference layer, loads the other layer's weights, forward /
    inverse / gradients agree with the reference-layout layer to GEMM rounding; the sliced Linear's gradient has exact zero rows
    for the untransformed channels — what the reference's masked blocks give it."""
    from categoricalnf_amd.layers.flows.coupling_layer import CouplingLayer
    from categoricalnf_amd.layers.flows.mixture_cdf_layer import MixtureCDFCoupling, RowSlicedLinear
    torch.manual_seed(0)
    mask = CouplingLayer.create_channel_mask(D)
    mf = lambda c_out: _Net(D, c_out)
    ref = MixtureCDFCoupling(D, mask, mf, num_mixtures=K, compact_params=False).cuda()
    cmp_ = MixtureCDFCoupling(D, mask, mf, num_mixtures=K, compact_params=True).cuda()
    assert list(ref.state_dict().keys()) == list(cmp_.state_dict().keys())
    with torch.no_grad():
        for p in ref.parameters():
            p.add_(0.05 * torch.randn_like(p))
    cmp_.load_state_dict(ref.state_dict())
    assert isinstance(cmp_.nn.out[3], RowSlicedLinear) and cmp_._compact_rows is not None
    B, N = 64, 16
    z = torch.randn(B, N, D, device="cuda")
    P = 2 + 3 * K
    with torch.no_grad():
        zr, lr, _ = ref(z)
        zc, lc, _ = cmp_(z)
        assert cmp_.run_network(z * cmp_._prepare_mask(cmp_.mask, z)).shape[-1] == (D - D // 2) * P
        close(zc, zr, rtol=1e-4, atol=1e-4); loglik_close(lc, lr)
        zi, li, _ = cmp_(zc, reverse=True)
        close(zi, z, rtol=1e-3, atol=1e-3)
    outs = []
    for layer in (ref, cmp_):
        layer.zero_grad()
        zz = z.clone().requires_grad_(True)
        zo, lo, _ = layer(zz)
        (zo.square().mean() - lo.mean()).backward()
        outs.append([zz.grad] + [p.grad for p in layer.parameters()])
    for a, b in zip(*outs):
        scale = a.abs().max().item() + 1e-6
        assert (a - b).abs().max().item() <= 2e-3 * scale
    gw = cmp_.nn.out[3].weight.grad
    a, b = cmp_._compact_rows
    assert gw[:a].abs().max().item() == 0.0 and gw[a:b].abs().max().item() > 0.0
    # a chess mask keeps the reference layout; so does a sub-network whose output is not its last Linear's rows
    chess = MixtureCDFCoupling(1, CouplingLayer.create_chess_mask(), lambda c_out: _Net(1, c_out), num_mixtures=K, compact_params=True)
    assert chess._compact_linear is None

    class Scrambled(_Net):
        def forward(self, x, **kwargs):
            return super().forward(x).flip(-1)
    odd = MixtureCDFCoupling(D, mask, lambda c_out: Scrambled(D, c_out), num_mixtures=K, compact_params=True).cuda()
    with torch.no_grad(), pytest.warns(UserWarning):
        odd(z)
    assert odd._compact_linear is None
    ops().check_flags(torch.device("cuda"), "module switch")


def test_flow_model_fusions_take_the_compact_layout():
    """FlowModel's fused groups (coupling + ActNorm + 1x1 conv; last coupling + NLL; the training Functions) hand run_network's output
    to the kernels whatever its layout: a flow of compact layers equals the flow of reference-layout layers to GEMM rounding, eval
    and training, and the bits-per-dimension figure with it."""
    from categoricalnf_amd.layers.flows.activation_normalization import ActNormFlow
    from categoricalnf_amd.layers.flows.coupling_layer import CouplingLayer
    from categoricalnf_amd.layers.flows.flow_model import FlowModel
    from categoricalnf_amd.layers.flows.mixture_cdf_layer import MixtureCDFCoupling
    from categoricalnf_amd.layers.flows.permutation_layers import InvertibleConv
    D, K, B, N = 6, 8, 128, 16
    torch.manual_seed(1)
    mask = CouplingLayer.create_channel_mask(D)

    def build(compact):
        layers = []
        for _ in range(3):
            layers += [ActNormFlow(D), InvertibleConv(D), MixtureCDFCoupling(D, mask, lambda c_out: _Net(D, c_out), num_mixtures=K,
                                                                           compact_params=compact)]
        return FlowModel(layers).cuda()
    ref = build(False)
    with torch.no_grad():
        for p in ref.parameters():
            p.add_(0.03 * torch.randn_like(p))
    cmp_ = build(True)
    cmp_.load_state_dict(ref.state_dict())
    z = torch.randn(B, N, D, device="cuda")
    ln = torch.full((B,), float(N), device="cuda")
    with torch.no_grad():
        zr, lr, nr = ref.nll(z, length=ln)
        zc, lc, nc = cmp_.nll(z, length=ln)
    close(zc, zr, rtol=2e-4, atol=2e-4); loglik_close(lc, lr); loglik_close(nc, nr)
    assert abs(O.bits_per_dim(nc.mean().item()) - O.bits_per_dim(nr.mean().item())) < 1e-3
    grads = []
    for flow in (ref, cmp_):
        flow.zero_grad()
        _, _, nll = flow.nll_loss(z, length=ln)
        nll.mean().backward()
        grads.append([p.grad for p in flow.parameters()])
    for a, b in zip(*grads):
        scale = a.abs().max().item() + 1e-6
        assert (a - b).abs().max().item() <= 3e-3 * scale
