"""The mixture-model encoder's kernels against each other and against the oracle (linear_encoding.py:59-133,153-196):
* two independent implementations of the LDS-resident kernels (one / two tokens per lane, cnf_set_encoder_kernel(1 / 2)) must agree
  bit for bit — latents, class posteriors, log-det, decoded indices;
* the forward's density sum and the class-tiled backward's, with their log-domain fallbacks, against the float64 oracle on
  extreme inputs;
* the fused entry points (sampler, ActNorm + 1x1 convolution behind the encoder / in front of the decode) against the chains of
  calls they replace, bit for bit, and inside a real flow model.
The goldens and the oracle comparisons of test_gpu_parity.py run on whichever kernel a shape selects."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup():
    from categoricalnf_amd import _lib, ops
    return _lib.load(), ops


def _inputs(B, N, D, C, seed, pad_mode, dev):
    g = torch.Generator(device=dev).manual_seed(seed)
    categ = torch.randint(0, C, (B, N), generator=g, device=dev)
    table = 0.7 * torch.randn(C, 2 * D, generator=g, device=dev)
    prior = torch.log_softmax(torch.randn(C, generator=g, device=dev), 0)
    u = torch.rand(B * N, D, generator=g, device=dev) * (1 - 1e-4) + 5e-5
    eps = (torch.log(u.double()) - torch.log1p(-u.double())).float() / 1.81
    pad = None
    if pad_mode:
        length = torch.randint(max(1, N // 2), N + 1, (B,), generator=g, device=dev)
        pad = (torch.arange(N, device=dev)[None, :] < length[:, None]).float().unsqueeze(-1)
    ldj = torch.randn(B, generator=g, device=dev)
    return categ, eps, table, prior, pad, ldj


# (B, N, D, C, pair expected): tilings with whole rows per wave and an even number of tokens per tile take the new kernel
SHAPES = [
    (4096, 64, 6, 16, True), (4096, 16, 4, 16, True), (2049, 16, 2, 2, True), (3000, 20, 2, 3, True), (2500, 38, 6, 9, True),
    (2200, 50, 6, 3, True), (2100, 7, 3, 5, True), (4099, 3, 1, 4, True), (2048, 64, 8, 51, True), (2304, 36, 6, 51, True),
    (128, 288, 3, 51, False),           # block-per-row tiling: round-2 kernel
    (2100, 5, 5, 7, False),             # D = 5 has no pair instantiation
    (7, 16, 4, 16, True),               # tiny batch, partial tile
]


@pytest.mark.parametrize("B,N,D,C,expect_pair", SHAPES)
@pytest.mark.parametrize("pad_mode", [0, 1])
def test_pair_forward_is_bit_identical_to_round2_kernel(B, N, D, C, expect_pair, pad_mode):
    lib, ops = _setup()
    dev = torch.device("cuda:0")
    categ, eps, table, prior, pad, ldj = _inputs(B, N, D, C, 100 + B + N, pad_mode, dev)
    out = {}
    for which in (1, 2):
        lib.cnf_set_encoder_kernel(which)
        try:
            n0 = lib.cnf_encoder_pair_launches()
            z, l, cpl = ops.encoder_forward(categ, eps, table, prior, beta=1.7, channel_padding_mask=pad, ldj=ldj,
                                            want_class_prob=True, tiled=False)
            torch.cuda.synchronize()
            out[which] = (z, l, cpl, lib.cnf_encoder_pair_launches() - n0)
        finally:
            lib.cnf_set_encoder_kernel(0)
    assert out[1][3] == 0
    assert out[2][3] == (1 if expect_pair else 0)
    assert torch.equal(out[2][0], out[1][0])
    assert torch.equal(out[2][2], out[1][2])
    assert torch.equal(out[2][1], out[1][1])
    assert torch.isfinite(out[2][1]).all()


@pytest.mark.parametrize("B,N,D,C", [(4096, 64, 6, 16), (4096, 16, 4, 16), (1, 1, 6, 16), (3, 43, 6, 9), (1000, 127, 2, 3), (129, 1, 1, 2),
                                     (2048, 64, 8, 51), (37, 5, 3, 5), (1, 129, 4, 300), (64, 64, 5, 7)])
def test_pair_decode_is_bit_identical_to_round2_kernel(B, N, D, C):
    lib, ops = _setup()
    dev = torch.device("cuda:0")
    categ, eps, table, prior, _, _ = _inputs(B, N, D, C, 7 + B, 0, dev)
    z, _, _ = ops.encoder_forward(categ, eps, table, prior, tiled=False)
    z = z + 0.3 * torch.randn_like(z)                       # perturbed probes: not only the clean forward outputs
    out = {}
    for which in (1, 2):
        lib.cnf_set_encoder_kernel(which)
        try:
            n0 = lib.cnf_encoder_pair_launches()
            out[which] = (ops.encoder_decode(z, table, prior, tiled=False), lib.cnf_encoder_pair_launches() - n0)
        finally:
            lib.cnf_set_encoder_kernel(0)
    assert out[2][1] == (1 if D != 5 else 0) and out[1][1] == 0
    assert torch.equal(out[2][0], out[1][0])


def test_pair_kernels_decline_unaligned_views():
    """A view that starts at an odd token (8 bytes off a 16-byte boundary at D = 6) falls back to the round-2 kernel."""
    lib, ops = _setup()
    dev = torch.device("cuda:0")
    B, N, D, C = 64, 64, 6, 16
    categ, eps, table, prior, _, _ = _inputs(B, N, D, C, 5, 0, dev)
    z, _, _ = ops.encoder_forward(categ, eps, table, prior, tiled=False)
    flat = torch.empty(B * N * D + D, device=dev)
    view = flat[D:].view(B, N, D)
    view.copy_(z)
    assert view.data_ptr() % 16 != 0
    lib.cnf_set_encoder_kernel(2)
    try:
        n0 = lib.cnf_encoder_pair_launches()
        dec = ops.encoder_decode(view, table, prior, tiled=False)
        assert lib.cnf_encoder_pair_launches() == n0
        assert torch.equal(dec, ops.encoder_decode(z, table, prior, tiled=False))
        assert lib.cnf_encoder_pair_launches() == n0 + 1
    finally:
        lib.cnf_set_encoder_kernel(0)


def test_pair_forward_reports_out_of_range_categories():
    lib, ops = _setup()
    dev = torch.device("cuda:0")
    B, N, D, C = 2048, 16, 4, 5
    categ, eps, table, prior, _, _ = _inputs(B, N, D, C, 9, 0, dev)
    categ[3, 2] = C + 4
    lib.cnf_set_encoder_kernel(2)
    try:
        with pytest.raises(AssertionError):
            ops.encoder_forward(categ, eps, table, prior, tiled=False)
            ops.check_flags(dev)
    finally:
        lib.cnf_set_encoder_kernel(0)


def test_automatic_selection_is_the_one_token_kernel():
    """With the density-sum class loop the one-token kernel is as fast or faster at every vocabulary size (cnf_encoder.hip,
    cnf_encoder_forward): the pair kernels run only when asked for."""
    lib, ops = _setup()
    dev = torch.device("cuda:0")
    for C in (16, 24, 51):
        categ, eps, table, prior, _, _ = _inputs(4096, 64, 6, C, 11, 0, dev)
        n0 = lib.cnf_encoder_pair_launches()
        z, _, _ = ops.encoder_forward(categ, eps, table, prior, tiled=False)
        ops.encoder_decode(z, table, prior, tiled=False)
        assert lib.cnf_encoder_pair_launches() - n0 == 0


def _oracle64(categ, eps, table, prior, beta, pad):
    """the oracle's encoder forward on float64 copies of the inputs (linear_encoding.py:59-133,153-174): the reference
    value the fp32 kernels are held to in the extreme cases below, where fp32 torch itself is at its limits"""
    from oracle import cnf_oracle as O
    B, N = categ.shape
    D = eps.shape[-1]
    return O.encoder_forward(categ.cpu(), eps.double().cpu().reshape(B * N, 1, D), table.double().cpu(), prior.double().cpu(),
                             beta=beta, channel_padding_mask=None if pad is None else pad.double().cpu())


@pytest.mark.parametrize("B,N,D,C,which", [(2048, 16, 6, 16, 1), (2048, 16, 6, 16, 2), (2048, 16, 8, 32, 2), (512, 8, 16, 12, 1),
                                           (512, 8, 12, 7, 1), (2050, 6, 4, 51, 2)])
def test_forward_density_sum_and_its_log_domain_fallback(B, N, D, C, which):
    """Round 3's forward sums class DENSITIES relative to the token's own (cnf_encoder.hip: class_density) instead of
    streaming a log-sum-exp; a token whose own density is so small that 2^-lp2 (or its product with the sum) leaves the
    fp32 range takes the log-domain loop.  Inputs: ordinary tokens, tokens with every noise channel at the prior's clamp
    (|eps| = 5.47, the far tail), class tables with far-apart means and a prior with nearly impossible classes
    (log-prior -120: own density ~2^-170 and below).  class_prob_log, latents and per-sample log-det against the
    float64 oracle, at the literal 1e-4 bar; both kernels (one / two tokens per lane) must give the same bits."""
    lib, ops = _setup()
    dev = torch.device("cuda:0")
    categ, eps, table, prior, pad, ldj = _inputs(B, N, D, C, 31 + D + C, 1, dev)
    g = torch.Generator(device=dev).manual_seed(5)
    eps = eps.reshape(B, N, D).clone()
    tail = 9.903487 / 1.81
    eps[: B // 4] = tail * torch.sign(torch.randn(B // 4, N, D, generator=g, device=dev))       # every channel at the clamp
    table = table.clone()
    table[:, :D] *= 6.0                                                                           # class means far apart
    prior = prior.clone()
    prior[::3] = -120.0                                                                           # nearly impossible classes
    eps = eps.reshape(B * N, D).contiguous()
    out = {}
    for k in (0, 1, which):                     # 0: the automatic choice (production tiling), 1 / 2: the forced kernels
        lib.cnf_set_encoder_kernel(k)
        try:
            out[k] = ops.encoder_forward(categ, eps, table, prior, beta=1.3, channel_padding_mask=pad, ldj=ldj,
                                         want_class_prob=True, tiled=False)
            torch.cuda.synchronize()
        finally:
            lib.cnf_set_encoder_kernel(0)
    for a, b in zip(out[1], out[which]):
        assert torch.equal(a, b)
    assert torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][2], out[1][2])     # per-token results: the same bits
    z, l, cpl = out[0]                                                                  # row sums: the tiling's order
    zo, lo, co = _oracle64(categ, eps, table, prior, 1.3, pad)
    assert torch.isfinite(cpl).all() and torch.isfinite(l).all()
    worst = ((cpl.double().cpu() - co).abs() / co.abs().clamp(min=1.0)).max().item()
    assert worst <= 1e-4, worst
    assert torch.allclose(z.double().cpu(), zo, rtol=2e-5, atol=2e-5)
    ref = lo + ldj.double().cpu()
    assert ((l.double().cpu() - ref).abs() / ref.abs().clamp(min=1.0)).max().item() <= 1e-4
    # the extreme tokens did reach the fallback's territory: own log2-density below -127
    assert float(co.min()) < -50.0


@pytest.mark.parametrize("tiled", [False, True])
def test_forward_with_own_density_near_the_denormal_range_and_comparable_rivals(tiled):
    """a token whose OWN density is ~2^-120 keeps 2^-lp2 finite, but rival classes of about the same density
    could flush to zero in the density products and bias the posterior without any overflow.  Every class here has a
    log-prior of about -83 (own density ~2^-120 ... 2^-135 with the noise term), so the rivals are comparable to the token's own
    class for every token; such tokens take the log-domain sweep (density_sum_ok) and match the float64 oracle at 1e-4."""
    lib, ops = _setup()
    dev = torch.device("cuda:0")
    B, N, D, C = 96, 16, 6, 7
    categ, eps, table, prior, pad, ldj = _inputs(B, N, D, C, 77, 1, dev)
    g = torch.Generator(device=dev).manual_seed(3)
    prior = -83.0 + 0.7 * torch.randn(C, generator=g, device=dev)        # not normalised: the kernels take log-priors as given
    table = table.clone()
    table[:, :D] *= 0.5                                                   # overlapping classes: several rivals matter
    z, l, cpl = ops.encoder_forward(categ, eps, table, prior, beta=1.0, channel_padding_mask=pad, ldj=ldj, want_class_prob=True, tiled=tiled)
    zo, lo, co = _oracle64(categ, eps, table, prior, 1.0, pad)
    assert torch.isfinite(cpl).all()
    worst = ((cpl.double().cpu() - co).abs() / co.abs().clamp(min=1.0)).max().item()
    assert worst <= 1e-4, worst
    ref = lo + ldj.double().cpu()
    assert ((l.double().cpu() - ref).abs() / ref.abs().clamp(min=1.0)).max().item() <= 1e-4
    assert float((co < -0.05).double().mean()) > 0.5          # the rivals do carry weight for most tokens


@pytest.mark.parametrize("B,N,D,C", [(64, 16, 6, 16), (40, 8, 16, 12), (6, 30, 4, 1500), (33, 8, 8, 51)])
def test_backward_density_sum_and_its_log_domain_fallback(B, N, D, C):
    """The class-tiled backward's token lanes sum class densities like the forward (cnf_encoder_bwd_tiled.hip) and take a
    log-domain sweep straight from the raw table for a token outside the fp32 range.  Same extreme inputs as the forward's
    test (noise at the prior's clamp in every channel, far-apart class means, log-priors of -120); d loss / d class table
    against float64 autograd through the oracle."""
    from categoricalnf_amd import functional as Fn
    from oracle import cnf_oracle as O
    lib, ops = _setup()
    dev = torch.device("cuda:0")
    categ, eps, table, prior, pad, _ = _inputs(B, N, D, C, 77 + D + C, 1, dev)
    g = torch.Generator(device=dev).manual_seed(6)
    eps = eps.reshape(B, N, D).clone()
    eps[: B // 4] = (9.903487 / 1.81) * torch.sign(torch.randn(B // 4, N, D, generator=g, device=dev))
    eps = eps.reshape(B * N, D).contiguous()
    table = table.clone()
    table[:, :D] *= 6.0
    prior = prior.clone()
    prior[::3] = -120.0
    wz = torch.randn(B, N, D, generator=g, device=dev)
    wl = torch.randn(B, generator=g, device=dev)
    tg = table.clone().requires_grad_()
    z, ldj, _ = Fn.EncoderForwardFn.apply(tg, categ, eps, prior, pad, 1.3, False, True)
    ((z * wz).sum() + (ldj * wl).sum()).backward()
    tc = table.double().cpu().requires_grad_()
    zo, lo, co = O.encoder_forward(categ.cpu(), eps.double().cpu().reshape(B * N, 1, D), tc, prior.double().cpu(), beta=1.3,
                                   channel_padding_mask=pad.double().cpu())
    ((zo * wz.double().cpu()).sum() + (lo * wl.double().cpu()).sum()).backward()
    assert float(co.detach().min()) < -50.0             # tokens in the fallback's territory are present
    assert torch.isfinite(tg.grad).all()
    scale = float(tc.grad.abs().max())
    err = (tg.grad.double().cpu() - tc.grad).abs()
    assert float((err / (2e-3 * tc.grad.abs() + 2e-4 * max(scale, 1.0))).max()) <= 1.0, float(err.max())


@pytest.mark.parametrize("B,N,D,C,tiled", [(2048, 64, 6, 16, False), (512, 16, 4, 51, False), (7, 33, 3, 5, False), (300, 9, 5, 7, False),
                                           (64, 20, 6, 700, True), (9, 12, 4, 2500, True), (2048, 64, 8, 40, False)])
def test_fused_sampler_gives_the_bits_of_the_two_calls(B, N, D, C, tiled):
    """cnf_encoder_forward[_tiled]_sampled (LogisticDistribution.sample fused into the encoder forward, distributions.py:139-145 +
    linear_encoding.py:59-106) against cnf_logistic_from_uniform followed by cnf_encoder_forward[_tiled]: latents, log-det,
    class posterior and the noise it hands to the backward are bit-identical; in math mode 0 the wrapper makes the two calls."""
    lib, ops = _setup()
    dev = torch.device("cuda:0")
    categ, _, table, prior, pad, ldj = _inputs(B, N, D, C, 5 + B + C, 1, dev)
    g = torch.Generator(device=dev).manual_seed(B)
    u = torch.rand(B * N, D, generator=g, device=dev)
    u[0, 0] = 0.0                                         # the interval's closed end (squeezed to 5e-5)
    eps = ops.logistic_from_uniform(u, mu=0.0, sigma=ops.LOGISTIC_SIGMA, eps=1e-4)
    for which in ((0,) if tiled else (0, 1, 2)):
        lib.cnf_set_encoder_kernel(which)
        try:
            ref = ops.encoder_forward(categ, eps, table, prior, beta=1.2, channel_padding_mask=pad, ldj=ldj, want_class_prob=True, tiled=tiled)
            got = ops.encoder_forward(categ, u, table, prior, beta=1.2, channel_padding_mask=pad, ldj=ldj, want_class_prob=True, tiled=tiled,
                                      uniform_squeeze=1e-4, want_noise=True)
        finally:
            lib.cnf_set_encoder_kernel(0)
        for a, b in zip(ref, got[:3]):
            assert torch.equal(a, b)
        assert torch.equal(got[3].reshape(-1), eps.reshape(-1))
    lib.cnf_set_math_mode(0)
    try:
        got0 = ops.encoder_forward(categ, u, table, prior, beta=1.2, channel_padding_mask=pad, ldj=ldj, tiled=tiled, uniform_squeeze=1e-4, want_noise=True)
        eps0 = ops.logistic_from_uniform(u, mu=0.0, sigma=ops.LOGISTIC_SIGMA, eps=1e-4)
        ref0 = ops.encoder_forward(categ, eps0, table, prior, beta=1.2, channel_padding_mask=pad, ldj=ldj, tiled=tiled)
    finally:
        lib.cnf_set_math_mode(1)
    assert torch.equal(got0[0], ref0[0]) and torch.equal(got0[1], ref0[1]) and torch.equal(got0[3].reshape(-1), eps0.reshape(-1))


def test_encoder_module_trains_through_the_fused_sampler():
    """The module's forward (uniform draw in, noise sampled in the kernel) and its backward (which takes the noise the forward
    handed back): parameter gradients equal those of the explicit two-call route."""
    from categoricalnf_amd import functional as Fn
    from categoricalnf_amd.layers.categorical_encoding.linear_encoding import LinearCategoricalEncoding
    lib, ops = _setup()
    torch.manual_seed(1)
    enc = LinearCategoricalEncoding(num_dimensions=6, flow_config={"num_flows": 0}, vocab_size=16).cuda().train()
    for p in enc.parameters():
        p.data.normal_(0.0, 0.4)
    x = torch.randint(0, 16, (32, 24), device="cuda")
    u = torch.rand(32 * 24, 1, 6, device="cuda")
    z, ldj, _ = enc(x, noise=u)
    (z.sum() + ldj.sum()).backward()
    grads = {n: p.grad.clone() for n, p in enc.named_parameters()}
    enc.zero_grad()
    table = enc.class_table()
    eps = ops.logistic_from_uniform(u.reshape(-1, 6), mu=0.0, sigma=ops.LOGISTIC_SIGMA, eps=1e-4)
    z2, ldj2, _ = Fn.EncoderForwardFn.apply(table, x, eps, enc.category_prior, None, 1.0, True)
    (z2.sum() + ldj2.sum()).backward()
    assert torch.equal(z, z2) and torch.equal(ldj, ldj2)
    for n, p in enc.named_parameters():
        assert torch.equal(p.grad, grads[n]), n


@pytest.mark.parametrize("B,N,D,C,padded,use_len", [(2048, 64, 6, 16, 0, 0), (300, 16, 4, 16, 1, 1), (77, 20, 2, 3, 1, 0), (9, 5, 5, 7, 0, 1),
                                                    (2048, 16, 8, 51, 1, 1), (4, 7, 3, 9, 0, 0), (6, 12, 6, 700, 1, 1)])
def test_encoder_actconv_kernel_gives_the_bits_of_the_three_layers(B, N, D, C, padded, use_len):
    """cnf_encoder_forward_actconv (sampled encoder forward + ActNorm + 1x1 convolution in one launch) against the chain
    cnf_encoder_forward_sampled -> cnf_actnorm_invconv: latents and running log-det bit-identical, with and without a padding
    mask / a length vector; a vocabulary beyond the LDS-resident table takes the chain inside the wrapper."""
    lib, ops = _setup()
    dev = torch.device("cuda:0")
    categ, _, table, prior, pad, ldj = _inputs(B, N, D, C, 3 + B + C, padded, dev)
    g = torch.Generator(device=dev).manual_seed(B + 1)
    u = torch.rand(B * N, D, generator=g, device=dev)
    bias, scales = torch.randn(1, 1, D, generator=g, device=dev), 0.2 * torch.randn(1, 1, D, generator=g, device=dev)
    w = torch.linalg.qr(torch.randn(D, D))[0].to(dev) + 0.05 * torch.randn(D, D, generator=g, device=dev)
    sldj = torch.randn((), generator=g, device=dev)
    length = torch.randint(max(1, N // 2), N + 1, (B,), generator=g, device=dev).float() if use_len else None
    z1, l1, _ = ops.encoder_forward(categ, u, table, prior, beta=1.1, channel_padding_mask=pad, uniform_squeeze=1e-4)
    zc, lc = ops.actnorm_invconv(z1, bias, scales, w, sldj, length=length, channel_padding_mask=pad, ldj=ldj + l1)
    zf, lf = ops.encoder_forward_actconv(categ, u, table, prior, bias, scales, w, sldj, beta=1.1, channel_padding_mask=pad,
                                         length=length, ldj=ldj, uniform_squeeze=1e-4)
    assert torch.equal(zf, zc) and torch.equal(lf, lc)
    ops.check_flags(dev, "encoder + actconv")


def test_flow_model_fuses_the_encoding_direction_of_a_real_model(monkeypatch):
    """A set-modelling flow (encoder, then ActNorm / 1x1 conv / mixture coupling steps) evaluated from int64 categories: the
    fused kernels run in the ENCODING direction too (until round 3 the fusion test looked at the dtype of the pass's input
    and switched them all off there) and give the bits of the layer-by-layer pass."""
    from categoricalnf_amd import ops as O
    from categoricalnf_amd.experiments.set_modeling import FlowSetModeling, SetShufflingDataset
    torch.manual_seed(0)
    params = {"set_size": 16, "coupling_hidden_layers": 1, "coupling_hidden_size": 32, "coupling_num_flows": 3, "coupling_mask_ratio": 0.5,
              "coupling_num_mixtures": 8,
              "categ_encoding": {"use_dequantization": False, "use_variational": False, "use_decoder": False, "num_dimensions": 4,
                                 "flow_config": {"num_flows": 0, "hidden_layers": 2, "hidden_size": 64},
                                 "decoder_config": {"num_layers": 1, "hidden_size": 64}}}
    model = FlowSetModeling(params, SetShufflingDataset).cuda().eval()
    for p in model.parameters():
        p.data.normal_(0.0, 0.1)
    x = torch.randint(0, model.vocab_size, (64, 16), device="cuda")
    ln = torch.full((64,), 16, dtype=torch.long, device="cuda")
    u = torch.rand(64 * 16, 1, 4, device="cuda")
    calls = {"enc": 0, "three": 0, "dec": 0}
    real_enc, real_three, real_dec = O.encoder_forward_actconv, O.mixture_coupling_actconv, O.encoder_decode_actconv
    monkeypatch.setattr(O, "encoder_decode_actconv", lambda *a, **k: (calls.__setitem__("dec", calls["dec"] + 1), real_dec(*a, **k))[1])
    monkeypatch.setattr(O, "encoder_forward_actconv", lambda *a, **k: (calls.__setitem__("enc", calls["enc"] + 1), real_enc(*a, **k))[1])
    monkeypatch.setattr(O, "mixture_coupling_actconv", lambda *a, **k: (calls.__setitem__("three", calls["three"] + 1), real_three(*a, **k))[1])
    with torch.no_grad():
        z, ldj = model(x, reverse=False, length=ln, noise=u)
        zs = z + 0.1 * torch.randn_like(z)
        xr, ldjr = model(zs, reverse=True, length=ln)
        monkeypatch.setattr(O, "FUSE_LAYERS", False)
        z0, ldj0 = model(x, reverse=False, length=ln, noise=u)
        xr0, ldjr0 = model(zs, reverse=True, length=ln)
    assert calls["enc"] == 1 and calls["three"] >= 1 and calls["dec"] == 1, calls
    assert torch.equal(z, z0) and torch.equal(ldj, ldj0)
    assert torch.equal(xr, xr0) and torch.equal(ldjr, ldjr0)


@pytest.mark.parametrize("B,N,D,C,padded,use_len", [(2048, 64, 6, 16, 0, 0), (300, 16, 4, 16, 1, 1), (77, 20, 2, 3, 1, 0), (9, 5, 5, 7, 0, 1),
                                                    (2048, 16, 8, 51, 1, 1), (4, 7, 3, 9, 0, 0), (6, 12, 6, 700, 1, 1)])
def test_decode_actconv_kernel_gives_the_bits_of_the_three_layers(B, N, D, C, padded, use_len):
    """cnf_encoder_decode_actconv (inverse 1x1 conv + inverse ActNorm + arg-max decode in one launch) against the chain
    cnf_actnorm_invconv(reverse) -> cnf_encoder_decode: decoded categories and running log-det bit-identical."""
    lib, ops = _setup()
    dev = torch.device("cuda:0")
    categ, eps, table, prior, pad, ldj = _inputs(B, N, D, C, 13 + B + C, padded, dev)
    g = torch.Generator(device=dev).manual_seed(B + 2)
    zenc, _, _ = ops.encoder_forward(categ, eps, table, prior, tiled=False if ops.encoder_fused_supported(C, D) else True)
    bias, scales = torch.randn(1, 1, D, generator=g, device=dev), 0.2 * torch.randn(1, 1, D, generator=g, device=dev)
    w = torch.linalg.qr(torch.randn(D, D))[0].to(dev) + 0.05 * torch.randn(D, D, generator=g, device=dev)
    sldj = torch.randn((), generator=g, device=dev)
    length = torch.randint(max(1, N // 2), N + 1, (B,), generator=g, device=dev).float() if use_len else None
    # latents a sampling pass would hand over: the forward pair applied to the encoder's output, plus noise
    z, _ = ops.actnorm_invconv(zenc, bias, scales, w, sldj, channel_padding_mask=pad)
    z = z + 0.2 * torch.randn(B, N, D, generator=g, device=dev)
    w_inv = torch.inverse(w.double()).float()
    zc, lc = ops.actnorm_invconv(z, bias, scales, w_inv, sldj, reverse=True, length=length, channel_padding_mask=pad, ldj=ldj)
    dc = ops.encoder_decode(zc, table, prior)
    df, lf = ops.encoder_decode_actconv(z, bias, scales, w_inv, sldj, table, prior, channel_padding_mask=pad, length=length, ldj=ldj)
    assert torch.equal(df, dc) and torch.equal(lf, lc + torch.zeros_like(lc))
    ops.check_flags(dev, "decode + actconv")


# ---- round 5: the one-pass (token, class) pair kernel of the backward (cnf_encoder_forward_bwd_cpl) ---------------------------

def _bwd_call(lib, ops, which, use_cpl, categ, eps, table, prior, pad, beta, gz, gl):
    """g_table through the C ABI with cnf_set_encoder_bwd_kernel(which); use_cpl: hand over the forward's class_prob_log."""
    from categoricalnf_amd.ops import _ptr, _stream, _launch
    dev = table.device
    B, N = categ.shape
    C, D = table.shape[0], table.shape[1] // 2
    p2 = pad.reshape(B, N).contiguous() if pad is not None else None
    cpl = ops.encoder_forward(categ, eps, table, prior, beta=beta, channel_padding_mask=pad, want_class_prob=True)[2] if use_cpl else None
    ws = torch.empty(int(lib.cnf_encoder_bwd_tiled_workspace_floats(B, N, D, C)), device=dev)
    out = torch.full_like(table, float("nan"))
    lib.cnf_set_encoder_bwd_kernel(which)
    try:
        if use_cpl:
            _launch(dev, "cnf_encoder_forward_bwd_cpl", _ptr(categ), _ptr(eps), _ptr(table), _ptr(prior), _ptr(p2), float(beta), _ptr(cpl),
                    _ptr(gz), _ptr(gl), _ptr(out), _ptr(ws), B, N, D, C, float(ops.LOGISTIC_SIGMA), float(ops.LOGISTIC_LOG_SIGMA), _stream(dev))
        else:
            _launch(dev, "cnf_encoder_forward_bwd_tiled", _ptr(categ), _ptr(eps), _ptr(table), _ptr(prior), _ptr(p2), float(beta),
                    _ptr(gz), _ptr(gl), _ptr(out), _ptr(ws), B, N, D, C, float(ops.LOGISTIC_SIGMA), float(ops.LOGISTIC_LOG_SIGMA), _stream(dev))
        torch.cuda.synchronize()
    finally:
        lib.cnf_set_encoder_bwd_kernel(0)
    return out


def _oracle_table_grad(categ, eps, table, prior, pad, beta, gz, gl):
    from oracle import cnf_oracle as O
    B, N = categ.shape
    D = table.shape[1] // 2
    tc = table.double().cpu().requires_grad_()
    zo, lo, _ = O.encoder_forward(categ.cpu(), eps.double().cpu().reshape(B * N, 1, D), tc, prior.double().cpu(), beta=beta,
                                  channel_padding_mask=pad.double().cpu() if pad is not None else None)
    loss = 0.0
    if gz is not None:
        loss = loss + (zo * gz.double().cpu()).sum()
    if gl is not None:
        loss = loss + (lo * gl.double().cpu()).sum()
    loss.backward()
    return tc.grad


def _within(got, ref, rtol=2e-3, atol_rel=2e-4):
    scale = max(float(ref.abs().max()), 1.0)
    err = (got.double().cpu() - ref).abs()
    return float((err / (rtol * ref.abs() + atol_rel * scale)).max())


# (B, N, D, C): one stage and many stages per workgroup, ragged last stage, every templated D, class counts at the limits of the
# pair lanes (192 = 3 waves), a vocabulary of one class, more tokens per stage than the token wave's 64 lanes (C = 2, 3)
PAIR_SHAPES = [(64, 16, 6, 16), (33, 8, 8, 51), (6, 30, 4, 64), (5, 7, 3, 5), (4, 5, 1, 2), (2, 300, 2, 120), (7, 33, 6, 192),
               (300, 64, 6, 16), (129, 17, 6, 9), (3, 11, 4, 1), (40, 50, 6, 3), (512, 64, 6, 27), (5, 9, 6, 300), (3, 7, 4, 448),
               (700, 64, 6, 51)]


@pytest.mark.parametrize("B,N,D,C", PAIR_SHAPES)
@pytest.mark.parametrize("pad_mode", [0, 1])
def test_backward_pair_kernel_against_the_oracle(B, N, D, C, pad_mode):
    """cnf_encoder_forward_bwd_cpl / _tiled on the pair kernel (cnf_set_encoder_bwd_kernel(2)), with the forward's
    class_prob_log and with the library's own pre-pass: d loss / d class table against float64 autograd through the oracle
    (linear_encoding.py:59-106,153-174), bit-identical from run to run, and the two-pass kernels' result within the same
    tolerance; either upstream gradient may be absent."""
    lib, ops = _setup()
    dev = torch.device("cuda:0")
    categ, eps, table, prior, pad, _ = _inputs(B, N, D, C, 31 + B + C, pad_mode, dev)
    eps = eps.contiguous()
    g = torch.Generator(device=dev).manual_seed(B + N)
    gz, gl = torch.randn(B, N, D, generator=g, device=dev), torch.randn(B, generator=g, device=dev)
    ref = _oracle_table_grad(categ, eps, table, prior, pad, 1.3, gz, gl)
    forced = {}
    for knob in (2, 3):                                   # the 256-lane and the 512-lane pair workgroup
        for use_cpl in (True, False):
            got = _bwd_call(lib, ops, knob, use_cpl, categ, eps, table, prior, pad, 1.3, gz, gl)
            assert torch.isfinite(got).all()
            assert _within(got, ref) <= 1.0, (knob, use_cpl, _within(got, ref))
            assert torch.equal(got, _bwd_call(lib, ops, knob, use_cpl, categ, eps, table, prior, pad, 1.3, gz, gl))      # fixed summation order
            forced[(knob, use_cpl)] = got
    two = _bwd_call(lib, ops, 1, False, categ, eps, table, prior, pad, 1.3, gz, gl)
    assert _within(two, ref) <= 1.0
    # the default route (by shape) is one of the three
    dflt = _bwd_call(lib, ops, 0, True, categ, eps, table, prior, pad, 1.3, gz, gl)
    assert torch.equal(dflt, forced[(2, True)]) or torch.equal(dflt, forced[(3, True)]) or torch.equal(dflt, two)
    # one upstream gradient only
    for a, b_ in ((gz, None), (None, gl)):
        r1 = _oracle_table_grad(categ, eps, table, prior, pad, 0.7, a, b_)
        assert _within(_bwd_call(lib, ops, 2, True, categ, eps, table, prior, pad, 0.7, a, b_), r1) <= 1.0


def test_backward_pair_kernel_shape_limits_fall_back_to_the_two_passes():
    """Beyond the pair lanes' class count (192 / 448), where the stage does not fit 64 KB of LDS (448 classes at D = 8) or at a
    D without an instantiation the forced pair kernel is the two passes (same bits)."""
    lib, ops = _setup()
    dev = torch.device("cuda:0")
    for knob, B, N, D, C in ((2, 3, 9, 6, 193), (2, 4, 6, 5, 7), (3, 3, 9, 6, 449), (3, 2, 5, 8, 448)):
        categ, eps, table, prior, pad, _ = _inputs(B, N, D, C, 5, 1, dev)
        gz, gl = torch.randn(B, N, D, device=dev), torch.randn(B, device=dev)
        a = _bwd_call(lib, ops, knob, True, categ, eps.contiguous(), table, prior, pad, 1.0, gz, gl)
        b_ = _bwd_call(lib, ops, 1, False, categ, eps.contiguous(), table, prior, pad, 1.0, gz, gl)
        assert torch.equal(a, b_)


@pytest.mark.parametrize("B,N,D,C", [(64, 16, 6, 16), (33, 8, 8, 51), (96, 16, 6, 7)])
def test_backward_pair_kernel_log_domain_tokens(B, N, D, C):
    """The extreme inputs of test_backward_density_sum_and_its_log_domain_fallback (noise at the prior's clamp, far-apart class
    means, log-priors of -120) and the near-denormal rivals of the forward's test: tokens whose own density is outside the density
    sum's fp32 range are scored in the log domain by the pair kernel too (record flag from lp2 and the forward's log q_c)."""
    lib, ops = _setup()
    dev = torch.device("cuda:0")
    categ, eps, table, prior, pad, _ = _inputs(B, N, D, C, 77 + D + C, 1, dev)
    g = torch.Generator(device=dev).manual_seed(6)
    eps = eps.reshape(B, N, D).clone()
    eps[: B // 4] = (9.903487 / 1.81) * torch.sign(torch.randn(B // 4, N, D, generator=g, device=dev))
    eps = eps.reshape(B * N, D).contiguous()
    gz, gl = torch.randn(B, N, D, generator=g, device=dev), torch.randn(B, generator=g, device=dev)
    cases = []
    t1, p1 = table.clone(), prior.clone()
    t1[:, :D] *= 6.0
    p1[::3] = -120.0
    cases.append((t1, p1))
    t2 = table.clone()
    t2[:, :D] *= 0.5
    cases.append((t2, -83.0 + 0.7 * torch.randn(C, generator=g, device=dev)))        # every own density ~2^-120 ... 2^-135
    for tb, pr in cases:
        ref = _oracle_table_grad(categ, eps, tb, pr, pad, 1.3, gz, gl)
        for knob in (2, 3):
            for use_cpl in (True, False):
                got = _bwd_call(lib, ops, knob, use_cpl, categ, eps, tb, pr, pad, 1.3, gz, gl)
                assert torch.isfinite(got).all()
                assert _within(got, ref) <= 1.0, (knob, use_cpl, _within(got, ref))


def test_backward_pair_kernel_at_the_benchmark_size():
    """1 048 576 tokens x 16 classes (22 stages per workgroup, every workgroup slot of the chip taken): the pair kernel and the
    two passes agree to 2e-4 of the largest entry, both are reproducible, and the autograd Function (which keeps the forward's
    class_prob_log) takes the pair kernel."""
    from categoricalnf_amd import functional as Fn
    lib, ops = _setup()
    dev = torch.device("cuda:0")
    B, N, D, C = 16384, 64, 6, 16
    categ, eps, table, prior, _, _ = _inputs(B, N, D, C, 3, 0, dev)
    eps = eps.contiguous()
    g = torch.Generator(device=dev).manual_seed(9)
    gz, gl = torch.randn(B, N, D, generator=g, device=dev), torch.randn(B, generator=g, device=dev)
    pair = _bwd_call(lib, ops, 2, True, categ, eps, table, prior, None, 1.0, gz, gl)
    assert torch.equal(pair, _bwd_call(lib, ops, 2, True, categ, eps, table, prior, None, 1.0, gz, gl))
    two = _bwd_call(lib, ops, 1, False, categ, eps, table, prior, None, 1.0, gz, gl)
    scale = float(two.abs().max())
    assert float((pair - two).abs().max()) <= 2e-4 * scale
    tg = table.clone().requires_grad_()
    z, ldj, _ = Fn.EncoderForwardFn.apply(tg, categ, eps, prior, None, 1.0, False, None)
    ((z * gz).sum() + (ldj * gl).sum()).backward()
    assert torch.equal(tg.grad, pair)
    # 51 classes: the 512-lane workgroup is the library's choice there
    C = 51
    categ, eps, table, prior, _, _ = _inputs(B, N, D, C, 4, 0, dev)
    eps = eps.contiguous()
    wide = _bwd_call(lib, ops, 3, True, categ, eps, table, prior, None, 1.0, gz, gl)
    assert torch.equal(wide, _bwd_call(lib, ops, 0, True, categ, eps, table, prior, None, 1.0, gz, gl))
    two = _bwd_call(lib, ops, 1, False, categ, eps, table, prior, None, 1.0, gz, gl)
    assert float((wide - two).abs().max()) <= 2e-4 * float(two.abs().max())


def test_actconv_forward_hands_out_the_class_posterior():
    """cnf_encoder_forward_actconv_cpl: latents and log-det of cnf_encoder_forward_actconv, class_prob_log of the plain sampled
    forward — bit for bit."""
    lib, ops = _setup()
    dev = torch.device("cuda:0")
    B, N, D, C = 96, 20, 6, 16
    categ, _, table, prior, pad, ldj = _inputs(B, N, D, C, 11, 1, dev)
    g = torch.Generator(device=dev).manual_seed(2)
    u = torch.rand(B * N, D, generator=g, device=dev)
    bias, scales = 0.3 * torch.randn(D, generator=g, device=dev), 0.2 * torch.randn(D, generator=g, device=dev)
    w = torch.linalg.qr(torch.randn(D, D, generator=g, device=dev))[0].contiguous()
    sldj = torch.slogdet(w)[1].reshape(1)
    z0, l0 = ops.encoder_forward_actconv(categ, u, table, prior, bias, scales, w, sldj, beta=1.2, channel_padding_mask=pad, ldj=ldj)
    z1, l1, c1 = ops.encoder_forward_actconv(categ, u, table, prior, bias, scales, w, sldj, beta=1.2, channel_padding_mask=pad, ldj=ldj,
                                             want_class_prob=True)
    cref = ops.encoder_forward(categ, u, table, prior, beta=1.2, channel_padding_mask=pad, want_class_prob=True, uniform_squeeze=1e-4)[2]
    assert torch.equal(z0, z1) and torch.equal(l0, l1) and torch.equal(c1, cref)
