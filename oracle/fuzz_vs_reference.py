"""Seeded fuzz of the CPU oracle against the LIVE reference (phlippe/CategoricalNF imported from /root/reference, or
$CNF_REFERENCE): random shapes, masks, mixture counts, paddings, regulariser settings and train / eval modes beyond the
fixed cases of tests/golden/.  Test infrastructure, build container only (the reference cannot travel):

    python oracle/fuzz_vs_reference.py [cases] [seed]

The reference's own layer objects are called with a stand-in sub-network that returns a pre-drawn `nn_out`; the oracle gets
the same tensors.  Tolerances: 1e-6 absolute / relative in fp32 (same op order and dtypes as the reference), 1e-5 on the
bisection inverse.  Prints one line per layer family and `FUZZ OK <n>`."""
import contextlib
import io
import os
import sys

import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("CNF_REFERENCE", "/root/reference")
sys.path.insert(0, ROOT)
sys.path.insert(1, REF)
os.environ.setdefault("MPLBACKEND", "Agg")
from oracle import cnf_oracle as O                                                     # noqa: E402
with contextlib.redirect_stdout(io.StringIO()):
    from layers.flows.coupling_layer import CouplingLayer                              # noqa: E402
    from layers.flows.mixture_cdf_layer import MixtureCDFCoupling                      # noqa: E402
    from layers.flows.autoregressive_coupling import AutoregressiveMixtureCDFCoupling  # noqa: E402
    from layers.flows.activation_normalization import ActNormFlow, ExtActNormFlow      # noqa: E402
    from layers.flows.permutation_layers import InvertibleConv                         # noqa: E402
    from layers.flows.distributions import LogisticDistribution                        # noqa: E402
    from general.mutils import create_channel_mask                                     # noqa: E402
    from layers.categorical_encoding.linear_encoding import LinearCategoricalEncoding  # noqa: E402
    from layers.flows.sigmoid_layer import SigmoidFlow                                 # noqa: E402
assert CouplingLayer.__module__ == "layers.flows.coupling_layer" and "categoricalnf_amd" not in sys.modules


class Inject(nn.Module):
    def __init__(self):
        super().__init__()
        self.value = None

    def forward(self, x=None, **kwargs):
        return self.value


def close(a, b, tol=1e-6, what=""):
    a, b = a.detach().double(), b.detach().double()
    err = (a - b).abs()
    bound = tol + tol * b.abs()
    assert bool((err <= bound).all()), "%s: max error %.3g (allowed %.1g abs/rel)" % (what, float(err.max()), tol)


def quiet(fn):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn()


def main(cases=60, seed=0):
    g = torch.Generator().manual_seed(seed)
    rng = np.random.RandomState(seed)
    ri = lambda lo, hi: int(rng.randint(lo, hi + 1))
    counts = {}

    def lengths(B, N):
        ln = torch.from_numpy(rng.randint(max(1, N // 3), N + 1, size=B)).long()
        ln[ri(0, B - 1)] = N
        return ln

    for _ in range(cases):
        B, N, D = ri(1, 6), ri(1, 20), ri(1, 8)
        # ---- affine coupling (coupling_layer.py:42-98) ----
        kind = "chess" if (D == 1 or rng.rand() < 0.2) else "channel"
        mask = CouplingLayer.create_chess_mask() if kind == "chess" else CouplingLayer.create_channel_mask(
            D, ratio=float(rng.choice([0.25, 0.5, 0.75])), mask_floor=bool(rng.rand() < 0.5))
        if rng.rand() < 0.3:
            mask = 1 - mask
        layer = quiet(lambda: CouplingLayer(c_in=D, mask=mask, model_func=lambda c_out: Inject()))
        sf = 0.6 * torch.randn(D, generator=g)
        layer.scaling_factor.data = sf.clone()
        z = torch.randn(B, N, D, generator=g)
        nn_out = 1.5 * torch.randn(B, N, 2 * D, generator=g)
        layer.nn.value = nn_out
        ldj_in = torch.randn(B, generator=g)
        zf, lf = layer(z, ldj=ldj_in.clone(), reverse=False)
        zr, lr = layer(zf, ldj=None, reverse=True)
        of, olf = O.affine_coupling(z, nn_out, mask, sf, reverse=False, ldj=ldj_in)
        orr, olr = O.affine_coupling(zf, nn_out, mask, sf, reverse=True)
        close(of, zf, what="affine z"); close(olf, lf, what="affine ldj")
        close(orr, zr, what="affine inverse z"); close(olr, lr, what="affine inverse ldj")
        counts["affine coupling"] = counts.get("affine coupling", 0) + 1

        # ---- mixture-CDF coupling (mixture_cdf_layer.py:45-276, autoregressive_coupling.py:25-47) ----
        K = int(rng.choice([1, 2, 4, 8, 16, 27]))
        Bm, Nm, Dm = ri(1, 4), ri(1, 10), ri(1, 6)
        zt = float(rng.choice([1.0, 1.0, 3.0, 8.0])) * torch.randn(Bm, Nm, Dm, generator=g)
        nn_m = 0.8 * torch.randn(Bm, Nm, Dm * (2 + 3 * K), generator=g)
        sfm, msf = 0.4 * torch.randn(Dm, generator=g), 0.4 * torch.randn(Dm, K, generator=g)
        reg_max, reg_factor = (3.5, float(rng.choice([1, 2]))) if rng.rand() < 0.4 else (-1, 1)
        training = bool(rng.rand() < 0.5)
        if rng.rand() < 0.25:
            lay = quiet(lambda: AutoregressiveMixtureCDFCoupling(c_in=Dm, model_func=lambda c_out: Inject(), num_mixtures=K))
            lay.scaling_factor.data, lay.mixture_scaling_factor.data = sfm.clone(), msf.clone()
            lay.nn.value = nn_m
            lay.train(training)
            zf, lf = lay(zt, reverse=False)
            of, olf, _ = O.mixture_coupling(zt, nn_m, None, K, sfm, msf, reverse=False, is_training=training)
            close(of, zf, what="AR mixture z"); close(olf, lf, tol=2e-6, what="AR mixture ldj")
            counts["autoregressive mixture coupling"] = counts.get("autoregressive mixture coupling", 0) + 1
        else:
            mk = CouplingLayer.create_chess_mask() if (Dm == 1 or rng.rand() < 0.2) else CouplingLayer.create_channel_mask(Dm)
            padded = bool(rng.rand() < 0.5)
            pad = create_channel_mask(lengths(Bm, Nm), max_len=Nm) if padded else None
            lay = quiet(lambda: MixtureCDFCoupling(c_in=Dm, mask=mk, model_func=lambda c_out: Inject(), num_mixtures=K,
                                                   regularizer_max=reg_max, regularizer_factor=reg_factor))
            lay.scaling_factor.data, lay.mixture_scaling_factor.data = sfm.clone(), msf.clone()
            lay.nn.value = nn_m
            lay.train(training)
            kw_m = dict(channel_padding_mask=pad) if padded else {}
            zf, lf, det = lay(zt, reverse=False, **kw_m)
            zr, lr, _ = lay(zf, reverse=True, **kw_m)
            okw = dict(channel_padding_mask=pad, reg_max=reg_max, reg_factor=reg_factor, is_training=training)
            of, olf, oreg = O.mixture_coupling(zt, nn_m, mk, K, sfm, msf, reverse=False, **okw)
            orr, olr, _ = O.mixture_coupling(zf, nn_m, mk, K, sfm, msf, reverse=True, **okw)
            close(of, zf, what="mixture z"); close(olf, lf, tol=2e-6, what="mixture ldj")
            close(oreg, det["regularizer_ldj"], tol=2e-6, what="mixture regulariser")
            close(orr, zr, tol=1e-5, what="mixture inverse z"); close(olr, lr, tol=1e-5, what="mixture inverse ldj")
            counts["mixture-CDF coupling"] = counts.get("mixture-CDF coupling", 0) + 1

        # ---- ActNorm / 1x1 convolution / logistic prior ----
        an = ActNormFlow(c_in=D, data_init=False)
        an.bias.data, an.scales.data = torch.randn(1, 1, D, generator=g), 0.5 * torch.randn(1, 1, D, generator=g)
        ln = lengths(B, N)
        pad = create_channel_mask(ln, max_len=N)
        mode = int(rng.randint(0, 3))
        kw = {} if mode == 0 else (dict(length=ln) if mode == 1 else dict(length=ln, channel_padding_mask=pad))
        zf, lf = an(z, ldj=ldj_in.clone(), reverse=False, **kw)
        of, olf = O.actnorm(z, an.bias.data, an.scales.data, reverse=False, ldj=ldj_in, **kw)
        close(of, zf, what="ActNorm z"); close(olf, lf, what="ActNorm ldj")
        zr, lr = an(zf, ldj=None, reverse=True, **kw)
        orr, olr = O.actnorm(zf, an.bias.data, an.scales.data, reverse=True, **kw)
        close(orr, zr, what="ActNorm inverse z"); close(olr, lr, what="ActNorm inverse ldj")
        counts["ActNorm"] = counts.get("ActNorm", 0) + 1

        # the reference's dense 2 x 2 initialisation is a float64 rotation matrix that its own forward cannot multiply with
        # float activations (permutation_layers.py:30, :119), so two channels are fuzzed in the LU parametrisation only
        use_lu = bool(rng.rand() < 0.5) or D == 2
        conv = quiet(lambda: InvertibleConv(c_in=D, LU_decomposed=use_lu))
        conv.eval()
        with torch.no_grad():
            w, sldj = conv._get_weight(device_name="cpu", inverse=False)
            zf, lf = conv(z, ldj=ldj_in.clone(), reverse=False, **kw)
            zr, lr = conv(zf, ldj=None, reverse=True, **kw)
        of, olf = O.invconv(z, w, sldj, reverse=False, ldj=ldj_in, **kw)
        orr, olr = O.invconv(zf, w, sldj, reverse=True, **kw)
        close(of, zf, tol=2e-6, what="1x1 conv z"); close(olf, lf, tol=2e-6, what="1x1 conv ldj")
        close(orr, zr, tol=1e-5, what="1x1 conv inverse z"); close(olr, lr, tol=2e-6, what="1x1 conv inverse ldj")
        counts["invertible 1x1 convolution"] = counts.get("invertible 1x1 convolution", 0) + 1

        prior = LogisticDistribution(mu=0.0, sigma=1.0)
        x = float(rng.choice([1.0, 5.0, 30.0])) * torch.randn(B, N, D, generator=g)
        close(O.logistic_log_prob(x), prior.log_prob(x), what="logistic log-prob")
        counts["logistic prior"] = counts.get("logistic prior", 0) + 1
        # ---- gradients: the reference's autograd through its own layers vs autograd through the oracle (what the GPU
        #      suite differentiates when it checks the HIP backward kernels on seeded inputs) ----
        wz, wl = torch.randn(B, N, D, generator=g), torch.randn(B, generator=g)
        nn_g = nn_out.clone().requires_grad_()
        layer.nn.value = nn_g
        layer.scaling_factor.grad = None
        zf, lf = layer(z, ldj=None, reverse=False)
        ((zf * wz).sum() + (lf * wl).sum()).backward()
        nn_o, sf_o = nn_out.clone().requires_grad_(), sf.clone().requires_grad_()
        of, olf = O.affine_coupling(z, nn_o, mask, sf_o, reverse=False)
        ((of * wz).sum() + (olf * wl).sum()).backward()
        close(nn_o.grad, nn_g.grad, tol=2e-6, what="affine d/d nn_out")
        close(sf_o.grad, layer.scaling_factor.grad, tol=1e-5, what="affine d/d scaling_factor")
        layer.nn.value = nn_out
        if type(lay) is MixtureCDFCoupling:
            wzm, wlm = torch.randn(Bm, Nm, Dm, generator=g), torch.randn(Bm, generator=g)
            nn_g = nn_m.clone().requires_grad_()
            lay.nn.value = nn_g
            lay.scaling_factor.grad = lay.mixture_scaling_factor.grad = None
            zf, lf, _ = lay(zt, reverse=False, **kw_m)
            ((zf * wzm).sum() + (lf * wlm).sum()).backward()
            nn_o = nn_m.clone().requires_grad_()
            sf_o, msf_o = sfm.clone().requires_grad_(), msf.clone().requires_grad_()
            of, olf, _ = O.mixture_coupling(zt, nn_o, mk, K, sf_o, msf_o, reverse=False, **okw)
            ((of * wzm).sum() + (olf * wlm).sum()).backward()
            scale = float(nn_g.grad.abs().max()) + 1e-12
            close(nn_o.grad / scale, nn_g.grad / scale, tol=1e-5, what="mixture d/d nn_out")
            close(sf_o.grad, lay.scaling_factor.grad, tol=1e-4, what="mixture d/d scaling_factor")
            close(msf_o.grad, lay.mixture_scaling_factor.grad, tol=1e-4, what="mixture d/d mixture_scaling_factor")
            counts["gradients (affine + mixture)"] = counts.get("gradients (affine + mixture)", 0) + 1

        # ---- ExtActNorm with an injected predictor output (activation_normalization.py:116-144) ----
        ext = quiet(lambda: ExtActNormFlow(c_in=D, net=Inject()))
        nn_e = 0.7 * torch.randn(B, N, 2 * D, generator=g)
        ext.pred_net.value = nn_e
        pkw = dict(channel_padding_mask=pad) if rng.rand() < 0.5 else {}
        zf, lf = ext(z, ldj=ldj_in.clone(), reverse=False, ext_input=z, **pkw)
        of, olf = O.ext_actnorm(z, nn_e, reverse=False, ldj=ldj_in, **pkw)
        close(of, zf, what="ExtActNorm z"); close(olf, lf, tol=2e-6, what="ExtActNorm ldj")
        zr, lr = ext(zf, ldj=None, reverse=True, ext_input=z, **pkw)
        orr, olr = O.ext_actnorm(zf, nn_e, reverse=True, **pkw)
        close(orr, zr, what="ExtActNorm inverse z"); close(olr, lr, tol=2e-6, what="ExtActNorm inverse ldj")
        counts["ExtActNorm"] = counts.get("ExtActNorm", 0) + 1

        # ---- SigmoidFlow in both orientations (sigmoid_layer.py:24-47; the layer XORs its own flag with the call's) ----
        for own in (False, True):
            sg = SigmoidFlow(reverse=own)
            zs = 2.5 * torch.randn(B, N, D, generator=g)
            a, la = sg(zs, ldj=ldj_in.clone(), reverse=own)                  # net direction: real line -> (0, 1)
            oa, ola = O.sigmoid_flow(zs, reverse=False, ldj=ldj_in)
            close(oa, a, what="sigmoid"); close(ola, la, tol=2e-6, what="sigmoid ldj")
            us = torch.rand(B, N, D, generator=g)
            b, lb = sg(us, ldj=None, reverse=not own)                        # net direction: (0, 1) -> real line
            ob, olb = O.sigmoid_flow(us, reverse=True)
            close(ob, b, tol=2e-6, what="logit"); close(olb, lb, tol=2e-6, what="logit ldj")
        counts["SigmoidFlow"] = counts.get("SigmoidFlow", 0) + 1

        # ---- mixture-model categorical encoder (linear_encoding.py:59-196): forward with the CPU generator's noise,
        #      posterior over all classes, arg-max decode ----
        C = int(rng.choice([1, 2, 3, 5, 9, 16, 40]))
        De = ri(1, 6)
        use_prior = bool(rng.rand() < 0.5)
        tseed = ri(0, 10 ** 6)
        torch.manual_seed(tseed)
        np.random.seed(tseed % (2 ** 31))
        cp = torch.randn(C) if use_prior else None
        enc = quiet(lambda: LinearCategoricalEncoding(num_dimensions=De, flow_config={"num_flows": 0}, vocab_size=C,
                                                      category_prior=cp))
        lin = enc.flow_layers[0].pred_net.layer
        lin.weight.data[De:, :] = 0.2 * torch.randn(De, lin.weight.shape[1])
        lin.bias.data = 0.1 * torch.randn(2 * De)
        enc.train(bool(rng.rand() < 0.5))
        cat = torch.randint(0, C, (B, N))
        beta = float(rng.choice([1.0, 1.0, 1.5, 2.0]))
        ekw = dict(channel_padding_mask=pad) if rng.rand() < 0.5 else {}
        torch.manual_seed(tseed + 1)
        u = torch.rand(B * N, 1, De)
        torch.manual_seed(tseed + 1)
        with torch.no_grad():
            ze, le, _ = enc(cat, reverse=False, beta=beta, **ekw)
            dec, _, _ = enc(ze, reverse=True)
            probe = ze + 0.7 * torch.randn(ze.shape)
            dec_p, _, _ = enc(probe, reverse=True)
            table = enc.flow_layers[0].pred_net(enc.embed_layer.weight)
        oz, ol, _ = O.encoder_forward(cat, O.logistic_from_uniform(u), table, enc.category_prior, beta=beta,
                                      channel_padding_mask=ekw.get("channel_padding_mask"))
        # z = (noise + bias) * exp(scale): the logistic noise reaches |9.2| (one fp32 ulp there is 1e-6), and the sum
        # with a bias of the opposite sign keeps that absolute error on a small result
        close(oz, ze, tol=4e-6, what="encoder z"); close(ol, le, tol=5e-6, what="encoder ldj")
        assert torch.equal(O.encoder_decode(ze, table, enc.category_prior)[0], dec), "encoder decode"
        assert torch.equal(O.encoder_decode(probe, table, enc.category_prior)[0], dec_p), "encoder decode (perturbed)"
        counts["categorical encoder (fwd + decode)"] = counts.get("categorical encoder (fwd + decode)", 0) + 1
    for k, v in counts.items():
        print("%-34s %4d cases equal to the reference" % (k, v))
    print("FUZZ OK %d" % sum(counts.values()))


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 60, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
