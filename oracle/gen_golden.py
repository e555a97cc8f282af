"""Generate tests/golden/*.npz by running the REAL reference (phlippe/CategoricalNF) on seeded inputs.

Runs only in the build container, where the reference is mounted read-only at /root/reference:

    PYTHONPATH=/root/reference MPLBACKEND=Agg PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden.py

The reference's own modules are imported and called (nothing of their source is copied); the
coupling subnets are replaced by a stub that returns a pre-drawn ``nn_out`` so that the layer
arithmetic is captured in isolation.  Each .npz holds, per case, the inputs, the parameters and the
reference outputs plus a JSON ``meta`` entry describing the cases.  The fixtures are data only.
"""
import copy
import io
import json
import os
import sys
import contextlib

import numpy as np
import torch
import torch.nn as nn

REF = os.environ.get("CNF_REFERENCE", "/root/reference")
sys.path.insert(0, REF)
os.environ.setdefault("MPLBACKEND", "Agg")

with contextlib.redirect_stdout(io.StringIO()):
    from layers.flows.coupling_layer import CouplingLayer
    from layers.flows.mixture_cdf_layer import MixtureCDFCoupling
    from layers.flows.autoregressive_coupling import AutoregressiveMixtureCDFCoupling
    from layers.flows.activation_normalization import ActNormFlow, ExtActNormFlow
    from layers.flows.permutation_layers import InvertibleConv
    from layers.flows.distributions import LogisticDistribution
    from layers.flows.sigmoid_layer import SigmoidFlow
    from layers.flows.flow_model import FlowModel
    from layers.categorical_encoding.linear_encoding import LinearCategoricalEncoding
    from layers.categorical_encoding.variational_dequantization import VariationalDequantization
    from general.mutils import create_channel_mask

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
os.makedirs(OUT, exist_ok=True)


class Inject(nn.Module):
    """Stand-in coupling subnet: ignores its input and returns a fixed tensor."""

    def __init__(self):
        super().__init__()
        self.value = None

    def forward(self, x=None, **kwargs):
        return self.value


def npy(t):
    return t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)


def save(name, cases):
    flat, meta = {}, []
    for i, c in enumerate(cases):
        m = dict(c.get("meta", {}))
        m["keys"] = []
        for k, v in c.items():
            if k == "meta":
                continue
            flat["c%d_%s" % (i, k)] = npy(v)
            m["keys"].append(k)
        meta.append(m)
    flat["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **flat)
    print("%-28s %3d cases  %7.1f kB" % (name, len(cases), os.path.getsize(path) / 1e3))


def lengths(B, N, g):
    ln = torch.randint(max(1, N // 2), N + 1, (B,), generator=g)
    ln[0] = N
    return ln


# ------------------------------------------------------------------------------------------
def gen_affine():
    cases = []
    g = torch.Generator().manual_seed(42)
    cfgs = [(4, 5, 2, "channel", 0), (3, 7, 4, "channel", 0), (5, 8, 6, "channel", 0),
            (6, 16, 3, "channel", 0), (4, 5, 1, "chess", 0), (4, 6, 1, "chess", 1),
            (64, 1, 4, "channel", 0), (2, 64, 6, "channel", 0), (3, 9, 5, "channel", 0)]
    for (B, N, D, kind, flip) in cfgs:
        if kind == "channel":
            mask = CouplingLayer.create_channel_mask(D)
        else:
            mask = CouplingLayer.create_chess_mask()
            if flip:
                mask = 1 - mask
        with contextlib.redirect_stdout(io.StringIO()):
            layer = CouplingLayer(c_in=D, mask=mask, model_func=lambda c_out: Inject())
        sf = 0.6 * torch.randn(D, generator=g)
        layer.scaling_factor.data = sf.clone()
        z = torch.randn(B, N, D, generator=g)
        nn_out = 1.5 * torch.randn(B, N, 2 * D, generator=g)
        layer.nn.value = nn_out
        ldj_in = torch.randn(B, generator=g)
        zf, lf = layer(z, ldj=ldj_in.clone(), reverse=False)
        zr, lr = layer(zf, ldj=None, reverse=True)
        s, t = CouplingLayer.get_coup_params(nn_out, layer._prepare_mask(mask, z), scaling_factor=sf)
        s_raw, t_raw = CouplingLayer.get_coup_params(nn_out, layer._prepare_mask(mask, z), scaling_factor=None)
        cases.append(dict(meta=dict(B=B, N=N, D=D, mask_kind=kind, flip=flip),
                          z=z, nn_out=nn_out, scaling_factor=sf, mask=mask, ldj_in=ldj_in,
                          z_fwd=zf, ldj_fwd=lf, z_rev=zr, ldj_rev=lr, s=s, t=t, s_nofac=s_raw, t_nofac=t_raw))
    save("affine_coupling", cases)


def gen_mixture():
    cases = []
    g = torch.Generator().manual_seed(43)
    cfgs = [
        # B, N, D, K, mask, reg_max, reg_factor, training, padded, tail_scale
        (4, 6, 2, 4, "channel", -1, 1, True, False, 1.0),
        (3, 16, 4, 8, "channel", -1, 1, False, False, 1.0),
        (3, 9, 6, 16, "channel", 3.5, 2, True, True, 1.0),
        (3, 9, 6, 16, "channel", 3.5, 2, False, True, 1.0),
        (2, 11, 3, 51, "none", -1, 1, True, False, 1.0),
        (4, 7, 1, 8, "chess", -1, 1, True, True, 1.0),
        (4, 8, 4, 8, "channel", 3.5, 2, True, False, 8.0),
        (4, 8, 4, 8, "channel", -1, 1, False, False, 8.0),
        (2, 5, 2, 8, "channel", 3.5, 1, True, True, 3.0),
        (5, 10, 3, 4, "channel", -1, 1, True, True, 1.0),
    ]
    for (B, N, D, K, kind, reg_max, reg_factor, training, padded, tail) in cfgs:
        P = 2 + 3 * K
        z = tail * torch.randn(B, N, D, generator=g)
        nn_out = 0.8 * torch.randn(B, N, D * P, generator=g)
        sf = 0.4 * torch.randn(D, generator=g)
        msf = 0.4 * torch.randn(D, K, generator=g)
        ln = lengths(B, N, g) if padded else None
        pad = create_channel_mask(ln, max_len=N) if padded else None
        meta = dict(B=B, N=N, D=D, K=K, mask_kind=kind, reg_max=reg_max, reg_factor=reg_factor,
                    training=training, padded=padded, tail=tail)
        if kind == "none":
            with contextlib.redirect_stdout(io.StringIO()):
                layer = AutoregressiveMixtureCDFCoupling(c_in=D, model_func=lambda c_out: Inject(), num_mixtures=K)
            layer.scaling_factor.data, layer.mixture_scaling_factor.data = sf.clone(), msf.clone()
            layer.nn.value = nn_out
            layer.train(training)
            zf, lf = layer(z, reverse=False)
            t, log_s, log_pi, mu, ls = MixtureCDFCoupling.get_mixt_params(nn_out, None, K, sf, msf)
            zr64, lr64 = MixtureCDFCoupling.run_with_params(zf.double(), t, log_s, log_pi, mu, ls, reverse=True)
            cases.append(dict(meta=meta, z=z, nn_out=nn_out, scaling_factor=sf, mixture_scaling_factor=msf,
                              z_fwd=zf, ldj_fwd=lf, z_rev=zr64.float(), ldj_rev=lr64.float(),
                              p_t=t, p_log_s=log_s, p_log_pi=log_pi, p_mixt_t=mu, p_mixt_log_s=ls))
            continue
        mask = CouplingLayer.create_channel_mask(D) if kind == "channel" else CouplingLayer.create_chess_mask()
        with contextlib.redirect_stdout(io.StringIO()):
            layer = MixtureCDFCoupling(c_in=D, mask=mask, model_func=lambda c_out: Inject(), num_mixtures=K,
                                       regularizer_max=reg_max, regularizer_factor=reg_factor)
        layer.scaling_factor.data, layer.mixture_scaling_factor.data = sf.clone(), msf.clone()
        layer.nn.value = nn_out
        layer.train(training)
        kw = dict(channel_padding_mask=pad) if padded else {}
        zf, lf, det = layer(z, reverse=False, **kw)
        zr, lr, _ = layer(zf, reverse=True, **kw)
        t, log_s, log_pi, mu, ls = MixtureCDFCoupling.get_mixt_params(nn_out, layer._prepare_mask(mask, z), K, sf, msf)
        c = dict(meta=meta, z=z, nn_out=nn_out, scaling_factor=sf, mixture_scaling_factor=msf, mask=mask,
                 z_fwd=zf, ldj_fwd=lf, z_rev=zr, ldj_rev=lr, reg_ldj=det["regularizer_ldj"],
                 p_t=t, p_log_s=log_s, p_log_pi=log_pi, p_mixt_t=mu, p_mixt_log_s=ls)
        if padded:
            c["length"] = ln
            c["pad"] = pad
        cases.append(c)
    # the reference's own __main__ demo (mixture_cdf_layer.py:279-302): linear subnet, seed 42
    torch.manual_seed(42)
    c_in, K, hidden = 4, 10, 128
    with contextlib.redirect_stdout(io.StringIO()):
        layer = MixtureCDFCoupling(c_in=c_in, mask=CouplingLayer.create_channel_mask(c_in),
                                   model_func=lambda c_out: nn.Sequential(nn.Linear(c_in, hidden), nn.ReLU(),
                                                                          nn.Linear(hidden, c_out)),
                                   block_type="Linear net", num_mixtures=K)
    x = torch.randn(size=(8, 16, c_in))
    zf, lf, _ = layer(z=x, reverse=False)
    zr, lr, _ = layer(z=zf, reverse=True)
    mask = layer.mask
    nn_f = layer.nn(x * mask)
    nn_r = layer.nn(zf * mask)
    cases.append(dict(meta=dict(B=8, N=16, D=c_in, K=K, mask_kind="channel", reg_max=-1, reg_factor=1, training=True,
                                padded=False, tail=1.0, demo=True,
                                demo_max_recon_err=float((x - zr).abs().max()),
                                demo_max_ldj_err=float((lf + lr).abs().max())),
                      z=x, nn_out=nn_f, nn_out_rev=nn_r, scaling_factor=layer.scaling_factor.data,
                      mixture_scaling_factor=layer.mixture_scaling_factor.data, mask=mask,
                      z_fwd=zf, ldj_fwd=lf, z_rev=zr, ldj_rev=lr))
    save("mixture_coupling", cases)


def gen_actnorm():
    cases = []
    g = torch.Generator().manual_seed(44)
    for (B, N, D, mode) in [(4, 6, 3, "length"), (4, 6, 3, "none"), (5, 9, 6, "mask"), (3, 16, 4, "length_mask"),
                            (8, 1, 2, "none"), (2, 7, 1, "length_mask")]:
        layer = ActNormFlow(D)
        layer.bias.data = torch.randn(1, 1, D, generator=g)
        layer.scales.data = 0.5 * torch.randn(1, 1, D, generator=g)
        z = torch.randn(B, N, D, generator=g)
        ln = lengths(B, N, g)
        pad = create_channel_mask(ln, max_len=N)
        kw = {}
        if "length" in mode:
            kw["length"] = ln
        if "mask" in mode:
            kw["channel_padding_mask"] = pad
        ldj_in = torch.randn(B, generator=g)
        zf, lf = layer(z, ldj=ldj_in.clone(), reverse=False, **kw)
        zr, lr = layer(zf, ldj=None, reverse=True, **kw)
        # data-dependent init on a fresh layer
        init = ActNormFlow(D)
        with contextlib.redirect_stdout(io.StringIO()):
            init.data_init_forward(z, channel_padding_mask=(pad.expand(-1, -1, D) if "mask" in mode else None))
        cases.append(dict(meta=dict(B=B, N=N, D=D, mode=mode), z=z, bias=layer.bias.data, scales=layer.scales.data,
                          length=ln, pad=pad, ldj_in=ldj_in, z_fwd=zf, ldj_fwd=lf, z_rev=zr, ldj_rev=lr,
                          init_bias=init.bias.data, init_scales=init.scales.data))
    save("actnorm", cases)

    cases = []
    for (B, N, D, padded) in [(6, 1, 3, False), (4, 5, 4, True), (3, 8, 6, False), (16, 1, 2, False)]:
        net = Inject()
        layer = ExtActNormFlow(D, net=net)
        z = torch.randn(B, N, D, generator=g)
        nn_out = torch.randn(B, N, 2 * D, generator=g)
        net.value = nn_out
        ln = lengths(B, N, g)
        pad = create_channel_mask(ln, max_len=N)
        kw = dict(channel_padding_mask=pad) if padded else {}
        ldj_in = torch.randn(B, generator=g)
        zf, lf = layer(z, ldj_in.clone(), ext_input=z, reverse=False, **kw)
        zr, lr = layer(zf, None, ext_input=z, reverse=True, **kw)
        cases.append(dict(meta=dict(B=B, N=N, D=D, padded=padded), z=z, nn_out=nn_out, pad=pad, ldj_in=ldj_in,
                          z_fwd=zf, ldj_fwd=lf, z_rev=zr, ldj_rev=lr))
    save("ext_actnorm", cases)


def gen_invconv():
    cases = []
    g = torch.Generator().manual_seed(45)
    np.random.seed(45)
    # NB dense D=2 is unusable in the reference itself (fp64 rotation init -> dtype error in matmul)
    for (B, N, D, lu, mode) in [(4, 6, 2, True, "none"), (4, 6, 3, False, "none"), (3, 7, 3, True, "length"),
                                (3, 5, 4, True, "length_mask"), (2, 9, 6, True, "none"), (2, 9, 6, False, "length_mask"),
                                (5, 4, 8, True, "length"), (7, 1, 4, True, "none")]:
        layer = InvertibleConv(D, LU_decomposed=lu)
        z = torch.randn(B, N, D, generator=g)
        ln = lengths(B, N, g)
        pad = create_channel_mask(ln, max_len=N)
        kw = {}
        if "length" in mode:
            kw["length"] = ln
        if "mask" in mode:
            kw["channel_padding_mask"] = pad
        ldj_in = torch.randn(B, generator=g)
        layer.train()
        zf, lf = layer(z, ldj=ldj_in.clone(), reverse=False, **kw)
        zr, lr = layer(zf, ldj=None, reverse=True, **kw)
        w_train, sldj_train = layer._get_weight("cpu")
        layer.eval()
        zf_e, lf_e = layer(z, ldj=ldj_in.clone(), reverse=False, **kw)
        zr_e, lr_e = layer(zf_e, ldj=None, reverse=True, **kw)
        c = dict(meta=dict(B=B, N=N, D=D, lu=lu, mode=mode), z=z, length=ln, pad=pad, ldj_in=ldj_in,
                 weight=w_train, sldj=sldj_train, inv_weight=layer.eval_dict["cpu"]["inv_weight"],
                 z_fwd=zf, ldj_fwd=lf, z_rev=zr, ldj_rev=lr, z_fwd_eval=zf_e, ldj_fwd_eval=lf_e,
                 z_rev_eval=zr_e, ldj_rev_eval=lr_e)
        for k, v in layer.state_dict().items():
            c["sd_" + k] = v
        cases.append(c)
    save("invconv", cases)


def gen_prior():
    cases = []
    prior = LogisticDistribution(mu=0.0, sigma=1.0)
    g = torch.Generator().manual_seed(46)
    x = torch.cat([3 * torch.randn(200, generator=g), torch.tensor([0.0, 25.0, -25.0, 60.0, -60.0, 1e-8])])
    cases.append(dict(meta=dict(kind="log_prob", sigma=prior.sigma, log_sigma=float(prior.log_sigma)),
                      x=x, log_prob=prior.log_prob(x)))
    for seed, shape in [(7, (6, 1, 3)), (8, (5, 1, 1)), (9, (4, 1, 6))]:
        torch.manual_seed(seed)
        u = torch.rand(shape)                      # == Uniform(0,1).sample(shape) on the CPU generator
        torch.manual_seed(seed)
        s = prior.sample(shape=shape)
        cases.append(dict(meta=dict(kind="sample", seed=seed, shape=list(shape)), u=u, sample=s,
                          log_prob=prior.log_prob(s)))
    # NLL assembly as in experiments/set_modeling/task.py:96-118
    B, N, D = 6, 8, 4
    z = 1.5 * torch.randn(B, N, D, generator=g)
    ldj = torch.randn(B, generator=g)
    ln = lengths(B, N, g)
    pad = create_channel_mask(ln, max_len=N)
    neglog = -(prior.log_prob(z) * pad).sum(dim=[1, 2])
    nll = (-ldj) / ln.float() + neglog / ln.float()
    cases.append(dict(meta=dict(kind="nll", B=B, N=N, D=D), z=z, ldj=ldj, length=ln, pad=pad,
                      neglog=neglog, nll=nll, nll_mean=nll.mean(), bpd=np.log2(np.exp(1)) * nll.mean()))
    save("prior", cases)


def gen_encoder():
    cases = []
    for i, (B, N, D, C, beta, prior, padded, training) in enumerate([
            (3, 6, 3, 4, 1.0, False, False, True), (4, 5, 2, 1, 1.0, False, False, False),
            (4, 5, 2, 2, 1.0, False, True, False), (3, 7, 4, 3, 2.0, True, True, True),
            (2, 9, 6, 9, 1.0, True, False, False), (4, 16, 4, 16, 1.0, False, False, False),
            (2, 11, 3, 51, 1.0, True, True, False), (5, 3, 1, 5, 1.0, False, False, False)]):
        torch.manual_seed(100 + i)
        np.random.seed(100 + i)
        cp = torch.randn(C) if prior else None
        with contextlib.redirect_stdout(io.StringIO()):
            enc = LinearCategoricalEncoding(num_dimensions=D, flow_config={"num_flows": 0}, vocab_size=C,
                                            category_prior=cp)
        # give the scale half of the predictor non-trivial values (it is zero at init, help_layers.py:62-66)
        lin = enc.flow_layers[0].pred_net.layer
        lin.weight.data[D:, :] = 0.2 * torch.randn(D, lin.weight.shape[1])
        lin.bias.data = 0.1 * torch.randn(2 * D)
        enc.train(training)
        cat = torch.randint(0, C, (B, N))
        ln = lengths(B, N, torch.Generator().manual_seed(i))
        pad = create_channel_mask(ln, max_len=N)
        kw = dict(channel_padding_mask=pad) if padded else {}
        torch.manual_seed(500 + i)
        u = torch.rand(B * N, 1, D)
        torch.manual_seed(500 + i)
        z, ldj, det = enc(cat, reverse=False, beta=beta, **kw)
        dec, _, _ = enc(z, reverse=True)
        z_probe = z + 0.7 * torch.randn(z.shape)
        dec_probe, _, _ = enc(z_probe, reverse=True)
        table = enc.flow_layers[0].pred_net(enc.embed_layer.weight)     # [C, 2D]
        c = dict(meta=dict(B=B, N=N, D=D, C=C, beta=beta, prior=prior, padded=padded, training=training),
                 categ=cat, u=u, table=table, category_prior=enc.category_prior, pad=pad, z=z, ldj=ldj,
                 decoded=dec, z_probe=z_probe, decoded_probe=dec_probe)
        for k, v in det.items():
            c["detail_" + k] = v
        for k, v in enc.state_dict().items():
            c["sd_" + k] = v
        cases.append(c)
    save("encoder", cases)


def gen_encoder_large_vocab():
    """Vocabularies beyond the drop-in's LDS-resident class table (its class-tiled kernels, with and without class
    splits): the reference's own encoder — latents, log-det, decoded classes and, through its autograd, the parameter
    gradients.  Small embedding width keeps the fixture small."""
    cases = []
    for i, (B, N, D, C, beta, prior, padded) in enumerate([(2, 5, 6, 300, 1.0, True, True), (2, 4, 4, 1100, 1.5, True, False),
                                                            (3, 3, 3, 1300, 1.0, False, True)]):
        torch.manual_seed(1100 + i)
        np.random.seed(1100 + i)
        cp = torch.randn(C) if prior else None
        with contextlib.redirect_stdout(io.StringIO()):
            enc = LinearCategoricalEncoding(num_dimensions=D, flow_config={"num_flows": 0}, vocab_size=C, category_prior=cp,
                                            default_embed_layer_dims=8)
        lin = enc.flow_layers[0].pred_net.layer
        lin.weight.data[D:, :] = 0.2 * torch.randn(D, lin.weight.shape[1])
        lin.bias.data = 0.1 * torch.randn(2 * D)
        enc.embed_layer.weight.data = 1.5 * torch.randn(C, 8)           # spread the classes
        enc.eval()
        cat = torch.randint(0, C, (B, N))
        ln = lengths(B, N, torch.Generator().manual_seed(i))
        pad = create_channel_mask(ln, max_len=N)
        kw = dict(channel_padding_mask=pad) if padded else {}
        torch.manual_seed(1200 + i)
        u = torch.rand(B * N, 1, D)
        torch.manual_seed(1200 + i)
        z, ldj, _ = enc(cat, reverse=False, beta=beta, **kw)
        wz, wl = torch.randn(B, N, D), torch.randn(B)
        ((z * wz).sum() + (ldj * wl).sum()).backward()
        with torch.no_grad():
            dec, _, _ = enc(z, reverse=True)
            z_probe = z + 0.7 * torch.randn(z.shape)
            dec_probe, _, _ = enc(z_probe, reverse=True)
            table = enc.flow_layers[0].pred_net(enc.embed_layer.weight)
        c = dict(meta=dict(B=B, N=N, D=D, C=C, beta=beta, prior=prior, padded=padded, training=False),
                 categ=cat, u=u, table=table, category_prior=enc.category_prior, pad=pad, z=z.detach(), ldj=ldj.detach(),
                 decoded=dec, z_probe=z_probe.detach(), decoded_probe=dec_probe, wz=wz, wl=wl)
        for k, v in enc.state_dict().items():
            c["sd_" + k] = v
        for k, v in enc.named_parameters():
            c["gp_" + k] = v.grad
        cases.append(c)
    save("encoder_large_vocab", cases)


def gen_sigmoid_dequant():
    cases = []
    g = torch.Generator().manual_seed(48)
    for rev_layer in (False, True):
        layer = SigmoidFlow(reverse=rev_layer)
        z = 2 * torch.randn(4, 6, 1, generator=g)
        u = torch.rand(4, 6, 1, generator=g)
        a, la = layer(z if not rev_layer else u, reverse=False)
        b, lb = layer(u if not rev_layer else z, reverse=True)
        cases.append(dict(meta=dict(reverse_layer=rev_layer), z=z, u=u, out_fwd=a, ldj_fwd=la, out_rev=b, ldj_rev=lb))
    save("sigmoid", cases)

    # variational dequantisation round trip (variational_dequantization.py:101-144 demo shape)
    torch.manual_seed(42)
    B, N, C, hidden, emb = 3, 6, 4, 16, 8

    class Net(nn.Module):
        def __init__(self, c_out):
            super().__init__()
            self.inp = nn.Linear(1, hidden)
            self.main = nn.Sequential(nn.Linear(hidden + emb, hidden), nn.ReLU(), nn.Linear(hidden, c_out))

        def forward(self, x, ext_input, **kw):
            return self.main(torch.cat([self.inp(x), ext_input], dim=-1))

    with contextlib.redirect_stdout(io.StringIO()):
        vd = VariationalDequantization(vocab_size=C, flow_config={"num_flows": 2, "model_func": lambda c_out: Net(c_out),
                                                                  "block_type": "Linear"},
                                       default_embed_layer_dims=emb)
    for p in vd.parameters():
        p.data = p.data + 0.05 * torch.randn(p.shape)
    cat = torch.randint(high=C, size=(B, N))
    torch.manual_seed(77)
    u = torch.rand(B, N)
    torch.manual_seed(77)
    z, ldj = vd(cat, reverse=False)
    rec, _ = vd(z, reverse=True)
    c = dict(meta=dict(B=B, N=N, C=C, hidden=hidden, emb=emb, num_flows=2), categ=cat, u=u, z=z, ldj=ldj, decoded=rec)
    for k, v in vd.state_dict().items():
        c["sd_" + k] = v
    save("dequant", [c])


def gen_flow_stack():
    """configs[0]-style plumbing: encoder + 4 x (ActNorm, InvConv, affine Coupling) through FlowModel,
    C in {2,16}, D=2, batch 256, |S|=16.  Subnet = small MLP (weights stored)."""
    cases = []
    for C in (2, 16):
        torch.manual_seed(11 + C)
        np.random.seed(11 + C)
        B, N, D, hidden = 256, 16, 2, 32
        mk = lambda c_out: nn.Sequential(nn.Linear(D, hidden), nn.GELU(), nn.Linear(hidden, c_out))
        with contextlib.redirect_stdout(io.StringIO()):
            layers = [LinearCategoricalEncoding(num_dimensions=D, flow_config={"num_flows": 0}, vocab_size=C)]
            for _ in range(4):
                layers += [ActNormFlow(D), InvertibleConv(D),
                           CouplingLayer(D, CouplingLayer.create_channel_mask(D), mk)]
            model = FlowModel(layers)
        for p in model.parameters():
            p.data = p.data + 0.05 * torch.randn(p.shape)
        model.eval()
        cat = torch.randint(0, C, (B, N))
        ln = torch.full((B,), N, dtype=torch.long)
        torch.manual_seed(900 + C)
        u = torch.rand(B * N, 1, D)
        torch.manual_seed(900 + C)
        with torch.no_grad():
            z, ldj = model(cat, reverse=False, length=ln)
            prior = LogisticDistribution()
            neglog = -prior.log_prob(z).sum(dim=[1, 2])
            nll = (-ldj + neglog) / ln.float()
            dec, _ = model(z, reverse=True, length=ln)
        c = dict(meta=dict(B=B, N=N, D=D, C=C, hidden=hidden, flows=4, infos=[l.info() for l in model.flow_layers]),
                 categ=cat, u=u, z=z, ldj=ldj, nll=nll,
                 decoded=dec, bpd=np.log2(np.exp(1)) * nll.mean())
        for k, v in model.state_dict().items():
            c["sd_" + k] = v
        cases.append(c)
    save("flow_stack", cases)


def gen_linear_flow_encoder():
    """Linear-flow encoding (num_flows > 0): ExtActNorm + InvertibleConv + affine coupling with a LinearNet
    sub-network per flow, posterior over all classes by inverting every class flow (linear_encoding.py)."""
    cases = []
    for i, (B, N, D, C, flows, padded, training) in enumerate([(3, 5, 4, 5, 2, False, False), (2, 6, 2, 3, 1, True, True),
                                                               (4, 4, 3, 7, 2, False, False)]):
        torch.manual_seed(300 + i)
        np.random.seed(300 + i)
        with contextlib.redirect_stdout(io.StringIO()):
            enc = LinearCategoricalEncoding(num_dimensions=D, flow_config={"num_flows": flows, "hidden_layers": 1, "hidden_size": 16},
                                            vocab_size=C, default_embed_layer_dims=8)
        for p in enc.parameters():
            p.data = p.data + 0.05 * torch.randn(p.shape)
        enc.train(training)
        cat = torch.randint(0, C, (B, N))
        ln = lengths(B, N, torch.Generator().manual_seed(i))
        pad = create_channel_mask(ln, max_len=N)
        kw = dict(channel_padding_mask=pad) if padded else {}
        torch.manual_seed(700 + i)
        u = torch.rand(B * N, 1, D)
        torch.manual_seed(700 + i)
        with torch.no_grad():
            z, ldj, det = enc(cat, reverse=False, beta=1, **kw)
            dec, _, _ = enc(z, reverse=True)
        c = dict(meta=dict(B=B, N=N, D=D, C=C, flows=flows, padded=padded, training=training, embed=8, hidden=16,
                           infos=enc.info()),
                 categ=cat, u=u, pad=pad, z=z, ldj=ldj, decoded=dec)
        for k, v in enc.state_dict().items():
            c["sd_" + k] = v
        cases.append(c)
    save("encoder_linear_flows", cases)


def gen_node_edge():
    """NodeEdgeCoupling / NodeEdgeFlowWrapper (experiments/molecule_generation/graph_node_edge_coupling.py): the
    second caller of the mixture statics, nodes [B,Nn,6] K=16 and edges [B,E,2] K=8, regulariser 3.5 x 2."""
    with contextlib.redirect_stdout(io.StringIO()):
        from experiments.molecule_generation.graph_node_edge_coupling import NodeEdgeCoupling, NodeEdgeFlowWrapper

    class Stub(nn.Module):
        def __init__(self):
            super().__init__()
            self.out = None

        def forward(self, **kwargs):
            return self.out

    cases = []
    g = torch.Generator().manual_seed(49)
    for (B, Nn, training) in [(3, 7, True), (2, 9, False)]:
        E = Nn * (Nn - 1) // 2
        Dn, De, Kn, Ke = 6, 2, 16, 8
        mn, me = CouplingLayer.create_channel_mask(Dn), CouplingLayer.create_channel_mask(De)
        with contextlib.redirect_stdout(io.StringIO()):
            layer = NodeEdgeCoupling(c_in_nodes=Dn, c_in_edges=De, mask_nodes=mn, mask_edges=me, num_mixtures_nodes=Kn,
                                     num_mixtures_edges=Ke, model_func=lambda c_out_nodes, c_out_edges: Stub(),
                                     regularizer_max=3.5, regularizer_factor=2)
        layer.scaling_factor_nodes.data = 0.3 * torch.randn(Dn, generator=g)
        layer.scaling_factor_edges.data = 0.3 * torch.randn(De, generator=g)
        layer.mixture_scaling_factor_nodes.data = 0.3 * torch.randn(Dn, Kn, generator=g)
        layer.mixture_scaling_factor_edges.data = 0.3 * torch.randn(De, Ke, generator=g)
        layer.train(training)
        zn, ze = torch.randn(B, Nn, Dn, generator=g), torch.randn(B, E, De, generator=g)
        nn_n = 0.6 * torch.randn(B, Nn, Dn * (2 + 3 * Kn), generator=g)
        nn_e = 0.6 * torch.randn(B, E, De * (2 + 3 * Ke), generator=g)
        layer.nn.out = (nn_n, nn_e)
        ln = lengths(B, Nn, g)
        pad = create_channel_mask(ln, max_len=Nn)
        valid = (torch.rand(B, E, generator=g) > 0.3).float()
        kw = dict(length=ln, channel_padding_mask=pad, mask_valid=valid)
        zn_f, ze_f, ldj_f, det = layer(zn, ze, reverse=False, **kw)
        zn_r, ze_r, ldj_r, _ = layer(zn_f, ze_f, reverse=True, **kw)
        wrap = NodeEdgeFlowWrapper(ActNormFlow(Dn, data_init=False), ActNormFlow(De, data_init=False))
        wrap.node_flow.bias.data, wrap.node_flow.scales.data = torch.randn(1, 1, Dn, generator=g), 0.3 * torch.randn(1, 1, Dn, generator=g)
        wrap.edge_flow.bias.data, wrap.edge_flow.scales.data = torch.randn(1, 1, De, generator=g), 0.3 * torch.randn(1, 1, De, generator=g)
        ldj0 = torch.randn(B, generator=g)
        wn, we, wl = wrap(zn, ze, ldj=ldj0.clone(), reverse=False, **kw)
        c = dict(meta=dict(B=B, Nn=Nn, E=E, Dn=Dn, De=De, Kn=Kn, Ke=Ke, training=training, info=layer.info()),
                 z_nodes=zn, z_edges=ze, nn_nodes=nn_n, nn_edges=nn_e, length=ln, pad=pad, mask_valid=valid,
                 z_nodes_fwd=zn_f, z_edges_fwd=ze_f, ldj_fwd=ldj_f, z_nodes_rev=zn_r, z_edges_rev=ze_r, ldj_rev=ldj_r,
                 reg_nodes=det["regularizer_nodes_ldj"], reg_edges=det["regularizer_edges_ldj"],
                 wrap_ldj_in=ldj0, wrap_nodes=wn, wrap_edges=we, wrap_ldj=wl)
        for k, v in layer.state_dict().items():
            c["sd_" + k] = v
        for k, v in wrap.state_dict().items():
            c["wsd_" + k] = v
        cases.append(c)
    save("node_edge_coupling", cases)


def gen_data_init():
    """FlowModel.initialize_data_dependent (flow_model.py:96-131) through ActNorm -> InvConv -> ActNorm."""
    torch.manual_seed(51)
    np.random.seed(51)
    D, N = 4, 6
    with contextlib.redirect_stdout(io.StringIO()):
        model = FlowModel([ActNormFlow(D), InvertibleConv(D), ActNormFlow(D)])
    sd0 = {k: v.clone() for k, v in model.state_dict().items()}
    batches = []
    g = torch.Generator().manual_seed(52)
    for b in range(3):
        B = 5 + b
        ln = lengths(B, N, g)
        batches.append((2.0 * torch.randn(B, N, D, generator=g) + 0.5, {"length": ln, "channel_padding_mask": create_channel_mask(ln, max_len=N)}))
    with contextlib.redirect_stdout(io.StringIO()):
        model.initialize_data_dependent([(z.clone(), dict(kw)) for z, kw in batches])
    c = dict(meta=dict(D=D, N=N, batches=[int(z.size(0)) for z, _ in batches]))
    for i, (z, kw) in enumerate(batches):
        c["z%d" % i], c["length%d" % i], c["pad%d" % i] = z, kw["length"], kw["channel_padding_mask"]
    for k, v in sd0.items():
        c["sd0_" + k] = v
    for k, v in model.state_dict().items():
        c["sd_" + k] = v
    save("data_init", [c])


def gen_grads():
    """Gradients of every differentiable hot-path layer, taken with the reference's autograd:
    loss = sum(z_out * wz) + sum(ldj * wl) with fixed random weights."""
    g = torch.Generator().manual_seed(60)
    cases = []

    def leaf(t):
        return t.clone().requires_grad_(True)

    def leaf64(t):
        return t.detach().double().clone().requires_grad_(True)

    # Every case also carries the SAME gradients from a float64 run of the reference (`g64_*`: the module converted
    # with .double(), double inputs): |g - g64| is the rounding noise of the reference's own fp32 autograd, the unit in
    # which the GPU tests state their gradient tolerance (tests/test_gpu_parity.py, grad_close).  The float64 runs
    # consume no random numbers, so the float32 arrays are the ones round 1 wrote.

    # affine coupling, both directions, channel and chess masks
    for (B, N, D, kind, reverse) in [(4, 6, 4, "channel", False), (3, 5, 6, "channel", True), (4, 7, 1, "chess", False),
                                     (5, 3, 3, "channel", True), (2, 64, 6, "channel", False)]:
        mask = CouplingLayer.create_channel_mask(D) if kind == "channel" else CouplingLayer.create_chess_mask()
        with contextlib.redirect_stdout(io.StringIO()):
            layer = CouplingLayer(c_in=D, mask=mask, model_func=lambda c_out: Inject())
        layer.scaling_factor.data = torch.cat([torch.zeros(1), 0.5 * torch.randn(D - 1, generator=g)]) if D > 1 else torch.zeros(1)
        z, nn_out, ldj = leaf(torch.randn(B, N, D, generator=g)), leaf(1.2 * torch.randn(B, N, 2 * D, generator=g)), leaf(torch.randn(B, generator=g))
        layer.nn.value = nn_out
        wz, wl = torch.randn(B, N, D, generator=g), torch.randn(B, generator=g)
        zo, lo = layer(z, ldj=ldj, reverse=reverse)
        ((zo * wz).sum() + (lo * wl).sum()).backward()
        l64 = copy.deepcopy(layer).double()
        z64, nn64, ldj64 = leaf64(z), leaf64(nn_out), leaf64(ldj)
        l64.nn.value = nn64
        zo64, lo64 = l64(z64, ldj=ldj64, reverse=reverse)
        ((zo64 * wz.double()).sum() + (lo64 * wl.double()).sum()).backward()
        cases.append(dict(meta=dict(layer="affine", B=B, N=N, D=D, mask_kind=kind, reverse=reverse), z=z.detach(), nn_out=nn_out.detach(),
                          ldj=ldj.detach(), mask=mask, scaling_factor=layer.scaling_factor.data, wz=wz, wl=wl,
                          g_z=z.grad, g_nn=nn_out.grad, g_ldj=ldj.grad, g_sf=layer.scaling_factor.grad,
                          g64_z=z64.grad, g64_nn=nn64.grad, g64_ldj=ldj64.grad, g64_sf=l64.scaling_factor.grad))

    # ActNorm
    for (B, N, D, mode, reverse) in [(4, 6, 3, "length_mask", False), (3, 5, 4, "none", True), (5, 4, 6, "mask", False), (3, 7, 2, "length", True)]:
        layer = ActNormFlow(D)
        layer.bias.data, layer.scales.data = torch.randn(1, 1, D, generator=g), 0.4 * torch.randn(1, 1, D, generator=g)
        z, ldj = leaf(torch.randn(B, N, D, generator=g)), leaf(torch.randn(B, generator=g))
        ln = lengths(B, N, g)
        pad = create_channel_mask(ln, max_len=N)
        kw = {}
        if "length" in mode:
            kw["length"] = ln
        if "mask" in mode:
            kw["channel_padding_mask"] = pad
        wz, wl = torch.randn(B, N, D, generator=g), torch.randn(B, generator=g)
        zo, lo = layer(z, ldj=ldj * 1.0, reverse=reverse, **kw)
        ((zo * wz).sum() + (lo * wl).sum()).backward()
        l64 = copy.deepcopy(layer).double()
        z64, ldj64 = leaf64(z), leaf64(ldj)
        zo64, lo64 = l64(z64, ldj=ldj64 * 1.0, reverse=reverse, **{k: (v.double() if v.is_floating_point() else v) for k, v in kw.items()})
        ((zo64 * wz.double()).sum() + (lo64 * wl.double()).sum()).backward()
        cases.append(dict(meta=dict(layer="actnorm", B=B, N=N, D=D, mode=mode, reverse=reverse), z=z.detach(), ldj=ldj.detach(),
                          bias=layer.bias.data, scales=layer.scales.data, length=ln, pad=pad, wz=wz, wl=wl,
                          g_z=z.grad, g_ldj=ldj.grad, g_bias=layer.bias.grad, g_scales=layer.scales.grad,
                          g64_z=z64.grad, g64_ldj=ldj64.grad, g64_bias=l64.bias.grad, g64_scales=l64.scales.grad))

    # ExtActNorm
    for (B, N, D, padded, reverse) in [(6, 1, 3, False, False), (4, 5, 4, True, False), (5, 1, 2, False, True)]:
        net = Inject()
        layer = ExtActNormFlow(D, net=net)
        z, nn_out, ldj = leaf(torch.randn(B, N, D, generator=g)), leaf(torch.randn(B, N, 2 * D, generator=g)), leaf(torch.randn(B, generator=g))
        net.value = nn_out
        ln = lengths(B, N, g)
        pad = create_channel_mask(ln, max_len=N)
        wz, wl = torch.randn(B, N, D, generator=g), torch.randn(B, generator=g)
        zo, lo = layer(z, ldj * 1.0, ext_input=z, reverse=reverse, **(dict(channel_padding_mask=pad) if padded else {}))
        ((zo * wz).sum() + (lo * wl).sum()).backward()
        net64 = Inject()
        l64 = ExtActNormFlow(D, net=net64).double()
        z64, nn64, ldj64 = leaf64(z), leaf64(nn_out), leaf64(ldj)
        net64.value = nn64
        zo64, lo64 = l64(z64, ldj64 * 1.0, ext_input=z64, reverse=reverse, **(dict(channel_padding_mask=pad.double()) if padded else {}))
        ((zo64 * wz.double()).sum() + (lo64 * wl.double()).sum()).backward()
        cases.append(dict(meta=dict(layer="ext_actnorm", B=B, N=N, D=D, padded=padded, reverse=reverse), z=z.detach(), nn_out=nn_out.detach(),
                          ldj=ldj.detach(), pad=pad, wz=wz, wl=wl, g_z=z.grad, g_nn=nn_out.grad, g_ldj=ldj.grad,
                          g64_z=z64.grad, g64_nn=nn64.grad, g64_ldj=ldj64.grad))

    # invertible 1x1 convolution (LU parameters and dense weight)
    np.random.seed(61)
    for (B, N, D, lu, mode, reverse) in [(4, 6, 4, True, "length_mask", False), (3, 5, 3, False, "none", False), (4, 3, 6, True, "length", True),
                                         (3, 4, 2, True, "none", False)]:
        layer = InvertibleConv(D, LU_decomposed=lu)
        layer.train()
        x, ldj = leaf(torch.randn(B, N, D, generator=g)), leaf(torch.randn(B, generator=g))
        ln = lengths(B, N, g)
        pad = create_channel_mask(ln, max_len=N)
        kw = {}
        if "length" in mode:
            kw["length"] = ln
        if "mask" in mode:
            kw["channel_padding_mask"] = pad
        wz, wl = torch.randn(B, N, D, generator=g), torch.randn(B, generator=g)
        zo, lo = layer(x, ldj=ldj, reverse=reverse, **kw)
        ((zo * wz).sum() + (lo * wl).sum()).backward()
        c = dict(meta=dict(layer="invconv", B=B, N=N, D=D, lu=lu, mode=mode, reverse=reverse), x=x.detach(), ldj=ldj.detach(),
                 length=ln, pad=pad, wz=wz, wl=wl, g_x=x.grad, g_ldj=ldj.grad)
        for k, v in layer.state_dict().items():
            c["sd_" + k] = v
        for k, v in layer.named_parameters():
            c["gp_" + k] = v.grad
        if not reverse:      # the reference's inverse casts its weight to float (permutation_layers.py:86): no float64 run of it exists
            l64 = copy.deepcopy(layer).double()
            l64.zero_grad()
            x64, ldj64 = leaf64(x), leaf64(ldj)
            zo64, lo64 = l64(x64, ldj=ldj64, reverse=reverse, **{k: (v.double() if v.is_floating_point() else v) for k, v in kw.items()})
            ((zo64 * wz.double()).sum() + (lo64 * wl.double()).sum()).backward()
            c["g64_x"], c["g64_ldj"] = x64.grad, ldj64.grad
            for k, v in l64.named_parameters():
                c["gp64_" + k] = v.grad
        cases.append(c)

    # logistic prior log-prob, NLL assembly, sigmoid flows
    prior = LogisticDistribution()
    x = leaf(2 * torch.randn(5, 6, 3, generator=g))
    w = torch.randn(5, 6, 3, generator=g)
    (prior.log_prob(x) * w).sum().backward()
    x64 = leaf64(x)
    (prior.log_prob(x64) * w.double()).sum().backward()
    cases.append(dict(meta=dict(layer="log_prob"), x=x.detach(), w=w, g_x=x.grad, g64_x=x64.grad))
    B, N, D = 5, 6, 4
    z, ldj = leaf(1.5 * torch.randn(B, N, D, generator=g)), leaf(torch.randn(B, generator=g))
    ln = lengths(B, N, g)
    pad = create_channel_mask(ln, max_len=N)
    wl = torch.randn(B, generator=g)
    nll = (-ldj) / ln.float() + (-(prior.log_prob(z) * pad).sum(dim=[1, 2])) / ln.float()
    (nll * wl).sum().backward()
    z64, ldj64 = leaf64(z), leaf64(ldj)
    nll64 = (-ldj64) / ln.double() + (-(prior.log_prob(z64) * pad.double()).sum(dim=[1, 2])) / ln.double()
    (nll64 * wl.double()).sum().backward()
    cases.append(dict(meta=dict(layer="nll", B=B, N=N, D=D), z=z.detach(), ldj=ldj.detach(), length=ln, pad=pad, wl=wl,
                      g_z=z.grad, g_ldj=ldj.grad, g64_z=z64.grad, g64_ldj=ldj64.grad))
    for reverse in (False, True):
        layer = SigmoidFlow()
        zi = leaf(2 * torch.randn(4, 6, 1, generator=g)) if not reverse else leaf(torch.rand(4, 6, 1, generator=g))
        ldj = leaf(torch.randn(4, generator=g))
        wz, wl = torch.randn(4, 6, 1, generator=g), torch.randn(4, generator=g)
        zo, lo = layer(zi, ldj=ldj, reverse=reverse)
        ((zo * wz).sum() + (lo * wl).sum()).backward()
        z64, ldj64 = leaf64(zi), leaf64(ldj)
        zo64, lo64 = SigmoidFlow()(z64, ldj=ldj64, reverse=reverse)
        ((zo64 * wz.double()).sum() + (lo64 * wl.double()).sum()).backward()
        cases.append(dict(meta=dict(layer="sigmoid", reverse=reverse), z=zi.detach(), ldj=ldj.detach(), wz=wz, wl=wl, g_z=zi.grad, g_ldj=ldj.grad,
                          g64_z=z64.grad, g64_ldj=ldj64.grad))

    # mixture-CDF coupling forward (the inverse is never differentiated by the reference)
    for (B, N, D, K, kind, reg_max, training, padded) in [(3, 6, 4, 8, "channel", -1, True, False), (3, 5, 6, 16, "channel", 3.5, True, True),
                                                          (2, 7, 3, 4, "none", -1, True, False), (4, 6, 1, 8, "chess", -1, True, True),
                                                          (2, 4, 2, 10, "channel", -1, False, False)]:
        P = 2 + 3 * K
        z, nn_out = leaf(1.5 * torch.randn(B, N, D, generator=g)), leaf(0.8 * torch.randn(B, N, D * P, generator=g))
        sf, msf = 0.4 * torch.randn(D, generator=g), 0.4 * torch.randn(D, K, generator=g)
        sf[0] = 0.0
        ln = lengths(B, N, g)
        pad = create_channel_mask(ln, max_len=N)
        wz, wl = torch.randn(B, N, D, generator=g), torch.randn(B, generator=g)
        if kind == "none":
            with contextlib.redirect_stdout(io.StringIO()):
                layer = AutoregressiveMixtureCDFCoupling(c_in=D, model_func=lambda c_out: Inject(), num_mixtures=K)
            mask = None
        else:
            mask = CouplingLayer.create_channel_mask(D) if kind == "channel" else CouplingLayer.create_chess_mask()
            with contextlib.redirect_stdout(io.StringIO()):
                layer = MixtureCDFCoupling(c_in=D, mask=mask, model_func=lambda c_out: Inject(), num_mixtures=K,
                                           regularizer_max=reg_max, regularizer_factor=2)
        layer.scaling_factor.data, layer.mixture_scaling_factor.data = sf.clone(), msf.clone()
        layer.nn.value = nn_out
        layer.train(training)
        res = layer(z, reverse=False, **(dict(channel_padding_mask=pad) if padded else {}))
        zo, lo = res[0], res[1]
        ((zo * wz).sum() + (lo * wl).sum()).backward()
        c = dict(meta=dict(layer="mixture", B=B, N=N, D=D, K=K, mask_kind=kind, reg_max=reg_max, reg_factor=2, training=training, padded=padded),
                 z=z.detach(), nn_out=nn_out.detach(), scaling_factor=sf, mixture_scaling_factor=msf, pad=pad, wz=wz, wl=wl,
                 g_z=z.grad, g_nn=nn_out.grad, g_sf=layer.scaling_factor.grad, g_msf=layer.mixture_scaling_factor.grad)
        if mask is not None:
            c["mask"] = mask
        l64 = copy.deepcopy(layer).double()
        l64.zero_grad()
        z64, nn64 = leaf64(z), leaf64(nn_out)
        l64.nn.value = nn64
        res64 = l64(z64, reverse=False, **(dict(channel_padding_mask=pad.double()) if padded else {}))
        ((res64[0] * wz.double()).sum() + (res64[1] * wl.double()).sum()).backward()
        c.update(g64_z=z64.grad, g64_nn=nn64.grad, g64_sf=l64.scaling_factor.grad, g64_msf=l64.mixture_scaling_factor.grad)
        cases.append(c)

    # mixture-model encoder: gradients reach the embedding and the predictor through the class table
    for i, (B, N, D, C, beta, padded) in enumerate([(3, 6, 3, 4, 1.0, False), (4, 5, 4, 9, 1.5, True)]):
        torch.manual_seed(800 + i)
        with contextlib.redirect_stdout(io.StringIO()):
            enc = LinearCategoricalEncoding(num_dimensions=D, flow_config={"num_flows": 0}, vocab_size=C, default_embed_layer_dims=8)
        lin = enc.flow_layers[0].pred_net.layer
        lin.weight.data[D:, :] = 0.2 * torch.randn(D, lin.weight.shape[1])
        lin.bias.data = 0.1 * torch.randn(2 * D)
        enc.eval()
        cat = torch.randint(0, C, (B, N))
        ln = lengths(B, N, torch.Generator().manual_seed(i))
        pad = create_channel_mask(ln, max_len=N)
        torch.manual_seed(900 + i)
        u = torch.rand(B * N, 1, D)
        torch.manual_seed(900 + i)
        zo, lo, _ = enc(cat, reverse=False, beta=beta, **(dict(channel_padding_mask=pad) if padded else {}))
        wz, wl = torch.randn(B, N, D), torch.randn(B)
        ((zo * wz).sum() + (lo * wl).sum()).backward()
        c = dict(meta=dict(layer="encoder", B=B, N=N, D=D, C=C, beta=beta, padded=padded), categ=cat, u=u, pad=pad, wz=wz, wl=wl)
        for k, v in enc.state_dict().items():
            c["sd_" + k] = v
        for k, v in enc.named_parameters():
            c["gp_" + k] = v.grad
        e64 = copy.deepcopy(enc).double()
        e64.zero_grad()
        torch.manual_seed(900 + i)
        zo64, lo64, _ = e64(cat, reverse=False, beta=beta, **(dict(channel_padding_mask=pad.double()) if padded else {}))
        ((zo64 * wz.double()).sum() + (lo64 * wl.double()).sum()).backward()
        for k, v in e64.named_parameters():
            c["gp64_" + k] = v.grad
        cases.append(c)
    save("grads", cases)


def one_graph_case(GraphNodeFlow, Colours, seed, B, N, D, K, lo, hi):
    torch.manual_seed(seed)
    np.random.seed(seed)
    params = {"coupling_num_flows": 2, "coupling_hidden_size": 32, "coupling_hidden_layers": 2, "coupling_num_mixtures": K,
              "coupling_mask_ratio": 0.5, "coupling_dropout": 0.0,
              "categ_encoding": {"use_dequantization": False, "use_variational": False, "use_decoder": False, "num_dimensions": D,
                                 "flow_config": {"num_flows": 0, "hidden_layers": 2, "hidden_size": 128},
                                 "decoder_config": {"num_layers": 1, "hidden_size": 64}}}
    with contextlib.redirect_stdout(io.StringIO()):
        model = GraphNodeFlow(params, Colours)
    for p in model.parameters():
        p.data = p.data + 0.05 * torch.randn(p.shape)
    model.eval()
    g = torch.Generator().manual_seed(seed + 1)
    ln = torch.randint(lo, hi + 1, (B,), generator=g)
    ln[0] = N
    adj = (torch.rand(B, N, N, generator=g) < 0.3).long()
    adj = torch.triu(adj, 1)
    adj = adj + adj.transpose(1, 2)
    valid = (torch.arange(N).view(1, N) < ln.view(B, 1))
    adj = adj * (valid.unsqueeze(1) & valid.unsqueeze(2)).long()
    cat = torch.randint(0, 3, (B, N), generator=g) * valid.long()
    torch.manual_seed(seed + 2)
    u = torch.rand(B * N, 1, D)
    torch.manual_seed(seed + 2)
    with torch.no_grad():
        z, ldj = model(cat, adjacency=adj, reverse=False, length=ln)
        dec, _ = model(z, adjacency=adj, reverse=True, length=ln)
        with contextlib.redirect_stdout(io.StringIO()):
            rev_ok = bool(model.test_reversibility(cat, adj, ln))
            perm_ok = bool(model.test_permutation(cat, adj, ln))
        # the RGCN-attention sub-network of the first coupling in isolation (a plain PyTorch module on both sides)
        pad = create_channel_mask(ln, max_len=N)
        sub_in = torch.randn(B, N, D, generator=g) * pad
        sub_out = model.flow_layers[3].nn(sub_in, adjacency=adj, channel_padding_mask=pad)
    c = dict(meta=dict(B=B, N=N, D=D, K=K, hidden=32, layers=2, flows=2, rev_ok=rev_ok, perm_ok=perm_ok,
                       infos=[l.info() for l in model.flow_layers]),
             categ=cat, adjacency=adj, length=ln, u=u, z=z, ldj=ldj, decoded=dec, sub_in=sub_in, sub_out=sub_out)
    for k, v in model.state_dict().items():
        c["sd_" + k] = v
    return c


def gen_graph_node_flow():
    """configs[2]: node-based GraphCNF for graph colouring (experiments/graph_coloring/graph_node_flow.py) with the
    RGCN-attention sub-network, 3 colours, synthetic sparse graphs of 6..10 nodes, reg_max 3.5 x 2, eval mode."""
    with contextlib.redirect_stdout(io.StringIO()):
        from experiments.graph_coloring.graph_node_flow import GraphNodeFlow

    class Colours:
        @staticmethod
        def num_node_types():
            return 3

    cases = []
    # case 0: 6..10-node graphs (round 1); cases 1, 2: the sizes of the reference's tiny_3 (10..20 nodes, D = 2, K = 8)
    # and large_3 (25..50 nodes, D = 6, K = 16) data sets (experiments/graph_coloring/README.md:19-43)
    for seed, B, N, D, K, lo, hi in [(70, 8, 10, 2, 8, 6, 10), (170, 6, 20, 2, 8, 10, 20), (270, 3, 50, 6, 16, 25, 50)]:
        cases.append(one_graph_case(GraphNodeFlow, Colours, seed, B, N, D, K, lo, hi))
    save("graph_node_flow", cases)


def gen_language_model():
    """configs[3]: language-modelling flow (experiments/language_modeling/flow_model.py) — linear-flow encoder + ActNorm /
    1x1 conv / autoregressive mixture-CDF couplings with the LSTM sub-network, variable lengths, eval mode.
    The reference's general.mutils.create_T_one_hot clamps a long tensor with a float bound; torch >= 2 promotes the
    result to float and scatter_ then rejects it as an index.  The generator casts the index back to long inside
    general.mutils.one_hot (in memory, for this run only) — the value is what torch 1.x computed."""
    import general.mutils as gm
    orig_one_hot = gm.one_hot
    gm.one_hot = lambda x, num_classes, dtype=torch.float32: orig_one_hot(x.long() if isinstance(x, torch.Tensor) else x,
                                                                            num_classes, dtype)
    with contextlib.redirect_stdout(io.StringIO()):
        from experiments.language_modeling.flow_model import FlowLanguageModeling

    class Vocab:
        vectors = None

    cases = []
    # cases 0, 1: T = 24 (round 1); case 2: the Penn Treebank configuration's sizes — sequence length 288, 51 symbols,
    # D = 3, K = 51, one flow (experiments/language_modeling/README.md:8-21), LSTM hidden size reduced to 48
    for ci, (K, D, flows, enc_flows, hidden, V, T, B) in enumerate([(5, 3, 2, 1, 32, 20, 24, 6), (51, 3, 1, 0, 32, 20, 24, 6),
                                                                      (51, 3, 1, 0, 48, 51, 288, 3)]):
        torch.manual_seed(80 + ci)
        np.random.seed(80 + ci)
        params = {"max_seq_len": T, "coupling_hidden_layers": 1, "coupling_hidden_size": hidden, "coupling_num_flows": flows,
                  "coupling_num_mixtures": K, "coupling_dropout": 0.0, "coupling_input_dropout": 0.0,
                  "categ_encoding": {"use_dequantization": False, "use_variational": False, "use_decoder": False,
                                     "num_dimensions": D, "flow_config": {"num_flows": enc_flows, "hidden_layers": 1, "hidden_size": 32},
                                     "decoder_config": {"num_layers": 1, "hidden_size": 64}}}
        with contextlib.redirect_stdout(io.StringIO()):
            model = FlowLanguageModeling(params, None, vocab_size=V, vocab=Vocab())
        for p_ in model.parameters():
            p_.data = p_.data + 0.05 * torch.randn(p_.shape)
        model.eval()
        g = torch.Generator().manual_seed(81 + ci)
        ln = torch.randint(5 if T < 100 else 40, T + 1, (B,), generator=g)
        ln[0] = T
        valid = (torch.arange(T).view(1, T) < ln.view(B, 1))
        x = torch.randint(0, V, (B, T), generator=g) * valid.long()
        torch.manual_seed(82 + ci)
        u = torch.rand(B * T, 1, D)
        torch.manual_seed(82 + ci)
        with torch.no_grad():
            z, ldj = model(x, reverse=False, length=ln)
            pad = create_channel_mask(ln, max_len=T)
            sub_in = torch.randn(B, T, D, generator=g) * pad
            sub = [l for l in model.flow_layers if l.__class__.__name__ == "AutoregressiveMixtureCDFCoupling"][0].nn
            sub_out = sub(x=sub_in, length=ln, channel_padding_mask=pad)
        c = dict(meta=dict(B=B, T=T, V=V, D=D, K=K, flows=flows, enc_flows=enc_flows, hidden=hidden,
                           infos=[l.info() for l in model.flow_layers]),
                 tokens=x, length=ln, u=u, z=z, ldj=ldj, sub_in=sub_in, sub_out=sub_out)
        for k, v in model.state_dict().items():
            c["sd_" + k] = v
        cases.append(c)
    gm.one_hot = orig_one_hot
    save("language_model", cases)


def gen_graph_cnf():
    """Two cases: the reduced assembly of round 2 (9 nodes, 36 pairs, D = 4 / 2, K = 8 / 4, 5 node types, flows 1,2,2) and
    configs[4] AT ITS REAL SIZES (experiments/molecule_generation/README.md:19-30, train.py:70-76, zinc250k.py:150-164):
    38 nodes, 703 node pairs, D = 6 / 2, K = 16 / 8, 9 node types, 3 edge types, flows 4,6,6, graphs of 11..38 atoms."""
    _graph_cnf_case("graph_cnf", NT=5, ET=3, NMAX=9, NEIGH=4, DN=4, DE=2, KN=8, KE=4, flows="1,2,2", B=5, min_len=4, seed=90,
                    node_prior=[0.4, 0.3, 0.15, 0.1, 0.05], edge_prior=[0.7, 0.2, 0.1])
    _graph_cnf_case("graph_cnf_zinc", NT=9, ET=3, NMAX=38, NEIGH=5, DN=6, DE=2, KN=16, KE=8, flows="4,6,6", B=2, min_len=11, seed=190,
                    node_prior=[0.55, 0.12, 0.14, 0.02, 0.06, 0.05, 0.03, 0.02, 0.01], edge_prior=[0.75, 0.2, 0.05])


def _graph_cnf_case(name, NT, ET, NMAX, NEIGH, DN, DE, KN, KE, flows, B, min_len, seed, node_prior, edge_prior):
    """configs[4]: the three-stage molecule GraphCNF (experiments/molecule_generation/graphCNF.py) assembled by the
    REFERENCE from its own layers, with every coupling sub-network (RGCN in stage 1, Edge-GNN in stages 2 / 3 — the
    latter cannot run on torch >= 2, layers/networks/graph_layers.py:527,668) replaced by a stub that returns a
    pre-drawn tensor.  Captured: the three encoders' uniform noise, the injected sub-network outputs, node / edge latents
    after every stage, the log-det after every layer, the final outputs, and one reverse (sampling) pass from given
    latents: decoded node types and adjacency."""
    with contextlib.redirect_stdout(io.StringIO()):
        from experiments.molecule_generation.graphCNF import GraphCNF
        from experiments.molecule_generation.graph_node_edge_coupling import NodeEdgeCoupling

    class Molecules:
        @staticmethod
        def max_num_nodes():
            return NMAX

        @staticmethod
        def num_node_types():
            return NT

        @staticmethod
        def num_edge_types():
            return ET

        @staticmethod
        def num_max_neighbours():
            return NEIGH

        @staticmethod
        def get_node_prior(data_root=None):
            return np.array(node_prior, dtype=np.float32)

        @staticmethod
        def get_edge_prior(data_root=None):
            return np.array(edge_prior, dtype=np.float32)

    class InjectPair(nn.Module):
        def __init__(self):
            super().__init__()
            self.value = None

        def forward(self, **kwargs):
            return self.value

    torch.manual_seed(seed)
    np.random.seed(seed)
    enc = lambda d: {"use_dequantization": False, "use_variational": False, "use_decoder": False, "num_dimensions": d,
                     "flow_config": {"num_flows": 0, "hidden_layers": 2, "hidden_size": 128},
                     "decoder_config": {"num_layers": 1, "hidden_size": 64}}
    params = {"categ_encoding_nodes": enc(DN), "categ_encoding_edges": enc(DE), "encoding_virtual_num_flows": 0,
              "coupling_hidden_size_nodes": 16, "coupling_hidden_size_edges": 8, "coupling_num_flows": flows,
              "coupling_hidden_layers": 1, "coupling_num_mixtures_nodes": KN, "coupling_num_mixtures_edges": KE,
              "coupling_mask_ratio": 0.5, "coupling_dropout": 0.0}
    with contextlib.redirect_stdout(io.StringIO()):
        model = GraphCNF(params, Molecules)
    # the reference hands the decoder's last bias a float64 numpy prior (graphCNF.py:77); torch >= 2 no longer mixes
    # a double bias with float activations in addmm, so the bias is cast to fp32 here (in memory, this run only)
    model.edge_virtual_decoder.float()
    N = NMAX
    E = N * (N - 1) // 2
    g = torch.Generator().manual_seed(seed + 1)
    injected = []
    for flows in (model.step1_flows, model.step2_flows, model.step3_flows):
        for layer in flows:
            if isinstance(layer, NodeEdgeCoupling):
                layer.nn = InjectPair()
                layer.nn.value = (0.5 * torch.randn(B, N, layer.c_out_nodes, generator=g), 0.5 * torch.randn(B, E, layer.c_out_edges, generator=g))
                injected += list(layer.nn.value)
            elif layer.__class__.__name__ == "MixtureCDFCoupling":
                layer.nn = Inject()
                layer.nn.value = 0.5 * torch.randn(B, N, DN * (2 + 3 * KN), generator=g)
                injected.append(layer.nn.value)
    for p_ in model.parameters():
        p_.data = p_.data + 0.1 * torch.randn(p_.shape, generator=g)
    model.eval()
    ln = torch.randint(min_len, N + 1, (B,), generator=g)
    ln[0] = N
    valid = (torch.arange(N).view(1, N) < ln.view(B, 1))
    nodes = torch.randint(0, NT, (B, N), generator=g) * valid.long()
    adj = torch.randint(0, ET + 1, (B, N, N), generator=g) * (torch.rand(B, N, N, generator=g) < 0.35).long()
    adj = torch.triu(adj, 1)
    adj = (adj + adj.transpose(1, 2)) * (valid.unsqueeze(1) & valid.unsqueeze(2)).long()

    stage_out = {}
    hooks = [model.step1_flows[-1].register_forward_hook(lambda m, i, o: stage_out.__setitem__("s1", o)),
             model.step2_flows[-1].register_forward_hook(lambda m, i, o: stage_out.__setitem__("s2", o)),
             model.step3_flows[-1].register_forward_hook(lambda m, i, o: stage_out.__setitem__("s3", o))]
    torch.manual_seed(seed + 2)
    u_nodes, u_attr, u_virtual = torch.rand(B * N, 1, DN), torch.rand(B * E, 1, DE), torch.rand(B * E, 1, DE)
    torch.manual_seed(seed + 2)
    with torch.no_grad():
        z, ldj, per_layer = model(nodes, adjacency=adj, reverse=False, get_ldj_per_layer=True, length=ln)
    for h in hooks:
        h.remove()

    def layer_value(d):
        if isinstance(d, torch.Tensor):
            return d
        if "ldj" in d:
            return d["ldj"]
        if len(d) == 0:
            return torch.full((B,), float("nan"))         # an encoder in eval mode reports nothing
        return list(d.values())[0]
    layer_ldj = torch.stack([layer_value(d).float() for d in per_layer])
    # reverse (sampling) pass from fixed latents: node latents = the forward output, edge latents drawn like the reference
    torch.manual_seed(seed + 3)
    edge_latents = model.prior_distribution.sample(shape=(B, E, DE))
    torch.manual_seed(seed + 3)
    with torch.no_grad():
        (dec_nodes, dec_adj), ldj_rev = model(z, reverse=True, length=ln)
    extra = {} if name == "graph_cnf" else dict(NEIGH=NEIGH, node_prior=node_prior, edge_prior=edge_prior)     # round 2's file stays byte-identical
    c = dict(meta=dict(B=B, N=N, E=E, DN=DN, DE=DE, KN=KN, KE=KE, NT=NT, ET=ET, params=params, **extra,
                       infos=[l.info() for l in list(model.step1_flows) + list(model.step2_flows) + list(model.step3_flows)]),
             nodes=nodes, adjacency=adj, length=ln, u_nodes=u_nodes, u_attr=u_attr, u_virtual=u_virtual,
             z=z, ldj=ldj, layer_ldj=layer_ldj, s1_z=stage_out["s1"][0], s2_z_nodes=stage_out["s2"][0], s2_z_edges=stage_out["s2"][1],
             s3_z_nodes=stage_out["s3"][0], s3_z_edges=stage_out["s3"][1], edge_latents=edge_latents, dec_nodes=dec_nodes,
             dec_adjacency=dec_adj, ldj_rev=ldj_rev)
    for i, t in enumerate(injected):
        c["inj_%02d" % i] = t
    for k, v in model.state_dict().items():
        c["sd_" + k] = v
    save(name, [c])


if __name__ == "__main__":
    torch.set_num_threads(1)
    gen_affine()
    gen_mixture()
    gen_actnorm()
    gen_invconv()
    gen_prior()
    gen_encoder()
    gen_sigmoid_dequant()
    gen_flow_stack()
    gen_linear_flow_encoder()
    gen_node_edge()
    gen_data_init()
    gen_grads()
    gen_graph_node_flow()
    gen_language_model()
    gen_graph_cnf()
    gen_encoder_large_vocab()
