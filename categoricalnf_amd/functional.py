"""Differentiable entry points: torch.autograd.Function wrappers whose forward AND backward are HIP kernels.

The reference trains by differentiating its eager op chains; here each flow layer is one forward kernel and
one backward kernel (csrc/cnf_backward.hip, csrc/cnf_mixture_bwd.hip).  Without grad mode (evaluation,
sampling, benchmarks) the layer modules call `ops` directly; with it they go through these Functions."""
import ctypes

import torch

from . import _lib, ops
from .ops import _f32, _launch, _mask_desc, _opt_f32, _pad2d, _length, _ptr, _stream


_ws_cache = {}


def _ws(param_count, device):
    """Workspace of a backward entry point that reduces parameter gradients (cnf_bwd_workspace_floats): one buffer per
    (device, stream, size), reused — launches on one stream are ordered, so the next launch's kernels overwrite it only after
    this launch's reduction has read it.  Not under stream capture, where the buffer must belong to the graph's pool."""
    n = _ws_sizes.get(param_count)
    if n is None:
        n = _ws_sizes[param_count] = int(_lib.load().cnf_bwd_workspace_floats(int(param_count)))
    if torch.cuda.is_current_stream_capturing():
        return torch.empty(n, dtype=torch.float32, device=device)
    key = (device, _stream(device).value, n)
    buf = _ws_cache.get(key)
    if buf is None:
        if len(_ws_cache) > 64:
            _ws_cache.clear()
        buf = _ws_cache[key] = torch.empty(n, dtype=torch.float32, device=device)
    return buf


_ws_sizes = {}


def _g(t, like=None):
    """contiguous fp32 upstream gradient or None"""
    if t is None:
        return None
    return t.contiguous() if not t.is_contiguous() else t


class _Hold(list):
    """Raw pointers of upstream gradients for one kernel launch.  A non-contiguous gradient (e.g. the expanded
    gradient of `.mean()`) is copied first; the copy must stay alive until the launch is enqueued — were it dropped
    right after taking its address, the next temporary could be given the same block and overwrite it."""

    def __call__(self, t):
        c = _g(t)
        self.append(c)
        return _ptr(c)


class AffineCouplingFn(torch.autograd.Function):
    """(z, nn_out, scaling_factor, ldj) -> (z', ldj + layer_ldj); mask / reverse are constants."""

    @staticmethod
    def forward(ctx, z, nn_out, scaling_factor, ldj, mask, reverse):
        z_out, ldj_out = ops.affine_coupling(z, nn_out, scaling_factor, mask, reverse=reverse, ldj=ldj)
        ctx.save_for_backward(z_out, nn_out, scaling_factor if scaling_factor is not None else z_out.new_empty(0), mask if mask is not None else z_out.new_empty(0))
        ctx.has_sf, ctx.has_mask, ctx.has_ldj, ctx.reverse = scaling_factor is not None, mask is not None, ldj is not None, bool(reverse)
        return z_out, ldj_out

    @staticmethod
    def backward(ctx, g_zout, g_ldj):
        hold = _Hold()
        z_out, nn_out, sf, mask = ctx.saved_tensors
        sf = sf if ctx.has_sf else None
        mask = mask if ctx.has_mask else None
        dev = z_out.device
        B, N, D = z_out.shape
        nn_c = _f32(nn_out, "nn_out")
        sfc = _opt_f32(sf, "scaling_factor", dev)
        m, mr, mc = _mask_desc(mask, D, dev)
        g_z, g_nn = torch.empty_like(z_out), torch.empty_like(nn_c)
        g_sf = torch.empty(D, dtype=torch.float32, device=dev) if sf is not None else None
        ws = _ws(D, dev) if sf is not None else None
        _launch(dev, "cnf_affine_coupling_bwd", _ptr(z_out), _ptr(nn_c), _ptr(sfc), _ptr(m), mr, mc, hold(g_zout),
                                               hold(g_ldj), _ptr(g_z), _ptr(g_nn), _ptr(g_sf), _ptr(ws), B, N, D,
                                               int(ctx.reverse), _stream(dev))
        return g_z, g_nn.view_as(nn_out), (g_sf.view_as(sf) if sf is not None else None), (g_ldj if ctx.has_ldj else None), None, None


class AffineCouplingNllFn(torch.autograd.Function):
    """The LAST affine coupling of a flow + the NLL assembly in ONE forward kernel (cnf_affine_coupling_nll); the backward is the
    NLL assembly's and the coupling's backward kernels in one Function.  (z, nn_out, sf, ldj) -> (nll [B], z', ldj')."""

    @staticmethod
    def forward(ctx, z, nn_out, scaling_factor, ldj, mask, pad, length, sigma, log_sigma):
        z_out, ldj_out, _, nll = ops.affine_coupling_nll(z, nn_out, scaling_factor, mask, ldj=ldj, length=length,
                                                         channel_padding_mask=pad, sigma=sigma, log_sigma=log_sigma)
        empty = z_out.new_empty(0)
        ctx.save_for_backward(z_out, nn_out, scaling_factor if scaling_factor is not None else empty, mask if mask is not None else empty,
                              pad if isinstance(pad, torch.Tensor) else empty, length if isinstance(length, torch.Tensor) else empty)
        ctx.flags = (scaling_factor is not None, mask is not None, isinstance(pad, torch.Tensor), isinstance(length, torch.Tensor), ldj is not None)
        ctx.mark_non_differentiable(z_out, ldj_out)
        return nll, z_out, ldj_out

    @staticmethod
    def backward(ctx, g_nll, _g_z, _g_l):
        hold = _Hold()
        z_out, nn_out, sf, mask, pad, length = ctx.saved_tensors
        has_sf, has_mask, has_pad, has_len, has_ldj = ctx.flags
        g_zo, g_ldj = _prior_nll_bwd(z_out, length if has_len else None, pad if has_pad else None, g_nll, hold)
        sf = sf if has_sf else None
        dev = z_out.device
        B, N, D = z_out.shape
        nn_c = _f32(nn_out, "nn_out")
        sfc = _opt_f32(sf, "scaling_factor", dev)
        m, mr, mc = _mask_desc(mask if has_mask else None, D, dev)
        g_z, g_nn = torch.empty_like(z_out), torch.empty_like(nn_c)
        g_sf = torch.empty(D, dtype=torch.float32, device=dev) if sf is not None else None
        ws = _ws(D, dev) if sf is not None else None
        _launch(dev, "cnf_affine_coupling_bwd", _ptr(z_out), _ptr(nn_c), _ptr(sfc), _ptr(m), mr, mc, _ptr(g_zo), _ptr(g_ldj), _ptr(g_z),
                _ptr(g_nn), _ptr(g_sf), _ptr(ws), B, N, D, 0, _stream(dev))
        return (g_z, g_nn.view_as(nn_out), (g_sf.view_as(sf) if sf is not None else None), (g_ldj if has_ldj else None),
                None, None, None, None, None)


class ExtActNormFn(torch.autograd.Function):
    """(z, nn_out [B,N,2D], ldj) -> (z', ldj +- sum tanh(scales) pad)."""

    @staticmethod
    def forward(ctx, z, nn_out, ldj, pad, reverse):
        z_out, ldj_out = ops.ext_actnorm(z, nn_out, reverse=reverse, channel_padding_mask=pad,
                                         ldj=(ldj.clone() if ldj is not None else None))
        ctx.save_for_backward(z_out, nn_out, pad if isinstance(pad, torch.Tensor) else z_out.new_empty(0))
        ctx.has_pad, ctx.has_ldj, ctx.reverse = isinstance(pad, torch.Tensor), ldj is not None, bool(reverse)
        return z_out, ldj_out

    @staticmethod
    def backward(ctx, g_zout, g_ldj):
        hold = _Hold()
        z_out, nn_out, pad = ctx.saved_tensors
        dev = z_out.device
        B, N, D = z_out.shape
        nn_c = _f32(nn_out, "nn_out")
        p2 = _pad2d(pad, B, N, dev) if ctx.has_pad else None
        g_z, g_nn = torch.empty_like(z_out), torch.empty_like(nn_c)
        _launch(dev, "cnf_ext_actnorm_bwd", _ptr(z_out), _ptr(nn_c), _ptr(p2), hold(g_zout), hold(g_ldj), _ptr(g_z),
                                           _ptr(g_nn), B, N, D, int(ctx.reverse), _stream(dev))
        return g_z, g_nn.view_as(nn_out), (g_ldj if ctx.has_ldj else None), None, None


class ActNormFn(torch.autograd.Function):

    @staticmethod
    def forward(ctx, z, bias, scales, ldj, length, pad, reverse):
        z_out, ldj_out = ops.actnorm(z, bias, scales, reverse=reverse, length=length, channel_padding_mask=pad,
                                     ldj=(ldj.clone() if ldj is not None else None))
        empty = z_out.new_empty(0)
        ctx.save_for_backward(z_out, bias, scales, length if isinstance(length, torch.Tensor) else empty,
                              pad if isinstance(pad, torch.Tensor) else empty)
        ctx.has_len, ctx.has_pad, ctx.has_ldj, ctx.reverse = isinstance(length, torch.Tensor), isinstance(pad, torch.Tensor), ldj is not None, bool(reverse)
        return z_out, ldj_out

    @staticmethod
    def backward(ctx, g_zout, g_ldj):
        hold = _Hold()
        z_out, bias, scales, length, pad = ctx.saved_tensors
        dev = z_out.device
        B, N, D = z_out.shape
        p2 = _pad2d(pad, B, N, dev) if ctx.has_pad else None
        ln = _length(length, B, dev) if ctx.has_len else None
        g_z = torch.empty_like(z_out)
        g_b = torch.empty(D, dtype=torch.float32, device=dev)
        g_s = torch.empty(D, dtype=torch.float32, device=dev)
        ws = _ws(2 * D, dev)
        bias_c, scales_c = _f32(bias.reshape(-1), "bias"), _f32(scales.reshape(-1), "scales")
        _launch(dev, "cnf_actnorm_bwd", _ptr(z_out), _ptr(bias_c), _ptr(scales_c),
                                       _ptr(p2), _ptr(ln), hold(g_zout), hold(g_ldj), _ptr(g_z), _ptr(g_b), _ptr(g_s),
                                       _ptr(ws), B, N, D, int(ctx.reverse), _stream(dev))
        return g_z, g_b.view_as(bias), g_s.view_as(scales), (g_ldj if ctx.has_ldj else None), None, None, None


class InvConvFn(torch.autograd.Function):
    """(x, weight [D,D], sldj scalar, ldj) -> (x @ weight * pad, ldj +- sldj * len)."""

    @staticmethod
    def forward(ctx, x, weight, sldj, ldj, length, pad, reverse):
        z, ldj_out = ops.invconv(x, weight, sldj, reverse=reverse, length=length, channel_padding_mask=pad, ldj=ldj)
        empty = z.new_empty(0)
        ctx.save_for_backward(x, weight, sldj, length if isinstance(length, torch.Tensor) else empty,
                              pad if isinstance(pad, torch.Tensor) else empty)
        ctx.has_len, ctx.has_pad, ctx.has_ldj, ctx.reverse = isinstance(length, torch.Tensor), isinstance(pad, torch.Tensor), ldj is not None, bool(reverse)
        return z, ldj_out

    @staticmethod
    def backward(ctx, g_zout, g_ldj):
        hold = _Hold()
        x, weight, sldj, length, pad = ctx.saved_tensors
        dev = x.device
        B, N, D = x.shape
        xc, wc = _f32(x, "x"), _f32(weight, "weight")
        p2 = _pad2d(pad, B, N, dev) if ctx.has_pad else None
        ln = _length(length, B, dev) if ctx.has_len else None
        g_x = torch.empty_like(xc)
        g_w = torch.empty(D, D, dtype=torch.float32, device=dev)
        g_s = torch.empty(1, dtype=torch.float32, device=dev)
        ws = _ws(D * D + 1, dev)
        _launch(dev, "cnf_invconv_bwd", _ptr(xc), _ptr(wc), _ptr(p2), _ptr(ln), hold(g_zout), hold(g_ldj), _ptr(g_x),
                                       _ptr(g_w), _ptr(g_s), _ptr(ws), B, N, D, int(ctx.reverse), _stream(dev))
        return g_x, g_w, g_s.view_as(sldj), (g_ldj if ctx.has_ldj else None), None, None, None


def _known_inverse(weight):
    """W^-1 computed next to W by the LU weight assembly (InvertibleConv._build_weight stores it on the tensor), or None."""
    inv = getattr(weight, "_cnf_inverse", None)
    return inv if isinstance(inv, torch.Tensor) and inv.shape == weight.shape and inv.device == weight.device else None


def _actconv_bwd(saved, saved_is_output, bias, scales, weight, length, pad, g_zout, g_ldj, hold, weight_inv=None):
    """One launch of cnf_actnorm_invconv_bwd: (g_z, g_bias, g_scales, g_weight, g_sldj [1]).  weight_inv (with
    saved_is_output): W^-1 from the LU weight assembly; without it the library inverts W in a launch of its own."""
    dev = saved.device
    B, N, D = saved.shape
    sv, wc = _f32(saved, "z"), _f32(weight, "weight")
    bias_c, scales_c = _f32(bias.reshape(-1), "bias"), _f32(scales.reshape(-1), "scales")
    p2 = _pad2d(pad, B, N, dev) if isinstance(pad, torch.Tensor) else None
    ln = _length(length, B, dev) if isinstance(length, torch.Tensor) else None
    g_z = torch.empty_like(sv)
    g_p = torch.empty(D * D + 1 + 2 * D, dtype=torch.float32, device=dev)
    ws = _ws(D * D + 2 * D + 2, dev)
    wi = _f32(weight_inv, "weight_inv") if (saved_is_output and weight_inv is not None and weight_inv.numel() == D * D) else None
    status = _launch(dev, "cnf_actnorm_invconv_bwd", _ptr(sv), int(bool(saved_is_output)), _ptr(bias_c), _ptr(scales_c), _ptr(wc), _ptr(wi),
                     _ptr(p2), _ptr(ln), hold(g_zout), hold(g_ldj), _ptr(g_z), _ptr(g_p), _ptr(ws), B, N, D, _stream(dev),
                     allow_unsupported=True)
    if status != _lib.CNF_OK:              # D outside {1..6, 8}: the fused forward kernels do not exist there either
        raise ops.HipOnlyError("cnf_actnorm_invconv_bwd: " + _lib.load().cnf_last_error().decode())
    dd = D * D
    return g_z, g_p[dd + 1:dd + 1 + D].view_as(bias), g_p[dd + 1 + D:].view_as(scales), g_p[:dd].view(D, D), g_p[dd:dd + 1]


class ActConvFn(torch.autograd.Function):
    """ActNormFlow.forward -> InvertibleConv.forward of one flow step (forward direction): ONE forward kernel
    (cnf_actnorm_invconv) and ONE backward kernel (cnf_actnorm_invconv_bwd) that recomputes the pair's intermediate from the
    saved input.  (z, bias, scales, weight, sldj, ldj) -> (z', ldj'); the same numbers as ActNormFn then InvConvFn."""

    @staticmethod
    def forward(ctx, z, bias, scales, weight, sldj, ldj, length, pad):
        z_out, ldj_out = ops.actnorm_invconv(z, bias, scales, weight, sldj, length=length, channel_padding_mask=pad, ldj=ldj)
        empty = z_out.new_empty(0)
        ctx.save_for_backward(z, bias, scales, weight, length if isinstance(length, torch.Tensor) else empty,
                              pad if isinstance(pad, torch.Tensor) else empty)
        ctx.has_len, ctx.has_pad, ctx.has_ldj = isinstance(length, torch.Tensor), isinstance(pad, torch.Tensor), ldj is not None
        ctx.sldj_shape = sldj.shape
        return z_out, ldj_out

    @staticmethod
    def backward(ctx, g_zout, g_ldj):
        hold = _Hold()
        z, bias, scales, weight, length, pad = ctx.saved_tensors
        length, pad = (length if ctx.has_len else None), (pad if ctx.has_pad else None)
        g_z, g_b, g_s, g_w, g_sl = _actconv_bwd(z, False, bias, scales, weight, length, pad, g_zout, g_ldj, hold)
        return g_z, g_b, g_s, g_w, g_sl.view(ctx.sldj_shape), (g_ldj if ctx.has_ldj else None), None, None


class MixtureActConvFn(torch.autograd.Function):
    """Mixture-CDF coupling of one flow step + ActNorm + 1x1 convolution of the next: ONE forward kernel
    (cnf_mixture_coupling_actconv: the coupling's output stays in registers).  The backward recomputes what that output
    was from the saved result (cnf_actnorm_invconv_bwd, saved_is_output = 1), then runs the coupling's own backward kernel
    on the saved input.  (z, nn_out, sf, msf, bias, scales, weight, sldj, ldj) -> (z', ldj').
    Tolerance of the from-output form: the pair's input is rebuilt from fp32 outputs through W^-1 (inverted in fp64), so the error
    of g_weight / g_scales grows with cond(W) x fp32 epsilon — within 2e-3 of each tensor's largest entry up to cond(W) ~ 3000
    (log_s of the LU parametrisation spread over +-4: tests/test_gpu_backward.py); CNF_FUSE_TRAINING=0 runs the layers one by one,
    each from its own saved input, where a flow's convolutions are trained into worse conditioning than that."""

    @staticmethod
    def forward(ctx, z, nn_out, sf, msf, bias, scales, weight, sldj, ldj, mask, pad, length, K, reg_max, reg_factor, is_training):
        z_out, ldj_out, _ = ops.mixture_coupling_actconv(z, nn_out, mask, K, bias, scales, weight, sldj, scaling_factor=sf,
                                                         mixture_scaling_factor=msf, channel_padding_mask=pad, length=length,
                                                         reg_max=reg_max, reg_factor=reg_factor, is_training=is_training, ldj=ldj,
                                                         want_reg=False)
        empty = z_out.new_empty(0)
        inv = _known_inverse(weight)
        ctx.save_for_backward(z, nn_out, sf if sf is not None else empty, msf if msf is not None else empty,
                              mask if mask is not None else empty, pad if isinstance(pad, torch.Tensor) else empty,
                              length if isinstance(length, torch.Tensor) else empty, z_out, bias, scales, weight,
                              inv if inv is not None else empty)
        ctx.flags = (sf is not None, msf is not None, mask is not None, isinstance(pad, torch.Tensor), ldj is not None,
                     isinstance(length, torch.Tensor))
        ctx.cfg = (int(K), float(reg_max), float(reg_factor), bool(is_training))
        ctx.sldj_shape = sldj.shape
        return z_out, ldj_out

    @staticmethod
    def backward(ctx, g_zout, g_ldj):
        hold = _Hold()
        z, nn_out, sf, msf, mask, pad, length, z_out, bias, scales, weight, inv = ctx.saved_tensors
        has_sf, has_msf, has_mask, has_pad, has_ldj, has_len = ctx.flags
        K, reg_max, reg_factor, is_training = ctx.cfg
        pad_t, len_t = (pad if has_pad else None), (length if has_len else None)
        g_zc, g_b, g_s, g_w, g_sl = _actconv_bwd(z_out, True, bias, scales, weight, len_t, pad_t, g_zout, g_ldj, hold, weight_inv=inv)
        g_z, g_nn, g_sf, g_msf = _mixture_bwd(z, nn_out, sf if has_sf else None, msf if has_msf else None, mask if has_mask else None,
                                              pad_t, g_zc, g_ldj, K, reg_max, reg_factor, is_training, True, True, hold)
        return (g_z, g_nn, g_sf, g_msf, g_b, g_s, g_w, g_sl.view(ctx.sldj_shape), (g_ldj if has_ldj else None),
                None, None, None, None, None, None, None)


class EncoderActConvFn(torch.autograd.Function):
    """Categorical encoder (sampling its own logistic noise from the uniform draw) + ActNorm + 1x1 convolution of the first
    flow step: ONE forward kernel (cnf_encoder_forward_actconv).  The backward recovers the encoder's latents' gradient
    through cnf_actnorm_invconv_bwd (saved_is_output = 1) and re-samples the noise from the saved uniform draw for the
    encoder's own backward kernels.  (table, bias, scales, weight, sldj, ldj) -> (z', ldj')."""

    @staticmethod
    def forward(ctx, table, bias, scales, weight, sldj, ldj, categ, uniform, prior, pad, length, beta, squeeze):
        # class_prob_log rides along for the backward's pair kernel (cnf_encoder_forward_bwd_cpl)
        # (only when a gradient for the table can follow: ctx.needs_input_grad is False under torch.no_grad())
        keep_cpl = bool(ctx.needs_input_grad[0])
        res = ops.encoder_forward_actconv(categ, uniform, table, prior, bias, scales, weight, sldj, beta=beta,
                                          channel_padding_mask=pad, length=length, ldj=ldj, uniform_squeeze=squeeze,
                                          want_class_prob=keep_cpl)
        z_out, ldj_out = res[0], res[1]
        empty = z_out.new_empty(0)
        cpl = res[2] if len(res) > 2 and res[2] is not None else empty
        inv = _known_inverse(weight)
        ctx.save_for_backward(table, categ, uniform, prior, pad if isinstance(pad, torch.Tensor) else empty,
                              length if isinstance(length, torch.Tensor) else empty, z_out, bias, scales, weight, cpl,
                              inv if inv is not None else empty)
        ctx.has_pad, ctx.has_len, ctx.has_ldj = isinstance(pad, torch.Tensor), isinstance(length, torch.Tensor), ldj is not None
        ctx.beta, ctx.squeeze, ctx.sldj_shape = float(beta), float(squeeze), sldj.shape
        return z_out, ldj_out

    @staticmethod
    def backward(ctx, g_zout, g_ldj):
        hold = _Hold()
        table, categ, uniform, prior, pad, length, z_out, bias, scales, weight, cpl, inv = ctx.saved_tensors
        pad_t, len_t = (pad if ctx.has_pad else None), (length if ctx.has_len else None)
        g_ze, g_b, g_s, g_w, g_sl = _actconv_bwd(z_out, True, bias, scales, weight, len_t, pad_t, g_zout, g_ldj, hold, weight_inv=inv)
        dev = table.device
        B, N = categ.shape
        C, D = table.shape[0], table.shape[1] // 2
        eps = ops.logistic_from_uniform(uniform, mu=0.0, sigma=ops.LOGISTIC_SIGMA, eps=ctx.squeeze)
        tc, pc = _f32(table, "table"), _f32(prior, "category_prior")
        p2 = _pad2d(pad, B, N, dev) if ctx.has_pad else None
        g_table = torch.empty_like(tc)
        ws = torch.empty(int(_lib.load().cnf_encoder_bwd_tiled_workspace_floats(B, N, D, C)), dtype=torch.float32, device=dev)
        _launch(dev, "cnf_encoder_forward_bwd_cpl", _ptr(categ.contiguous()), _ptr(_f32(eps, "eps")), _ptr(tc), _ptr(pc), _ptr(p2), ctx.beta,
                (_ptr(cpl.contiguous()) if cpl.numel() else None), _ptr(g_ze), hold(g_ldj), _ptr(g_table), _ptr(ws), B, N, D, C,
                float(ops.LOGISTIC_SIGMA), float(ops.LOGISTIC_LOG_SIGMA), _stream(dev))
        return (g_table, g_b, g_s, g_w, g_sl.view(ctx.sldj_shape), (g_ldj if ctx.has_ldj else None),
                None, None, None, None, None, None, None)


class LUWeightFn(torch.autograd.Function):
    """InvertibleConv's LU-parametrised weight (permutation_layers.py:61-71): (l, u, log_s; p, sign_s constant) -> (W [D,D], sldj
    scalar) in ONE launch, its backward in one more — in place of ~11 tiny eager ops and the ~15 of their autograd per flow step."""

    @staticmethod
    def forward(ctx, l, u, log_s, p, sign_s):
        dev = l.device
        D = l.shape[0]
        lc, uc, sc, pc, gc = (_f32(t, n) for t, n in ((l, "l"), (u, "u"), (log_s, "log_s"), (p, "p"), (sign_s, "sign_s")))
        w = torch.empty(D, D, dtype=torch.float32, device=dev)
        sldj = torch.empty(1, dtype=torch.float32, device=dev)
        if D <= 8:
            # W^-1 rides along (third output, not differentiable): the backward of a fused group that kept only its output
            # (cnf_actnorm_invconv_bwd, saved_is_output = 1) needs it and would otherwise spend a launch on it
            w_inv = torch.empty(D, D, dtype=torch.float32, device=dev)
            _launch(dev, "cnf_invconv_lu_weight_inv", _ptr(pc), _ptr(lc), _ptr(uc), _ptr(sc), _ptr(gc), _ptr(w), _ptr(sldj), _ptr(w_inv), D,
                    _stream(dev))
        else:
            w_inv = w.new_empty(0)
            _launch(dev, "cnf_invconv_lu_weight", _ptr(pc), _ptr(lc), _ptr(uc), _ptr(sc), _ptr(gc), _ptr(w), _ptr(sldj), D, _stream(dev))
        ctx.save_for_backward(lc, uc, sc, pc, gc)
        ctx.mark_non_differentiable(w_inv)
        return w, sldj.reshape(()), w_inv

    @staticmethod
    def backward(ctx, g_w, g_sldj, _g_inv):
        hold = _Hold()
        l, u, log_s, p, sign_s = ctx.saved_tensors
        dev = l.device
        D = l.shape[0]
        g_l, g_u, g_s = torch.empty_like(l), torch.empty_like(u), torch.empty_like(log_s)
        _launch(dev, "cnf_invconv_lu_weight_bwd", _ptr(p), _ptr(l), _ptr(u), _ptr(log_s), _ptr(sign_s), hold(g_w),
                hold(g_sldj.reshape(1) if g_sldj is not None else None), _ptr(g_l), _ptr(g_u), _ptr(g_s), D, _stream(dev))
        return g_l, g_u, g_s, None, None


class LogisticLogProbFn(torch.autograd.Function):

    @staticmethod
    def forward(ctx, x, mu, sigma, log_sigma):
        ctx.save_for_backward(x)
        ctx.mu, ctx.sigma = float(mu), float(sigma)
        return ops.logistic_log_prob(x, mu=mu, sigma=sigma, log_sigma=log_sigma)

    @staticmethod
    def backward(ctx, g):
        hold = _Hold()
        (x,) = ctx.saved_tensors
        xc = _f32(x, "x")
        g_x = torch.empty_like(xc)
        _launch(xc.device, "cnf_logistic_log_prob_bwd", _ptr(xc), hold(g), _ptr(g_x), xc.numel(), ctx.mu, ctx.sigma,
                                                 _stream(xc.device))
        return g_x.view_as(x), None, None, None


class PriorNllFn(torch.autograd.Function):
    """(z, ldj) -> per-sample NLL [B] (length / padding are constants)."""

    @staticmethod
    def forward(ctx, z, ldj, length, pad):
        _, nll = ops.prior_nll(z, ldj, length=length, channel_padding_mask=pad)
        empty = z.new_empty(0)
        ctx.save_for_backward(z, length if isinstance(length, torch.Tensor) else empty, pad if isinstance(pad, torch.Tensor) else empty)
        ctx.has_len, ctx.has_pad = isinstance(length, torch.Tensor), isinstance(pad, torch.Tensor)
        return nll

    @staticmethod
    def backward(ctx, g_nll):
        z, length, pad = ctx.saved_tensors
        g_z, g_ldj = _prior_nll_bwd(z, length if ctx.has_len else None, pad if ctx.has_pad else None, g_nll, _Hold())
        return g_z, g_ldj, None, None


class SigmoidFlowFn(torch.autograd.Function):

    @staticmethod
    def forward(ctx, z, ldj, reverse, alpha):
        z_out, ldj_out = ops.sigmoid_flow(z, reverse=reverse, ldj=ldj, alpha=alpha)
        ctx.save_for_backward(z)
        ctx.reverse, ctx.alpha, ctx.has_ldj = bool(reverse), float(alpha), ldj is not None
        return z_out, ldj_out

    @staticmethod
    def backward(ctx, g_zout, g_ldj):
        hold = _Hold()
        (z,) = ctx.saved_tensors
        zc = _f32(z, "z")
        B = zc.shape[0]
        L = zc.numel() // B
        g_z = torch.empty_like(zc)
        _launch(zc.device, "cnf_sigmoid_flow_bwd", _ptr(zc), hold(g_zout), hold(g_ldj), _ptr(g_z), B, L, int(ctx.reverse),
                                            ctx.alpha, _stream(zc.device))
        return g_z, (g_ldj if ctx.has_ldj else None), None, None


def _mixture_bwd(z, nn_out, sf, msf, mask, pad, g_zout, g_ldj, K, reg_max, reg_factor, is_training, pit, pout, hold):
    """One launch of cnf_mixture_coupling_bwd_f32 (+ its reductions): (g_z, g_nn, g_sf or None, g_msf or None); sf / msf / mask /
    pad are tensors or None."""
    dev = z.device
    B, N, D = z.shape
    zc, nn_c = _f32(z, "z"), _f32(nn_out, "nn_out")
    sfc = _opt_f32(sf, "scaling_factor", dev) if sf is not None else None
    msfc = _opt_f32(msf, "mixture_scaling_factor", dev) if msf is not None else None
    m, mr, mc = _mask_desc(mask, D, dev)
    p2 = _pad2d(pad, B, N, dev) if pad is not None else None
    g_z, g_nn = torch.empty_like(zc), torch.empty_like(nn_c)
    g_sf = torch.empty(D, dtype=torch.float32, device=dev) if sf is not None else None
    g_msf = torch.empty(D, K, dtype=torch.float32, device=dev) if msf is not None else None
    ws = _ws(D + D * K, dev)
    act, n_act = ops._act_list(mask, m, mr, mc, D)
    compact = ops.mixture_layout(nn_c, zc, K, act, n_act)
    args = lambda nn_t, g_nn_t: (_ptr(zc), _ptr(nn_t), _ptr(sfc), _ptr(msfc), _ptr(m), mr, mc, act, n_act, _ptr(p2), int(pit), int(pout),
                                 hold(g_zout), hold(g_ldj), _ptr(g_z), _ptr(g_nn_t), _ptr(g_sf), _ptr(g_msf), _ptr(ws),
                                 B, N, D, K, reg_max, reg_factor, int(is_training), _stream(dev))
    if compact:
        # the compact layout (cnf_mixture_coupling_compact_bwd_f32): no zero blocks are written; a shape / mode its kernel declines
        # goes through the reference layout and is sliced back
        if _launch(dev, "cnf_mixture_coupling_compact_bwd_f32", *args(nn_c, g_nn), allow_unsupported=True) != _lib.CNF_OK:
            full = ops.expand_compact(nn_c, zc.shape, K, act, n_act)
            g_full = torch.empty_like(full)
            _launch(dev, "cnf_mixture_coupling_bwd_f32", *args(full, g_full))
            d0 = int(act[0])
            g_nn = g_full.view(B, N, D, 2 + 3 * K)[:, :, d0:d0 + n_act].reshape(nn_c.shape)
    else:
        _launch(dev, "cnf_mixture_coupling_bwd_f32", *args(nn_c, g_nn))
    return (g_z, g_nn.view_as(nn_out), (g_sf.view_as(sf) if sf is not None else None),
            (g_msf.view_as(msf) if msf is not None else None))


def _prior_nll_bwd(z, length, pad, g_nll, hold):
    """cnf_prior_nll_bwd: (g_z, g_ldj) of the NLL assembly."""
    dev = z.device
    B, N, D = z.shape
    zc = _f32(z, "z")
    p2 = _pad2d(pad, B, N, dev) if pad is not None else None
    ln = _length(length, B, dev) if length is not None else None
    g_z = torch.empty_like(zc)
    g_ldj = torch.empty(B, dtype=torch.float32, device=dev)
    _launch(dev, "cnf_prior_nll_bwd", _ptr(zc), _ptr(p2), _ptr(ln), hold(g_nll), _ptr(g_z), _ptr(g_ldj), B, N, D,
                                     float(ops.LOGISTIC_SIGMA), _stream(dev))
    return g_z, g_ldj


class MixtureCouplingFn(torch.autograd.Function):
    """(z, nn_out, scaling_factor, mixture_scaling_factor, ldj) -> (z', ldj + layer_ldj, reg_sum); forward direction only."""

    @staticmethod
    def forward(ctx, z, nn_out, sf, msf, ldj, mask, pad, K, reg_max, reg_factor, is_training, pad_in_transform, pad_output):
        z_out, ldj_out, reg = ops.mixture_coupling(z, nn_out, mask, K, sf, msf, reverse=False, channel_padding_mask=pad,
                                                   reg_max=reg_max, reg_factor=reg_factor, is_training=is_training,
                                                   pad_in_transform=pad_in_transform, pad_output=pad_output, ldj=ldj)
        empty = z_out.new_empty(0)
        ctx.save_for_backward(z, nn_out, sf if sf is not None else empty, msf if msf is not None else empty,
                              mask if mask is not None else empty, pad if isinstance(pad, torch.Tensor) else empty)
        ctx.flags = (sf is not None, msf is not None, mask is not None, isinstance(pad, torch.Tensor), ldj is not None)
        ctx.cfg = (int(K), float(reg_max), float(reg_factor), bool(is_training), bool(pad_in_transform), bool(pad_output))
        ctx.mark_non_differentiable(reg)
        return z_out, ldj_out, reg

    @staticmethod
    def backward(ctx, g_zout, g_ldj, _g_reg):
        hold = _Hold()
        z, nn_out, sf, msf, mask, pad = ctx.saved_tensors
        has_sf, has_msf, has_mask, has_pad, has_ldj = ctx.flags
        K, reg_max, reg_factor, is_training, pit, pout = ctx.cfg
        g_z, g_nn, g_sf, g_msf = _mixture_bwd(z, nn_out, sf if has_sf else None, msf if has_msf else None, mask if has_mask else None,
                                              pad if has_pad else None, g_zout, g_ldj, K, reg_max, reg_factor, is_training, pit, pout, hold)
        return (g_z, g_nn, g_sf, g_msf, (g_ldj if has_ldj else None), None, None, None, None, None, None, None, None)


class MixtureCouplingNllFn(torch.autograd.Function):
    """The LAST mixture coupling of a flow + the NLL assembly (set_modeling/task.py:96-118) in ONE forward kernel
    (cnf_mixture_coupling_nll); the backward is the NLL assembly's and the coupling's backward kernels in one Function.
    (z, nn_out, sf, msf, ldj) -> (nll [B], z', ldj'); z' and ldj' are for reports only (not differentiable)."""

    @staticmethod
    def forward(ctx, z, nn_out, sf, msf, ldj, mask, pad, length, K, reg_max, reg_factor, is_training, sigma, log_sigma):
        z_out, ldj_out, _, _, nll = ops.mixture_coupling_nll(z, nn_out, mask, K, sf, msf, channel_padding_mask=pad, reg_max=reg_max,
                                                             reg_factor=reg_factor, is_training=is_training, ldj=ldj, length=length,
                                                             want_reg=False, sigma=sigma, log_sigma=log_sigma)
        empty = z_out.new_empty(0)
        ctx.save_for_backward(z, nn_out, sf if sf is not None else empty, msf if msf is not None else empty,
                              mask if mask is not None else empty, pad if isinstance(pad, torch.Tensor) else empty,
                              length if isinstance(length, torch.Tensor) else empty, z_out)
        ctx.flags = (sf is not None, msf is not None, mask is not None, isinstance(pad, torch.Tensor), ldj is not None,
                     isinstance(length, torch.Tensor))
        ctx.cfg = (int(K), float(reg_max), float(reg_factor), bool(is_training))
        ctx.mark_non_differentiable(z_out, ldj_out)
        return nll, z_out, ldj_out

    @staticmethod
    def backward(ctx, g_nll, _g_z, _g_l):
        hold = _Hold()
        z, nn_out, sf, msf, mask, pad, length, z_out = ctx.saved_tensors
        has_sf, has_msf, has_mask, has_pad, has_ldj, has_len = ctx.flags
        K, reg_max, reg_factor, is_training = ctx.cfg
        pad_t = pad if has_pad else None
        g_zo, g_ldj = _prior_nll_bwd(z_out, length if has_len else None, pad_t, g_nll, hold)
        g_z, g_nn, g_sf, g_msf = _mixture_bwd(z, nn_out, sf if has_sf else None, msf if has_msf else None, mask if has_mask else None,
                                              pad_t, g_zo, g_ldj, K, reg_max, reg_factor, is_training, True, True, hold)
        return (g_z, g_nn, g_sf, g_msf, (g_ldj if has_ldj else None), None, None, None, None, None, None, None, None, None)


class EncoderForwardFn(torch.autograd.Function):
    """class table [C,2D] -> (z, ldj, class_prob_log); categories, noise, prior and padding are constants."""

    @staticmethod
    def forward(ctx, table, categ, eps, prior, pad, beta, want_class_prob, tiled=None, uniform_squeeze=None):
        # uniform_squeeze: `eps` is the uniform draw and the forward kernel samples the noise itself (and hands it back for
        # the backward kernels), see ops.encoder_forward
        # the backward's pair kernel takes every token's denominator from the forward's class_prob_log (4 bytes per token
        # more to write here, a whole sweep over the classes less there): ask for it whenever a gradient can follow
        # (ctx.needs_input_grad, not table.requires_grad: a Parameter still "requires grad" under torch.no_grad(), where no backward
        # can follow — evaluation and sampling passes must not write and keep 4 bytes per token for nothing)
        keep_cpl = bool(ctx.needs_input_grad[0])
        if uniform_squeeze is not None:
            z, ldj, cpl, eps = ops.encoder_forward(categ, eps, table, prior, beta=beta, channel_padding_mask=pad,
                                                   want_class_prob=want_class_prob or keep_cpl, tiled=tiled,
                                                   uniform_squeeze=uniform_squeeze, want_noise=True)
        else:
            z, ldj, cpl = ops.encoder_forward(categ, eps, table, prior, beta=beta, channel_padding_mask=pad,
                                              want_class_prob=want_class_prob or keep_cpl, tiled=tiled)
        ctx.tiled = tiled
        empty = z.new_empty(0)
        ctx.save_for_backward(table, categ, eps, prior, pad if isinstance(pad, torch.Tensor) else empty,
                              cpl if keep_cpl else empty)
        ctx.has_pad, ctx.beta, ctx.has_cpl = isinstance(pad, torch.Tensor), float(beta), keep_cpl
        if cpl is None or not want_class_prob:
            cpl = empty
        ctx.mark_non_differentiable(cpl)
        return z, ldj, cpl

    @staticmethod
    def backward(ctx, g_zout, g_ldj, _g_cpl):
        hold = _Hold()
        table, categ, eps, prior, pad, cpl = ctx.saved_tensors
        dev = table.device
        B, N = categ.shape
        C, D = table.shape[0], table.shape[1] // 2
        tc, ec, pc = _f32(table, "table"), _f32(eps, "eps"), _f32(prior, "category_prior")
        p2 = _pad2d(pad, B, N, dev) if ctx.has_pad else None
        g_table = torch.empty_like(tc)
        categ_c = categ.contiguous()
        # one pass over the (token, class) pairs with the forward's class_prob_log, or the token-lane + class-lane passes (the
        # library picks by shape): any vocabulary size, bit-reproducible.  (Round 1's LDS-table kernel with floating-point
        # atomics — 13x slower at 16 classes, not reproducible — is gone since round 5; `tiled` selects the forward kernel only.)
        ws = torch.empty(int(_lib.load().cnf_encoder_bwd_tiled_workspace_floats(B, N, D, C)), dtype=torch.float32, device=dev)
        _launch(dev, "cnf_encoder_forward_bwd_cpl", _ptr(categ_c), _ptr(ec), _ptr(tc), _ptr(pc), _ptr(p2), ctx.beta,
                _ptr(cpl.contiguous()) if ctx.has_cpl else None, hold(g_zout), hold(g_ldj), _ptr(g_table), _ptr(ws), B, N, D, C,
                float(ops.LOGISTIC_SIGMA), float(ops.LOGISTIC_LOG_SIGMA), _stream(dev))
        return g_table, None, None, None, None, None, None, None, None


class EncoderForwardDevBetaFn(torch.autograd.Function):
    """EncoderForwardFn with the posterior weight beta in a DEVICE scalar (a captured training step follows the beta schedule by
    writing that scalar between replays; the C ABI takes beta by value).  The token log-det is affine in beta —
    ldj_tok = (beta * class_prob_log - (log p(eps) - ldj_flow)) * pad (linear_encoding.py:120-133) — so the forward runs the
    kernel at beta = 1 and adds (beta - 1) * sum_n class_prob_log * pad, and the backward combines the kernel's table gradients
    at beta = 0 and beta = 1: g(beta) = g(0) + beta * (g(1) - g(0)).  Exact up to rounding; twice the encoder backward."""

    @staticmethod
    def forward(ctx, table, categ, uniform, prior, pad, beta_t, squeeze):
        z, ldj1, cpl, eps = ops.encoder_forward(categ, uniform, table, prior, beta=1.0, channel_padding_mask=pad,
                                                want_class_prob=True, uniform_squeeze=squeeze, want_noise=True)
        B, N = categ.shape
        w = cpl.view(B, N)
        if isinstance(pad, torch.Tensor):
            w = w * pad.reshape(B, N)
        ldj = ldj1 + (beta_t.reshape(()).to(torch.float32) - 1.0) * w.sum(dim=1)
        ctx.save_for_backward(table, categ, eps, prior, pad if isinstance(pad, torch.Tensor) else z.new_empty(0), beta_t, cpl)
        ctx.has_pad = isinstance(pad, torch.Tensor)
        ctx.mark_non_differentiable(cpl)
        return z, ldj, cpl

    @staticmethod
    def backward(ctx, g_zout, g_ldj, _g_cpl):
        hold = _Hold()
        table, categ, eps, prior, pad, beta_t, cpl = ctx.saved_tensors
        dev = table.device
        B, N = categ.shape
        C, D = table.shape[0], table.shape[1] // 2
        tc, ec, pc = _f32(table, "table"), _f32(eps, "eps"), _f32(prior, "category_prior")
        p2 = _pad2d(pad, B, N, dev) if ctx.has_pad else None
        categ_c = categ.contiguous()
        cpl_c = cpl.contiguous()                          # log q_c does not depend on beta: one forward serves both launches
        n_ws = int(_lib.load().cnf_encoder_bwd_tiled_workspace_floats(B, N, D, C))
        grads = []
        for beta in (0.0, 1.0):
            g_table = torch.empty_like(tc)
            ws = torch.empty(n_ws, dtype=torch.float32, device=dev)
            _launch(dev, "cnf_encoder_forward_bwd_cpl", _ptr(categ_c), _ptr(ec), _ptr(tc), _ptr(pc), _ptr(p2), beta, _ptr(cpl_c),
                    hold(g_zout), hold(g_ldj), _ptr(g_table), _ptr(ws), B, N, D, C,
                    float(ops.LOGISTIC_SIGMA), float(ops.LOGISTIC_LOG_SIGMA), _stream(dev))
            grads.append(g_table)
        g = grads[0] + beta_t.reshape(()).to(torch.float32) * (grads[1] - grads[0])
        return g, None, None, None, None, None, None


class AffineParamsFn(torch.autograd.Function):
    """CouplingLayer.get_coup_params: (nn_out, scaling_factor) -> (s, t)."""

    @staticmethod
    def forward(ctx, nn_out, sf, mask):
        s, t = ops.affine_params(nn_out, mask, sf)
        empty = s.new_empty(0)
        ctx.save_for_backward(nn_out, sf if sf is not None else empty, mask if mask is not None else empty)
        ctx.flags = (sf is not None, mask is not None)
        return s, t

    @staticmethod
    def backward(ctx, g_s, g_t):
        hold = _Hold()
        nn_out, sf, mask = ctx.saved_tensors
        has_sf, has_mask = ctx.flags
        dev = nn_out.device
        nn_c = _f32(nn_out, "nn_out")
        B, N = nn_c.shape[0], nn_c.shape[1]
        D = nn_c.shape[-1] // 2
        sfc = _opt_f32(sf, "scaling_factor", dev) if has_sf else None
        m, mr, mc = _mask_desc(mask if has_mask else None, D, dev)
        g_nn = torch.empty_like(nn_c)
        g_sf = torch.empty(D, dtype=torch.float32, device=dev) if has_sf else None
        ws = _ws(D, dev) if has_sf else None
        _launch(dev, "cnf_affine_params_bwd", _ptr(nn_c), _ptr(sfc), _ptr(m), mr, mc, hold(g_s), hold(g_t), _ptr(g_nn),
                                             _ptr(g_sf), _ptr(ws), B, N, D, _stream(dev))
        return g_nn.view_as(nn_out), (g_sf.view_as(sf) if has_sf else None), None


class AffineTransformFn(torch.autograd.Function):
    """CouplingLayer.run_with_params: (z, s, t) -> (z', ldj)."""

    @staticmethod
    def forward(ctx, z, s, t, reverse):
        z_out, ldj = ops.affine_transform(z, s, t, reverse=reverse)
        ctx.save_for_backward(z_out, s.expand_as(z).contiguous(), t.expand_as(z).contiguous())
        ctx.reverse, ctx.shapes = bool(reverse), (s.shape, t.shape)
        return z_out, ldj

    @staticmethod
    def backward(ctx, g_zout, g_ldj):
        hold = _Hold()
        z_out, s, t = ctx.saved_tensors
        dev = z_out.device
        B, N, D = z_out.shape
        g_z, g_s, g_t = torch.empty_like(z_out), torch.empty_like(z_out), torch.empty_like(z_out)
        _launch(dev, "cnf_affine_transform_bwd", _ptr(z_out), _ptr(s), _ptr(t), hold(g_zout), hold(g_ldj), _ptr(g_z),
                                                _ptr(g_s), _ptr(g_t), B, N, D, int(ctx.reverse), _stream(dev))
        return g_z, g_s.sum_to_size(ctx.shapes[0]), g_t.sum_to_size(ctx.shapes[1]), None


class MixtureParamsFn(torch.autograd.Function):
    """MixtureCDFCoupling.get_mixt_params: (nn_out, scaling_factor, mixture_scaling_factor) -> five fp64 tensors."""

    @staticmethod
    def forward(ctx, nn_out, sf, msf, mask, K):
        out = ops.mixture_params(nn_out, mask, K, sf, msf)
        empty = nn_out.new_empty(0)
        ctx.save_for_backward(nn_out, sf if sf is not None else empty, msf if msf is not None else empty,
                              mask if mask is not None else empty)
        ctx.flags, ctx.K = (sf is not None, msf is not None, mask is not None), int(K)
        return out

    @staticmethod
    def backward(ctx, g_t, g_log_s, g_log_pi, g_mu, g_ls):
        nn_out, sf, msf, mask = ctx.saved_tensors
        has_sf, has_msf, has_mask = ctx.flags
        K = ctx.K
        dev = nn_out.device
        nn_c = _f32(nn_out, "nn_out")
        P = 2 + 3 * K
        D = nn_c.shape[-1] // P
        B = nn_c.shape[0]
        N = nn_c.numel() // (B * D * P)
        sfc = _opt_f32(sf, "scaling_factor", dev) if has_sf else None
        msfc = _opt_f32(msf, "mixture_scaling_factor", dev) if has_msf else None
        m, mr, mc = _mask_desc(mask if has_mask else None, D, dev)
        g_nn = torch.empty_like(nn_c)
        g_sf = torch.empty(D, dtype=torch.float32, device=dev) if has_sf else None
        g_msf = torch.empty(D, K, dtype=torch.float32, device=dev) if has_msf else None
        ws = _ws(D + D * K, dev)
        dd = lambda t: None if t is None else (t.double().contiguous())
        gs = [dd(x) for x in (g_t, g_log_s, g_log_pi, g_mu, g_ls)]
        _launch(dev, "cnf_mixture_params_bwd", _ptr(nn_c), _ptr(sfc), _ptr(msfc), _ptr(m), mr, mc, *[_ptr(x) for x in gs], _ptr(g_nn),
                                              _ptr(g_sf), _ptr(g_msf), _ptr(ws), B, N, D, K, _stream(dev))
        return g_nn.view_as(nn_out), (g_sf.view_as(sf) if has_sf else None), (g_msf.view_as(msf) if has_msf else None), None, None


class MixtureTransformFn(torch.autograd.Function):
    """MixtureCDFCoupling.run_with_params (forward direction) on fp64 split parameters."""

    @staticmethod
    def forward(ctx, z, t, log_s, log_pi, mu, ls, mask, pad, reg_max, reg_factor, is_training):
        z_out, ldj, reg = ops.mixture_transform(z, t, log_s, log_pi, mu, ls, reverse=False, reg_max=reg_max, reg_factor=reg_factor,
                                                mask=mask, channel_padding_mask=pad, is_training=is_training)
        empty = z_out.new_empty(0)
        ctx.save_for_backward(z, t, log_s, log_pi, mu, ls, mask if mask is not None else empty,
                              pad if isinstance(pad, torch.Tensor) else empty)
        ctx.flags = (mask is not None, isinstance(pad, torch.Tensor))
        ctx.cfg = (float(reg_max), float(reg_factor), bool(is_training))
        ctx.mark_non_differentiable(reg)
        return z_out, ldj, reg

    @staticmethod
    def backward(ctx, g_zout, g_ldj, _g_reg):
        z, t, log_s, log_pi, mu, ls, mask, pad = ctx.saved_tensors
        has_mask, has_pad = ctx.flags
        reg_max, reg_factor, is_training = ctx.cfg
        z64 = ops._f64(z, "orig_z")
        dev = z64.device
        B, N, D = z64.shape
        K = log_pi.shape[-1]
        kshape = tuple(z64.shape) + (K,)
        t64, s64 = ops._f64(t, "t", z64.shape), ops._f64(log_s, "log_s", z64.shape)
        pi64, mu64, ls64 = ops._f64(log_pi, "log_pi", kshape), ops._f64(mu, "mixt_t", kshape), ops._f64(ls, "mixt_log_s", kshape)
        m, mr, mc = _mask_desc(mask if has_mask else None, D, dev)
        p2 = _pad2d(pad, B, N, dev) if has_pad else None
        g_z, g_t, g_s = torch.empty_like(z64), torch.empty_like(z64), torch.empty_like(z64)
        g_pi, g_mu, g_ls = torch.empty_like(pi64), torch.empty_like(pi64), torch.empty_like(pi64)
        # fp64 copies of the upstream gradients: named, so that both stay alive until the launch is enqueued
        gz64 = None if g_zout is None else g_zout.double().contiguous()
        gl64 = None if g_ldj is None else g_ldj.double().contiguous()
        _launch(dev, "cnf_mixture_transform_bwd", _ptr(z64), _ptr(t64), _ptr(s64), _ptr(pi64), _ptr(mu64), _ptr(ls64), _ptr(m), mr, mc, _ptr(p2),
                                                 _ptr(gz64), _ptr(gl64), _ptr(g_z), _ptr(g_t), _ptr(g_s), _ptr(g_pi), _ptr(g_mu), _ptr(g_ls),
                                                 B, N, D, K, reg_max, reg_factor, int(is_training), _stream(dev))
        return (g_z.to(z.dtype), g_t.sum_to_size(t.shape), g_s.sum_to_size(log_s.shape), g_pi.sum_to_size(log_pi.shape),
                g_mu.sum_to_size(mu.shape), g_ls.sum_to_size(ls.shape), None, None, None, None, None)


def needs_grad(*tensors):
    return torch.is_grad_enabled() and any(isinstance(t, torch.Tensor) and t.requires_grad for t in tensors)
