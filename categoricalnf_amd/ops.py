"""Tensor-level entry points: torch CUDA tensors in, HIP kernels (libcnf_hip.so) on the current
stream, torch tensors out.  PyTorch is plumbing here (device memory + streams); no arithmetic of the
hot path is done with torch ops and there is no CPU path — CPU tensors raise.
"""
import ctypes
import math
import os
import weakref

import numpy as np
import torch

from . import _lib

LOGISTIC_SIGMA = 1.0 / 1.81                       # reference: layers/flows/distributions.py:95
LOGISTIC_LOG_SIGMA = float(np.log(LOGISTIC_SIGMA))

_STRICT = os.environ.get("CNF_STRICT_ASSERTS", "0") == "1"
CAPTURING = False            # set by graphs.GraphedFlow while a pass is recorded into a HIP graph (no host syncs)
FUSE_LAYERS = os.environ.get("CNF_FUSE_LAYERS", "1") == "1"      # FlowModel: ActNorm + InvertibleConv in one kernel
FUSE_LU_WEIGHT = os.environ.get("CNF_FUSE_LU_WEIGHT", "1") == "1"  # InvertibleConv: W = P L U and its log-det in one launch (cnf_invconv_lu_weight)
LU_WEIGHT_MAX_D = 16
FUSE_TRAINING = os.environ.get("CNF_FUSE_TRAINING", "1") == "1"  # the same groups with autograd on (one Function and one backward per group)
_flags = {}


class HipOnlyError(RuntimeError):
    pass


def _dev(t):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise HipOnlyError("categoricalnf_amd kernels run on MI355X only: expected a CUDA(HIP) tensor, got %s. "
                           "There is no CPU fallback (the CPU oracle lives in oracle/ and is test-only)."
                           % (t.device if isinstance(t, torch.Tensor) else type(t)))
    return t.device


def _f32(t, name):
    # the common case first (a contiguous fp32 device tensor): these helpers are most of a small launch's host time
    if type(t) is torch.Tensor and t.dtype is torch.float32 and t.is_cuda and t.is_contiguous():
        return t
    _dev(t)
    if t.dtype != torch.float32:
        raise TypeError("%s must be float32, got %s" % (name, t.dtype))
    return t if t.is_contiguous() else t.contiguous()


def _opt_f32(t, name, device):
    if t is None:
        return None
    if not isinstance(t, torch.Tensor):
        raise TypeError("%s must be a tensor" % name)
    if t.dtype is torch.float32 and t.device == device and t.is_contiguous():
        return t
    if t.device != device:
        t = t.to(device)
    if t.dtype != torch.float32:
        t = t.float()
    return t if t.is_contiguous() else t.contiguous()


def _ptr(t):
    # an empty tensor has a null data pointer; hand the kernels a harmless non-null address instead
    # (every entry point returns before touching memory when B == 0)
    if t is None:
        return None
    return ctypes.c_void_p(t.data_ptr() if t.numel() > 0 else 8)


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream(device):
    """The current HIP stream of `device` as a void*: the raw handle straight from torch's C side (0.3 us) where this torch has
    that entry point, else through the Stream object (3 us)."""
    if _raw_stream is not None:
        idx = device.index
        return ctypes.c_void_p(_raw_stream(idx if idx is not None else torch.cuda.current_device()))
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


_fn_cache = {}


def _launch(dev, name, *args, allow_unsupported=False):
    """Call one C-ABI entry point with `dev` as the current HIP device (a stream can only be launched into
    from its own device; multi-GPU-per-process callers such as nn.DataParallel threads rely on this).
    `allow_unsupported`: CNF_ERR_UNSUPPORTED (nothing was launched) is returned to the caller, which then takes the
    unfused route; every other failure raises as usual.  Returns the status."""
    fn = _fn_cache.get(name)
    if fn is None:
        fn = _fn_cache[name] = getattr(_lib.load(), name)
    if dev.index is not None and dev.index != torch.cuda.current_device():
        with torch.cuda.device(dev):
            status = fn(*args)
    else:
        status = fn(*args)
    if status != _lib.CNF_OK:
        if allow_unsupported and status == _lib.CNF_ERR_UNSUPPORTED:
            return status
        # a failed launch may have left a split-row workspace (tickets, fixed-point sums) half-written: the next
        # launch must not inherit it
        _mix_ws.clear()
        _lib.check(status, name)
    return status


def flag_word(device):
    """The per-device int32 word the kernels OR their CNF_FLAG_* bits into."""
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
    w = _flags.get(key)
    if w is None:
        w = torch.zeros(1, dtype=torch.int32, device=device)
        _flags[key] = w
    return w


def check_flags(device=None, where=""):
    """Turn device-side failure bits into the reference's exceptions (one host sync).

    NaN in z / ldj -> AssertionError (flow_model.py:42, activation_normalization.py:45-46);
    inverse-CDF input outside (0,1) -> RuntimeError (mixture_cdf_layer.py:238-239)."""
    if CAPTURING:
        return
    for key, w in list(_flags.items()):
        if device is not None and (device.type, device.index if device.index is not None
                                   else torch.cuda.current_device()) != key:
            continue
        bits = int(w.item())
        if bits:
            w.zero_()
            if bits & _lib.FLAG_RANGE:
                raise RuntimeError('Inverse logisitic CDF got y outside (0, 1)')
            if bits & _lib.FLAG_CATEGORY:
                raise AssertionError("[!] ERROR: One-hot tensor has larger entries than classes (category index outside "
                                     "[0, num_classes)). %s" % where)
            what = []
            if bits & _lib.FLAG_NAN_Z:
                what.append("latent values (z)")
            if bits & _lib.FLAG_NAN_LDJ:
                what.append("log-det (ldj)")
            raise AssertionError("[!] ERROR: Found NaN %s. %s" % (" and ".join(what), where))


def _after(device, where):
    if _STRICT:
        check_flags(device, where)


def _mask_desc(mask, D, device):
    """(ptr-holder tensor, rows, cols) of a coupling mask buffer shaped [1,D], [rows,1], [1,1,D], [1,rows,1]..."""
    if mask is None:
        return None, 0, 0
    cached = getattr(mask, "_cnf_mask_desc", None)
    if cached is not None and cached[0] == (mask._version, D, device):
        return cached[1]
    desc = _mask_desc_uncached(mask, D, device)
    try:
        mask._cnf_mask_desc = ((mask._version, D, device), desc)       # masks are module buffers: described once
    except Exception:
        pass
    return desc


def _mask_desc_uncached(mask, D, device):
    m = mask
    while m.dim() > 2:
        if m.size(0) != 1:
            raise ValueError("coupling mask must broadcast over the batch, got shape %s" % (tuple(mask.shape),))
        m = m[0]
    if m.dim() == 1:
        m = m.view(1, -1)
    rows, cols = int(m.size(0)), int(m.size(1))
    if cols not in (1, D):
        raise ValueError("coupling mask last dim must be 1 or %d, got %s" % (D, tuple(mask.shape)))
    return _opt_f32(m, "mask", device), rows, cols


def _pad2d(pad, B, N, device):
    """[B,N,1] (or [B,N], [B*N,1,1]) 0/1 padding mask -> contiguous fp32 [B,N]."""
    if pad is None:
        return None
    if not isinstance(pad, torch.Tensor):
        raise TypeError("channel_padding_mask must be a tensor")
    if pad.numel() != B * N:
        if pad.dim() == 3 and pad.size(0) == B and pad.size(1) == N:
            # mask expanded over channels (e.g. the data-init path passes [B,N,D]); all channels agree
            pad = pad[:, :, 0]
        else:
            raise ValueError("padding mask with %d entries does not match [B=%d, N=%d]" % (pad.numel(), B, N))
    return _opt_f32(pad.reshape(B, N), "channel_padding_mask", device)


def _length(length, B, device):
    if length is None:
        return None
    if not isinstance(length, torch.Tensor):
        return torch.full((B,), float(length), dtype=torch.float32, device=device)
    return _opt_f32(length.reshape(-1), "length", device)


def _ldj_io(ldj, B, device, inplace):
    """returns (ldj_in tensor or None, ldj_out tensor)"""
    if ldj is None:
        return None, torch.empty(B, dtype=torch.float32, device=device)
    ldj = _f32(ldj, "ldj")
    return ldj, (ldj if inplace else torch.empty_like(ldj))


# ------------------------------------------------------------------------------------------------
def affine_coupling(z, nn_out, scaling_factor, mask, reverse=False, ldj=None):
    z = _f32(z, "z")
    dev = z.device
    B, N, D = z.shape
    nn_out = _f32(nn_out, "nn_out")
    if nn_out.numel() != z.numel() * 2:
        raise ValueError("nn_out must be [B,N,2D]; got %s for z %s" % (tuple(nn_out.shape), tuple(z.shape)))
    sf = _opt_f32(scaling_factor, "scaling_factor", dev)
    m, mr, mc = _mask_desc(mask, D, dev)
    ldj_in, ldj_out = _ldj_io(ldj, B, dev, inplace=False)
    z_out = torch.empty_like(z)
    _launch(dev, "cnf_affine_coupling", _ptr(z), _ptr(nn_out), _ptr(sf), _ptr(m), mr, mc, _ptr(ldj_in),
                                       _ptr(z_out), _ptr(ldj_out), B, N, D, int(bool(reverse)),
                                       _ptr(flag_word(dev)), _stream(dev))
    _after(dev, "affine coupling")
    return z_out, ldj_out


def affine_coupling_actconv(z, nn_out, scaling_factor, mask, an_bias, an_scales, conv_weight, conv_sldj, reverse=False,
                            length=None, channel_padding_mask=None, ldj=None):
    """Affine coupling of one flow step + ActNorm + 1x1 convolution of the next step (forward), or the coupling's inverse + the
    inverted convolution and ActNorm of its own step (reverse; conv_weight = the inverse weight), in one kernel
    (cnf_affine_coupling_actconv); where that kernel does not apply the two kernels run one after the other — the same bits
    either way.  Returns (z_out, ldj_out)."""
    z = _f32(z, "z")
    dev = z.device
    B, N, D = z.shape
    nn_out = _f32(nn_out, "nn_out")
    if nn_out.numel() != z.numel() * 2:
        raise ValueError("nn_out must be [B,N,2D]; got %s for z %s" % (tuple(nn_out.shape), tuple(z.shape)))
    sf = _opt_f32(scaling_factor, "scaling_factor", dev)
    m, mr, mc = _mask_desc(mask, D, dev)
    pad = _pad2d(channel_padding_mask, B, N, dev)
    ln = _length(length, B, dev) if isinstance(length, torch.Tensor) else None
    b, sc = _f32(an_bias.reshape(-1), "bias"), _f32(an_scales.reshape(-1), "scales")
    w, sl = _f32(conv_weight, "weight"), _f32(conv_sldj.reshape(1), "sldj")
    ldj_in, ldj_out = _ldj_io(ldj, B, dev, inplace=False)
    z_out = torch.empty_like(z)
    status = _launch(dev, "cnf_affine_coupling_actconv", _ptr(z), _ptr(nn_out), _ptr(sf), _ptr(m), mr, mc, _ptr(ldj_in), _ptr(z_out),
                     _ptr(ldj_out), _ptr(b), _ptr(sc), _ptr(w), _ptr(sl), _ptr(pad), _ptr(ln), B, N, D, int(bool(reverse)),
                     _ptr(flag_word(dev)), _stream(dev), allow_unsupported=True)
    if status != _lib.CNF_OK:
        zc, lc = affine_coupling(z, nn_out, scaling_factor, mask, reverse=reverse, ldj=ldj)
        if D in FUSED_ACTCONV_DIMS:
            return actnorm_invconv(zc, an_bias, an_scales, conv_weight, conv_sldj, reverse=reverse, length=length,
                                   channel_padding_mask=channel_padding_mask, ldj=lc)
        kw = dict(reverse=reverse, length=length, channel_padding_mask=channel_padding_mask)
        if reverse:
            zc, lc = invconv(zc, conv_weight, conv_sldj, ldj=lc, **kw)
            return actnorm(zc, an_bias, an_scales, ldj=lc, **kw)
        zc, lc = actnorm(zc, an_bias, an_scales, ldj=lc, **kw)
        return invconv(zc, conv_weight, conv_sldj, ldj=lc, **kw)
    _after(dev, "affine coupling + ActNorm + InvertibleConv")
    return z_out, ldj_out


def affine_coupling_nll(z, nn_out, scaling_factor, mask, ldj=None, length=None, channel_padding_mask=None, sums=None,
                        sigma=LOGISTIC_SIGMA, log_sigma=LOGISTIC_LOG_SIGMA, acc=None):
    """Last coupling layer + NLL assembly in one kernel: == affine_coupling(reverse=False) then prior_nll on its
    outputs.  Returns (z_out, ldj_out, neglog [B], nll [B])."""
    z = _f32(z, "z")
    dev = z.device
    B, N, D = z.shape
    nn_out = _f32(nn_out, "nn_out")
    if nn_out.numel() != z.numel() * 2:
        raise ValueError("nn_out must be [B,N,2D]; got %s for z %s" % (tuple(nn_out.shape), tuple(z.shape)))
    sf = _opt_f32(scaling_factor, "scaling_factor", dev)
    m, mr, mc = _mask_desc(mask, D, dev)
    ldj_in, ldj_out = _ldj_io(ldj, B, dev, inplace=False)
    pad = _pad2d(channel_padding_mask, B, N, dev)
    ln = _length(length, B, dev)
    z_out = torch.empty_like(z)
    neglog = torch.empty(B, dtype=torch.float32, device=dev)
    nll = torch.empty(B, dtype=torch.float32, device=dev)
    if acc is not None:
        # batch sum inside the kernel (int64 [64] fixed-point partials, accumulated; read with nll_acc_read)
        if acc.dtype != torch.int64 or acc.numel() < NLL_ACC_SLOTS or not acc.is_contiguous() or sums is not None:
            raise ValueError("acc must be a contiguous int64 tensor with at least %d elements (and excludes `sums`)" % NLL_ACC_SLOTS)
        _launch(dev, "cnf_affine_coupling_nll_acc", _ptr(z), _ptr(nn_out), _ptr(sf), _ptr(m), mr, mc, _ptr(ldj_in),
                                                   _ptr(z_out), _ptr(ldj_out), _ptr(pad), _ptr(ln), _ptr(neglog), _ptr(nll),
                                                   _ptr(acc), B, N, D, float(sigma), float(log_sigma),
                                                   _ptr(flag_word(dev)), _stream(dev))
    else:
        _launch(dev, "cnf_affine_coupling_nll", _ptr(z), _ptr(nn_out), _ptr(sf), _ptr(m), mr, mc, _ptr(ldj_in),
                                               _ptr(z_out), _ptr(ldj_out), _ptr(pad), _ptr(ln), _ptr(neglog), _ptr(nll),
                                               _ptr(sums), B, N, D, float(sigma), float(log_sigma),
                                               _ptr(flag_word(dev)), _stream(dev))
    _after(dev, "affine coupling + NLL")
    return z_out, ldj_out, neglog, nll


def affine_params(nn_out, mask, scaling_factor=None):
    nn_out = _f32(nn_out, "nn_out")
    dev = nn_out.device
    B, N = nn_out.shape[0], nn_out.shape[1]
    D = nn_out.shape[-1] // 2
    sf = _opt_f32(scaling_factor, "scaling_factor", dev)
    m, mr, mc = _mask_desc(mask, D, dev)
    s = torch.empty(B, N, D, dtype=torch.float32, device=dev)
    t = torch.empty_like(s)
    _launch(dev, "cnf_affine_params", _ptr(nn_out), _ptr(sf), _ptr(m), mr, mc, _ptr(s), _ptr(t), B, N, D,
                                     _stream(dev))
    return s, t


def affine_transform(z, s, t, reverse=False):
    z, s, t = _f32(z, "z"), _f32(s, "s"), _f32(t, "t")
    dev = z.device
    B, N, D = z.shape
    if s.shape != z.shape or t.shape != z.shape:
        s, t = s.expand_as(z).contiguous(), t.expand_as(z).contiguous()
    z_out = torch.empty_like(z)
    ldj = torch.empty(B, dtype=torch.float32, device=dev)
    _launch(dev, "cnf_affine_transform", _ptr(z), _ptr(s), _ptr(t), None, _ptr(z_out), _ptr(ldj), B, N, D,
                                        int(bool(reverse)), _ptr(flag_word(dev)), _stream(dev))
    return z_out, ldj


def actnorm(z, bias, scales, reverse=False, length=None, channel_padding_mask=None, ldj=None):
    """In-place on `ldj` when given (the reference's `ldj +=`, activation_normalization.py:37,40)."""
    z = _f32(z, "z")
    dev = z.device
    B, N, D = z.shape
    b = _f32(bias.reshape(-1), "bias")
    s = _f32(scales.reshape(-1), "scales")
    pad = _pad2d(channel_padding_mask, B, N, dev)
    ln = _length(length, B, dev)
    ldj_in, ldj_out = _ldj_io(ldj, B, dev, inplace=True)
    z_out = torch.empty_like(z)
    _launch(dev, "cnf_actnorm", _ptr(z), _ptr(b), _ptr(s), _ptr(pad), _ptr(ln), _ptr(ldj_in), _ptr(z_out),
                               _ptr(ldj_out), B, N, D, int(bool(reverse)), _ptr(flag_word(dev)), _stream(dev))
    _after(dev, "ActNorm")
    return z_out, ldj_out


def actnorm_data_init(x, channel_padding_mask=None):
    """bias = -mean, scales = -0.5 log var over (batch, sequence), weighted by the padding mask."""
    x = _f32(x, "input_data")
    dev = x.device
    B, N, D = x.shape
    pad = _pad2d(channel_padding_mask, B, N, dev)
    acc = torch.zeros(D + 1, dtype=torch.float64, device=dev)
    _launch(dev, "cnf_actnorm_stats", _ptr(x), _ptr(pad), None, _ptr(acc), B, N, D, 0, _stream(dev))
    from .distributed import allreduce_init_stats          # no-op unless distributed.sync_data_init() was switched on
    allreduce_init_stats(acc)
    mean = acc[:D] / acc[D]
    acc2 = torch.zeros(D + 1, dtype=torch.float64, device=dev)
    mean_c = mean.contiguous()
    _launch(dev, "cnf_actnorm_stats", _ptr(x), _ptr(pad), _ptr(mean_c), _ptr(acc2), B, N, D, 1,
                                     _stream(dev))
    allreduce_init_stats(acc2)
    var = acc2[:D] / acc[D]
    bias = (-mean).float().view(1, 1, D)
    scales = (-0.5 * var.log()).float().view(1, 1, D)
    return bias, scales


def ext_actnorm(z, nn_out, reverse=False, channel_padding_mask=None, ldj=None):
    z = _f32(z, "z")
    dev = z.device
    B, N, D = z.shape
    nn_out = _f32(nn_out, "nn_out")
    if nn_out.numel() != 2 * z.numel():
        raise ValueError("predictor output must be [B,N,2D]")
    pad = _pad2d(channel_padding_mask, B, N, dev) if isinstance(channel_padding_mask, torch.Tensor) else None
    ldj_in, ldj_out = _ldj_io(ldj, B, dev, inplace=True)
    z_out = torch.empty_like(z)
    _launch(dev, "cnf_ext_actnorm", _ptr(z), _ptr(nn_out), _ptr(pad), _ptr(ldj_in), _ptr(z_out), _ptr(ldj_out),
                                   B, N, D, int(bool(reverse)), _ptr(flag_word(dev)), _stream(dev))
    _after(dev, "ExtActNorm")
    return z_out, ldj_out


def invconv(x, weight, sldj, reverse=False, length=None, channel_padding_mask=None, ldj=None):
    """weight must already be the inverse for reverse=True (fp64 inverse, as the reference)."""
    x = _f32(x, "x")
    dev = x.device
    B, N, D = x.shape
    w = _f32(weight, "weight")
    sl = _f32(sldj.reshape(1), "sldj")
    pad = _pad2d(channel_padding_mask, B, N, dev)
    ln = _length(length, B, dev) if isinstance(length, torch.Tensor) else None
    ldj_in, ldj_out = _ldj_io(ldj, B, dev, inplace=False)
    z_out = torch.empty_like(x)
    _launch(dev, "cnf_invconv", _ptr(x), _ptr(w), _ptr(sl), _ptr(pad), _ptr(ln), _ptr(ldj_in), _ptr(z_out),
                               _ptr(ldj_out), B, N, D, int(bool(reverse)), _ptr(flag_word(dev)), _stream(dev))
    _after(dev, "InvertibleConv")
    return z_out, ldj_out


FUSED_ACTCONV_DIMS = (1, 2, 3, 4, 5, 6, 8)


def actnorm_invconv(z, bias, scales, weight, sldj, reverse=False, length=None, channel_padding_mask=None, ldj=None):
    """ActNorm then 1x1 convolution (reverse: the pair backwards) in one pass; `weight` already inverted for reverse."""
    z = _f32(z, "z")
    dev = z.device
    B, N, D = z.shape
    b, s = _f32(bias.reshape(-1), "bias"), _f32(scales.reshape(-1), "scales")
    w, sl = _f32(weight, "weight"), _f32(sldj.reshape(1), "sldj")
    pad = _pad2d(channel_padding_mask, B, N, dev)
    ln = _length(length, B, dev) if isinstance(length, torch.Tensor) else None
    ldj_in, ldj_out = _ldj_io(ldj, B, dev, inplace=False)
    z_out = torch.empty_like(z)
    _launch(dev, "cnf_actnorm_invconv", _ptr(z), _ptr(b), _ptr(s), _ptr(w), _ptr(sl), _ptr(pad), _ptr(ln), _ptr(ldj_in),
                                       _ptr(z_out), _ptr(ldj_out), B, N, D, int(bool(reverse)), _ptr(flag_word(dev)),
                                       _stream(dev))
    _after(dev, "ActNorm + InvertibleConv")
    return z_out, ldj_out


def _act_list(mask, mask2d, rows, cols, D):
    """Host list of the transformed channels of a channel mask (rows == 1), else (None, 0).

    Reading the D mask values costs a device-to-host copy, so the list is cached — per mask TENSOR OBJECT (a module's
    registered buffer is the same object on every call), validated by a weak reference and the tensor's version
    counter: another tensor that happens to live at the same address, or an in-place update (load_state_dict), is
    read again."""
    if mask2d is None or rows != 1 or cols != D:
        return None, 0
    key = id(mask)
    hit = _act_cache.get(key)
    if hit is not None and hit[0]() is mask and hit[1] == mask._version and hit[2] == D:
        return hit[3]
    host = mask2d.detach().reshape(-1).cpu().tolist()      # D floats
    idx = [i for i, v in enumerate(host) if v == 0.0]
    res = ((ctypes.c_int * max(len(idx), 1))(*idx), len(idx))
    if len(_act_cache) > 256:
        _act_cache.clear()
    try:
        _act_cache[key] = (weakref.ref(mask), mask._version, D, res)
    except TypeError:
        pass
    return res


_act_cache = {}


_mix_ws = {}


def _mixture_workspace(dev, B):
    """Zero-filled scratch through which several workgroups share one long row (cnf_mixture_coupling_ws): one per
    (device, stream); the kernels leave it zeroed, so it is filled once."""
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), torch.cuda.current_stream(dev).cuda_stream)
    need = int(_lib.load().cnf_mixture_workspace_bytes(int(B)))
    if torch.cuda.is_current_stream_capturing():
        # a buffer allocated during a capture belongs to the graph's private pool: the graph keeps it (its replays
        # leave it zeroed like every launch), the cache must not hand it to eager launches
        return torch.zeros(max(need, 1 << 16), dtype=torch.uint8, device=dev)
    w = _mix_ws.get(key)
    if w is None or w.numel() < need:
        w = torch.zeros(max(need, 1 << 16), dtype=torch.uint8, device=dev)
        if len(_mix_ws) > 64:
            _mix_ws.clear()
        _mix_ws[key] = w
    return w


def mixture_layout(nn_out, z, K, act, n_act):
    """Which layout `nn_out` has for latents z [B,N,D]: False = the reference's [B,N,D*P] (blocks for all channels), True = the
    compact [B,N,n_act*P] (blocks of the transformed channels of a channel mask only: cnf_mixture_coupling_compact*)."""
    P = 2 + 3 * int(K)
    if nn_out.numel() == z.numel() * P:
        return False
    B, N, D = z.shape
    if act is not None and 0 < n_act < D and nn_out.numel() == B * N * n_act * P:
        return True
    raise ValueError("nn_out must be [B,N,D*(2+3K)] (or, with a channel mask, the compact [B,N,n_act*(2+3K)]); got %s for z %s, K=%d"
                     % (tuple(nn_out.shape), tuple(z.shape), K))


def expand_compact(nn_compact, z_shape, K, act, n_act):
    """The compact parameter tensor scattered into the reference layout (zero blocks for the untransformed channels): the route of
    shapes / modes the compact-layout kernels decline (CNF_ERR_UNSUPPORTED)."""
    B, N, D = z_shape
    P = 2 + 3 * int(K)
    full = torch.zeros(B, N, D, P, dtype=nn_compact.dtype, device=nn_compact.device)
    d0 = int(act[0])
    full[:, :, d0:d0 + n_act] = nn_compact.reshape(B, N, n_act, P)
    return full.view(B, N, D * P)


def mixture_coupling(z, nn_out, mask, num_mixtures, scaling_factor=None, mixture_scaling_factor=None,
                     reverse=False, channel_padding_mask=None, reg_max=-1, reg_factor=1, is_training=True,
                     pad_in_transform=True, pad_output=True, ldj=None, want_reg=True):
    """Returns (z_out fp32, ldj fp32 [B], reg_sum fp32 [B] or None)."""
    z = _f32(z, "z")
    dev = z.device
    B, N, D = z.shape
    K = int(num_mixtures)
    nn_out = _f32(nn_out, "nn_out")
    sf = _opt_f32(scaling_factor, "scaling_factor", dev)
    msf = _opt_f32(mixture_scaling_factor, "mixture_scaling_factor", dev)
    m, mr, mc = _mask_desc(mask, D, dev)
    act, n_act = _act_list(mask, m, mr, mc, D)
    compact = mixture_layout(nn_out, z, K, act, n_act)
    pad = _pad2d(channel_padding_mask, B, N, dev)
    ldj_in, ldj_out = _ldj_io(ldj, B, dev, inplace=False)
    z_out = torch.empty_like(z)
    use_reg = (not reverse) and reg_max > 0 and is_training
    reg = torch.zeros(B, dtype=torch.float32, device=dev) if want_reg else None
    ws = _mixture_workspace(dev, B)
    for name in (("cnf_mixture_coupling_compact", "cnf_mixture_coupling_ws") if compact else ("cnf_mixture_coupling_ws",)):
        status = _launch(dev, name, _ptr(z), _ptr(nn_out), _ptr(sf), _ptr(msf), _ptr(m), mr, mc,
                         act, n_act, _ptr(pad), int(bool(pad_in_transform)), int(bool(pad_output)),
                         _ptr(ldj_in), _ptr(z_out), _ptr(ldj_out), _ptr(reg if use_reg else None),
                         B, N, D, K, int(bool(reverse)), float(reg_max), float(reg_factor),
                         int(bool(is_training)), _ptr(ws), int(ws.numel()),
                         _ptr(flag_word(dev)), _stream(dev), allow_unsupported=compact and name.endswith("compact"))
        if status == _lib.CNF_OK:
            break
        nn_out = expand_compact(nn_out, z.shape, K, act, n_act)       # declined: the reference layout through the plain entry point
    _after(dev, "mixture-CDF coupling")
    return z_out, ldj_out, reg


def mixture_coupling_nll(z, nn_out, mask, num_mixtures, scaling_factor=None, mixture_scaling_factor=None,
                         channel_padding_mask=None, reg_max=-1, reg_factor=1, is_training=True,
                         pad_in_transform=True, pad_output=True, ldj=None, length=None, want_reg=True, acc=None,
                         sigma=LOGISTIC_SIGMA, log_sigma=LOGISTIC_LOG_SIGMA):
    """Forward mixture coupling as the last layer of a flow + the NLL assembly in one kernel: == mixture_coupling then
    prior_nll on its outputs.  Returns (z_out, ldj_out, reg or None, neglog [B], nll [B])."""
    z = _f32(z, "z")
    dev = z.device
    B, N, D = z.shape
    K = int(num_mixtures)
    nn_out = _f32(nn_out, "nn_out")
    sf = _opt_f32(scaling_factor, "scaling_factor", dev)
    msf = _opt_f32(mixture_scaling_factor, "mixture_scaling_factor", dev)
    m, mr, mc = _mask_desc(mask, D, dev)
    act, n_act = _act_list(mask, m, mr, mc, D)
    compact = mixture_layout(nn_out, z, K, act, n_act)
    pad = _pad2d(channel_padding_mask, B, N, dev)
    ln = _length(length, B, dev)
    ldj_in, ldj_out = _ldj_io(ldj, B, dev, inplace=False)
    z_out = torch.empty_like(z)
    neglog = torch.empty(B, dtype=torch.float32, device=dev)
    nll = torch.empty(B, dtype=torch.float32, device=dev)
    use_reg = reg_max > 0 and is_training
    reg = torch.zeros(B, dtype=torch.float32, device=dev) if want_reg else None
    if acc is not None and (acc.dtype != torch.int64 or acc.numel() < NLL_ACC_SLOTS or not acc.is_contiguous()):
        raise ValueError("acc must be a contiguous int64 tensor with at least %d elements" % NLL_ACC_SLOTS)
    ws = _mixture_workspace(dev, B)
    for name in (("cnf_mixture_coupling_compact_nll", "cnf_mixture_coupling_nll") if compact else ("cnf_mixture_coupling_nll",)):
        status = _launch(dev, name, _ptr(z), _ptr(nn_out), _ptr(sf), _ptr(msf), _ptr(m), mr, mc,
                         act, n_act, _ptr(pad), int(bool(pad_in_transform)), int(bool(pad_output)),
                         _ptr(ldj_in), _ptr(z_out), _ptr(ldj_out), _ptr(reg if use_reg else None),
                         _ptr(ln), _ptr(neglog), _ptr(nll), _ptr(acc),
                         B, N, D, K, float(reg_max), float(reg_factor), int(bool(is_training)),
                         float(sigma), float(log_sigma), _ptr(ws), int(ws.numel()),
                         _ptr(flag_word(dev)), _stream(dev), allow_unsupported=compact and name.endswith("compact_nll"))
        if status == _lib.CNF_OK:
            break
        nn_out = expand_compact(nn_out, z.shape, K, act, n_act)
    _after(dev, "mixture-CDF coupling + NLL")
    return z_out, ldj_out, reg, neglog, nll


def mixture_coupling_actconv(z, nn_out, mask, num_mixtures, an_bias, an_scales, conv_weight, conv_sldj,
                             scaling_factor=None, mixture_scaling_factor=None, channel_padding_mask=None, length=None,
                             reg_max=-1, reg_factor=1, is_training=True, ldj=None, want_reg=True):
    """Forward mixture coupling of one flow step + ActNorm + 1x1 convolution of the next step in one kernel
    (== mixture_coupling then actnorm_invconv, bit for bit).  Returns (z_out, ldj_out, reg or None)."""
    z = _f32(z, "z")
    dev = z.device
    B, N, D = z.shape
    K = int(num_mixtures)
    nn_out = _f32(nn_out, "nn_out")
    sf = _opt_f32(scaling_factor, "scaling_factor", dev)
    msf = _opt_f32(mixture_scaling_factor, "mixture_scaling_factor", dev)
    m, mr, mc = _mask_desc(mask, D, dev)
    act, n_act = _act_list(mask, m, mr, mc, D)
    compact = mixture_layout(nn_out, z, K, act, n_act)
    pad = _pad2d(channel_padding_mask, B, N, dev)
    ln = _length(length, B, dev) if isinstance(length, torch.Tensor) else None
    b, sc = _f32(an_bias.reshape(-1), "bias"), _f32(an_scales.reshape(-1), "scales")
    w, sl = _f32(conv_weight, "weight"), _f32(conv_sldj.reshape(1), "sldj")
    ldj_in, ldj_out = _ldj_io(ldj, B, dev, inplace=False)
    z_out = torch.empty_like(z)
    use_reg = reg_max > 0 and is_training
    reg = torch.zeros(B, dtype=torch.float32, device=dev) if want_reg else None
    ws = _mixture_workspace(dev, B)
    for name in (("cnf_mixture_coupling_compact_actconv", "cnf_mixture_coupling_actconv") if compact else ("cnf_mixture_coupling_actconv",)):
        status = _launch(dev, name, _ptr(z), _ptr(nn_out), _ptr(sf), _ptr(msf), _ptr(m), mr, mc, act, n_act,
                         _ptr(pad), _ptr(ldj_in), _ptr(z_out), _ptr(ldj_out), _ptr(reg if use_reg else None),
                         _ptr(b), _ptr(sc), _ptr(w), _ptr(sl), _ptr(ln),
                         B, N, D, K, float(reg_max), float(reg_factor), int(bool(is_training)),
                         _ptr(ws), int(ws.numel()), _ptr(flag_word(dev)), _stream(dev),
                         allow_unsupported=compact and name.endswith("compact_actconv"))
        if status == _lib.CNF_OK:
            break
        nn_out = expand_compact(nn_out, z.shape, K, act, n_act)
    _after(dev, "mixture-CDF coupling + ActNorm + InvertibleConv")
    return z_out, ldj_out, reg


def mixture_params(nn_out, mask, num_mixtures, scaling_factor=None, mixture_scaling_factor=None):
    nn_out = _f32(nn_out, "nn_out")
    dev = nn_out.device
    K = int(num_mixtures)
    P = 2 + 3 * K
    lead = tuple(nn_out.shape[:-1])
    D = nn_out.shape[-1] // P
    B = lead[0]
    N = int(np.prod(lead[1:])) if len(lead) > 1 else 1
    sf = _opt_f32(scaling_factor, "scaling_factor", dev)
    msf = _opt_f32(mixture_scaling_factor, "mixture_scaling_factor", dev)
    m, mr, mc = _mask_desc(mask, D, dev)
    mk = lambda *s: torch.empty(*s, dtype=torch.float64, device=dev)
    t, log_s = mk(*lead, D), mk(*lead, D)
    log_pi, mu, ls = mk(*lead, D, K), mk(*lead, D, K), mk(*lead, D, K)
    _launch(dev, "cnf_mixture_params", _ptr(nn_out), _ptr(sf), _ptr(msf), _ptr(m), mr, mc, _ptr(t), _ptr(log_s),
                                      _ptr(log_pi), _ptr(mu), _ptr(ls), B, N, D, K, _stream(dev))
    return t, log_s, log_pi, mu, ls


def _f64(t, name, shape=None):
    _dev(t)
    if t.dtype != torch.float64:
        t = t.double()
    if shape is not None and tuple(t.shape) != tuple(shape):
        t = t.expand(shape)
    return t if t.is_contiguous() else t.contiguous()


def mixture_transform(orig_z, t, log_s, log_pi, mixt_t, mixt_log_s, reverse=False, reg_max=-1, reg_factor=1,
                      mask=None, channel_padding_mask=None, is_training=True):
    """fp64 in / fp64 out, as MixtureCDFCoupling.run_with_params.  Returns (z_out, ldj[B], reg_ldj[B,N,D])."""
    z = _f64(orig_z, "orig_z")
    dev = z.device
    B, N, D = z.shape
    K = log_pi.shape[-1]
    t, log_s = _f64(t, "t", z.shape), _f64(log_s, "log_s", z.shape)
    kshape = tuple(z.shape) + (K,)
    log_pi, mixt_t, mixt_log_s = _f64(log_pi, "log_pi", kshape), _f64(mixt_t, "mixt_t", kshape), _f64(mixt_log_s, "mixt_log_s", kshape)
    m, mr, mc = _mask_desc(mask, D, dev)
    act, n_act = _act_list(mask, m, mr, mc, D)
    pad = _pad2d(channel_padding_mask, B, N, dev)
    z_out = torch.empty_like(z)
    ldj = torch.empty(B, dtype=torch.float64, device=dev)
    reg = torch.empty_like(z)
    _launch(dev, "cnf_mixture_transform", _ptr(z), _ptr(t), _ptr(log_s), _ptr(log_pi), _ptr(mixt_t), _ptr(mixt_log_s),
                                         _ptr(m), mr, mc, act, n_act, _ptr(pad), _ptr(z_out), _ptr(ldj), _ptr(reg),
                                         B, N, D, K, int(bool(reverse)), float(reg_max), float(reg_factor),
                                         int(bool(is_training)), _ptr(flag_word(dev)), _stream(dev))
    _after(dev, "mixture-CDF transform")
    return z_out, ldj, reg


# ------------------------------------------------------------------------------------------------
def logistic_log_prob(x, mu=0.0, sigma=LOGISTIC_SIGMA, log_sigma=LOGISTIC_LOG_SIGMA):
    x = _f32(x, "x")
    dev = x.device
    out = torch.empty_like(x)
    _launch(dev, "cnf_logistic_log_prob", _ptr(x), _ptr(out), x.numel(), float(mu), float(sigma), float(log_sigma),
                                         _ptr(flag_word(dev)), _stream(dev))
    _after(dev, "log-prob of distribution")
    return out


def logistic_from_uniform(u, mu=0.0, sigma=LOGISTIC_SIGMA, eps=1e-4):
    u = _f32(u, "u")
    out = torch.empty_like(u)
    _launch(u.device, "cnf_logistic_from_uniform", _ptr(u), _ptr(out), u.numel(), float(mu), float(sigma), float(eps),
                                             _stream(u.device))
    return out


def prior_nll(z, ldj, length=None, channel_padding_mask=None, sums=None,
              sigma=LOGISTIC_SIGMA, log_sigma=LOGISTIC_LOG_SIGMA):
    """Returns (neglog [B], nll [B]); `sums` (fp64 [2], optional) is set to (sum nll, B) of this batch."""
    z = _f32(z, "z")
    dev = z.device
    B, N, D = z.shape
    pad = _pad2d(channel_padding_mask, B, N, dev)
    ln = _length(length, B, dev)
    ldj = _opt_f32(ldj, "ldj", dev)
    neglog = torch.empty(B, dtype=torch.float32, device=dev)
    nll = torch.empty(B, dtype=torch.float32, device=dev)
    _launch(dev, "cnf_prior_nll", _ptr(z), _ptr(pad), _ptr(ldj), _ptr(ln), _ptr(neglog), _ptr(nll), _ptr(sums),
                                 B, N, D, float(sigma), float(log_sigma), _stream(dev))
    return neglog, nll


def encoder_fused_supported(C, D):
    """Whether the LDS-resident mixture-model encoder kernels (cnf_encoder_forward / cnf_encoder_decode / the backward)
    take this vocabulary: the derived class table [C, 6D+3] and the row partials must fit 64 KiB of LDS, D <= 16
    (C <= 530 at D = 4, <= 227 at D = 10).  Larger vocabularies (wikitext: 10^4 classes) run the class-tiled kernels
    (cnf_encoder_forward_tiled / cnf_encoder_decode_tiled / cnf_encoder_forward_bwd_tiled)."""
    return D <= 16 and 4 * 512 * 4 + C * (6 * D + 3) * 4 <= 64 * 1024


def encoder_prefers_tiled_forward(C, D, B=None, N=None):
    """Forward pass: the class-tiled kernel when the class table does not fit LDS, or — for a small batch of long rows
    (fewer than 256 rows of at least 64 tokens: the LDS-resident kernel then runs one workgroup per row) — from ~9 KB of
    table on.  Measured after round 3's density-sum loop and tile sizes (profiles/r03_encoder_lds_vs_tiled.txt; LDS-resident
    vs tiled, us): 2048 x 64 tokens, D = 6: 30.6 vs 33.3 at 160 classes, 42.4 vs 47.1 at 256, 57.0 vs 62.2 at 350;
    16384 x 16, D = 4: 36.5 vs 41.7 at 160, 73.4 vs 79.8 at 350; 128 x 288, D = 3: 17.0 vs 18.6 at 100 classes but 24.8
    vs 20.9 at 160 and 70.4 vs 50.5 at 500; 64 x 703, D = 2: 27.7 vs 22.7 at 160.  (Round 2 switched at 9 KB for every
    shape: its LDS-resident kernel streamed a log-sum-exp on 256-token tiles and lost from 64 classes on.)  Decode keeps
    the LDS table whenever it fits (equal or slightly faster)."""
    if not encoder_fused_supported(C, D):
        return True
    return B is not None and N is not None and B < 256 and N >= 64 and C * (6 * D + 3) * 4 >= 9000


def encoder_forward(categ, eps, table, category_prior, beta=1.0, channel_padding_mask=None, ldj=None,
                    want_class_prob=False, sigma=LOGISTIC_SIGMA, log_sigma=LOGISTIC_LOG_SIGMA, tiled=None,
                    uniform_squeeze=None, want_noise=False):
    """`tiled`: None = by vocabulary size; True / False force the class-tiled / the LDS-resident kernel (tests).

    `uniform_squeeze` (the prior's eps, 1e-4 in the reference): `eps` then holds the UNIFORM draw and the kernel samples the
    logistic noise itself (cnf_encoder_forward_sampled: LogisticDistribution.sample fused in, one launch and a [T, D] round
    trip less, the same bits); `want_noise` returns that noise as a fourth value (the backward takes it).  In math mode 0
    the sampler's fp64 logit is not what the kernel holds: the two calls are made instead."""
    dev = _dev(categ)
    if categ.dtype != torch.int64:
        categ = categ.long()
    categ = categ.contiguous()
    B, N = categ.shape
    table = _f32(table, "table")
    C, D = table.shape[0], table.shape[1] // 2
    eps = _f32(eps, "eps")
    if eps.numel() != B * N * D:
        raise ValueError("noise must have B*N*D entries")
    prior = _f32(category_prior, "category_prior")
    pad = _pad2d(channel_padding_mask, B, N, dev)
    ldj_in, ldj_out = _ldj_io(ldj, B, dev, inplace=False)
    z = torch.empty(B, N, D, dtype=torch.float32, device=dev)
    cpl = torch.empty(B * N, dtype=torch.float32, device=dev) if want_class_prob else None
    if tiled is None:
        tiled = encoder_prefers_tiled_forward(C, D, B, N)
    # token log-det terms (+ the per-split partials above 1024 classes): only the class-tiled kernels take a workspace
    ws = torch.empty(int(_lib.load().cnf_encoder_workspace_floats(B, N, D, C)), dtype=torch.float32, device=dev) if tiled else None
    noise = None
    if uniform_squeeze is not None:
        noise = torch.empty(B * N, D, dtype=torch.float32, device=dev) if want_noise else None
        head = (_ptr(categ), _ptr(eps), float(uniform_squeeze), _ptr(table), _ptr(prior), _ptr(pad), float(beta), _ptr(ldj_in),
                _ptr(z), _ptr(ldj_out), _ptr(cpl), _ptr(noise))
        tail = (B, N, D, C, float(sigma), float(log_sigma), _ptr(flag_word(dev)), _stream(dev))
        if not tiled:
            status = _launch(dev, "cnf_encoder_forward_sampled", *head, *tail, allow_unsupported=True)
        else:
            status = _launch(dev, "cnf_encoder_forward_tiled_sampled", *head, _ptr(ws), *tail, allow_unsupported=True)
        if status == _lib.CNF_OK:
            _after(dev, "categorical encoder")
            return (z, ldj_out, cpl, noise) if want_noise else (z, ldj_out, cpl)
        eps = logistic_from_uniform(eps, mu=0.0, sigma=sigma, eps=uniform_squeeze)      # math mode 0: the two calls
        noise = eps
    if not tiled:
        _launch(dev, "cnf_encoder_forward", _ptr(categ), _ptr(eps), _ptr(table), _ptr(prior), _ptr(pad), float(beta),
                                           _ptr(ldj_in), _ptr(z), _ptr(ldj_out), _ptr(cpl), B, N, D, C, float(sigma),
                                           float(log_sigma), _ptr(flag_word(dev)), _stream(dev))
    else:
        _launch(dev, "cnf_encoder_forward_tiled", _ptr(categ), _ptr(eps), _ptr(table), _ptr(prior), _ptr(pad), float(beta),
                                                 _ptr(ldj_in), _ptr(z), _ptr(ldj_out), _ptr(cpl), _ptr(ws), B, N, D, C,
                                                 float(sigma), float(log_sigma), _ptr(flag_word(dev)), _stream(dev))
    _after(dev, "categorical encoder")
    if want_noise:
        return z, ldj_out, cpl, (noise if noise is not None else eps)
    return z, ldj_out, cpl


def encoder_forward_actconv(categ, uniform, table, category_prior, act_bias, act_scales, conv_weight, conv_sldj,
                            beta=1.0, channel_padding_mask=None, length=None, ldj=None, uniform_squeeze=1e-4,
                            sigma=LOGISTIC_SIGMA, log_sigma=LOGISTIC_LOG_SIGMA, want_class_prob=False):
    """The sampled encoder forward + the ActNorm and 1x1 convolution of the flow step behind it in ONE kernel
    (cnf_encoder_forward_actconv[_cpl]); where that kernel does not apply (class table beyond LDS, D outside
    FUSED_ACTCONV_DIMS, math mode 0) the same layers run one after the other — same bits either way.
    Returns (z after the convolution, running log-det) and, with want_class_prob, class_prob_log [B*N] as a third item."""
    dev = _dev(categ)
    if categ.dtype != torch.int64:
        categ = categ.long()
    categ = categ.contiguous()
    B, N = categ.shape
    table = _f32(table, "table")
    C, D = table.shape[0], table.shape[1] // 2
    if D in FUSED_ACTCONV_DIMS and not encoder_prefers_tiled_forward(C, D, B, N):
        u = _f32(uniform, "uniform")
        if u.numel() != B * N * D:
            raise ValueError("noise must have B*N*D entries")
        prior = _f32(category_prior, "category_prior")
        pad = _pad2d(channel_padding_mask, B, N, dev)
        b, s = _f32(act_bias.reshape(-1), "bias"), _f32(act_scales.reshape(-1), "scales")
        w, sl = _f32(conv_weight, "weight"), _f32(conv_sldj.reshape(1), "sldj")
        ln = _length(length, B, dev) if isinstance(length, torch.Tensor) else None
        ldj_in, ldj_out = _ldj_io(ldj, B, dev, inplace=False)
        z = torch.empty(B, N, D, dtype=torch.float32, device=dev)
        if want_class_prob:
            cpl = torch.empty(B * N, dtype=torch.float32, device=dev)
            status = _launch(dev, "cnf_encoder_forward_actconv_cpl",
                             _ptr(categ), _ptr(u), float(uniform_squeeze), _ptr(table), _ptr(prior), _ptr(pad), float(beta),
                             _ptr(b), _ptr(s), _ptr(w), _ptr(sl), _ptr(ln), _ptr(ldj_in), _ptr(z), _ptr(ldj_out), _ptr(cpl), B, N, D, C,
                             float(sigma), float(log_sigma), _ptr(flag_word(dev)), _stream(dev), allow_unsupported=True)
        else:
            status = _launch(dev, "cnf_encoder_forward_actconv",
                             _ptr(categ), _ptr(u), float(uniform_squeeze), _ptr(table), _ptr(prior), _ptr(pad), float(beta),
                             _ptr(b), _ptr(s), _ptr(w), _ptr(sl), _ptr(ln), _ptr(ldj_in), _ptr(z), _ptr(ldj_out), B, N, D, C,
                             float(sigma), float(log_sigma), _ptr(flag_word(dev)), _stream(dev), allow_unsupported=True)
        if status == _lib.CNF_OK:
            _after(dev, "categorical encoder + ActNorm + InvertibleConv")
            return (z, ldj_out, cpl) if want_class_prob else (z, ldj_out)
    z, ldj_enc, cpl = encoder_forward(categ, uniform, table, category_prior, beta=beta, channel_padding_mask=channel_padding_mask,
                                      sigma=sigma, log_sigma=log_sigma, uniform_squeeze=uniform_squeeze, want_class_prob=want_class_prob)
    ldj_run = ldj_enc if ldj is None else ldj + ldj_enc
    out = actnorm_invconv(z, act_bias, act_scales, conv_weight, conv_sldj, length=length,
                          channel_padding_mask=channel_padding_mask, ldj=ldj_run)
    return (out[0], out[1], cpl) if want_class_prob else out


def encoder_decode(z, table, category_prior, sigma=LOGISTIC_SIGMA, log_sigma=LOGISTIC_LOG_SIGMA, tiled=None):
    z = _f32(z, "z")
    dev = z.device
    B, N, D = z.shape
    table = _f32(table, "table")
    C = table.shape[0]
    prior = _f32(category_prior, "category_prior")
    out = torch.empty(B, N, dtype=torch.int64, device=dev)
    if encoder_fused_supported(C, D) and not tiled:
        _launch(dev, "cnf_encoder_decode", _ptr(z), _ptr(table), _ptr(prior), _ptr(out), B, N, D, C, float(sigma),
                                          float(log_sigma), _stream(dev))
    else:
        ws = torch.empty(int(_lib.load().cnf_encoder_workspace_floats(B, N, D, C)), dtype=torch.float32, device=dev)
        _launch(dev, "cnf_encoder_decode_tiled", _ptr(z), _ptr(table), _ptr(prior), _ptr(out), _ptr(ws), B, N, D, C,
                                                float(sigma), float(log_sigma), _stream(dev))
    return out


def encoder_decode_actconv(z, act_bias, act_scales, conv_weight_inv, conv_sldj, table, category_prior,
                           channel_padding_mask=None, length=None, ldj=None, sigma=LOGISTIC_SIGMA, log_sigma=LOGISTIC_LOG_SIGMA):
    """The sampling direction's last three layers — inverse 1x1 convolution, inverse ActNorm, arg-max decode — in ONE
    kernel (cnf_encoder_decode_actconv); where it does not apply the layers run one after the other, same bits.
    Returns (int64 categories [B,N], running log-det)."""
    z = _f32(z, "z")
    dev = z.device
    B, N, D = z.shape
    table = _f32(table, "table")
    C = table.shape[0]
    if D in FUSED_ACTCONV_DIMS and encoder_fused_supported(C, D):
        prior = _f32(category_prior, "category_prior")
        pad = _pad2d(channel_padding_mask, B, N, dev)
        b, s = _f32(act_bias.reshape(-1), "bias"), _f32(act_scales.reshape(-1), "scales")
        w, sl = _f32(conv_weight_inv, "weight"), _f32(conv_sldj.reshape(1), "sldj")
        ln = _length(length, B, dev) if isinstance(length, torch.Tensor) else None
        ldj_in, ldj_out = _ldj_io(ldj, B, dev, inplace=False)
        out = torch.empty(B, N, dtype=torch.int64, device=dev)
        status = _launch(dev, "cnf_encoder_decode_actconv",
                         _ptr(z), _ptr(b), _ptr(s), _ptr(w), _ptr(sl), _ptr(pad), _ptr(ln), _ptr(table), _ptr(prior), _ptr(ldj_in),
                         _ptr(out), _ptr(ldj_out), B, N, D, C, float(sigma), float(log_sigma), _ptr(flag_word(dev)), _stream(dev),
                         allow_unsupported=True)
        if status == _lib.CNF_OK:
            _after(dev, "InvertibleConv + ActNorm (reverse) + categorical decoding")
            return out, ldj_out
    zz, ldj_run = actnorm_invconv(z, act_bias, act_scales, conv_weight_inv, conv_sldj, reverse=True, length=length,
                                  channel_padding_mask=channel_padding_mask, ldj=ldj)
    return encoder_decode(zz, table, category_prior, sigma=sigma, log_sigma=log_sigma), ldj_run + torch.zeros_like(ldj_run)


def sigmoid_flow(z, reverse=False, ldj=None, alpha=1e-5):
    z = _f32(z, "z")
    dev = z.device
    B = z.shape[0]
    L = z.numel() // B
    ldj_in, ldj_out = _ldj_io(ldj, B, dev, inplace=False)
    z_out = torch.empty_like(z)
    _launch(dev, "cnf_sigmoid_flow", _ptr(z), _ptr(ldj_in), _ptr(z_out), _ptr(ldj_out), B, L, int(bool(reverse)),
                                    float(alpha), _ptr(flag_word(dev)), _stream(dev))
    _after(dev, "SigmoidFlow")
    return z_out, ldj_out


# ------------------------------------------------------------------------------------------------
class Launch:
    """A pre-bound kernel launch on fixed buffers: argument marshalling is done once, calling it only
    enqueues the kernel on the CURRENT stream (host cost ~2 us instead of ~30 us of checks and
    allocations).  The serving loop / benchmark builds these once per buffer set; they are also what a
    hipGraph capture records."""

    def __init__(self, name, fn, args, device, keep):
        self.name, self.fn, self.args, self.device = name, fn, list(args), device
        self.keep = keep          # tensors whose memory the raw pointers refer to
        self.stream_index = len(self.args) - 1

    def __call__(self):
        self.args[self.stream_index] = _stream(self.device)
        status = self.fn(*self.args)
        if status != _lib.CNF_OK:
            _lib.check(status, self.name)


def affine_coupling_launch(z, nn_out, scaling_factor, mask, z_out, ldj_out, reverse=False, ldj=None):
    z, nn_out = _f32(z, "z"), _f32(nn_out, "nn_out")
    dev = z.device
    B, N, D = z.shape
    sf = _opt_f32(scaling_factor, "scaling_factor", dev)
    m, mr, mc = _mask_desc(mask, D, dev)
    lib = _lib.load()
    args = [_ptr(z), _ptr(nn_out), _ptr(sf), _ptr(m), mr, mc, _ptr(ldj), _ptr(z_out), _ptr(ldj_out), B, N, D,
            int(bool(reverse)), _ptr(flag_word(dev)), None]
    return Launch("cnf_affine_coupling", lib.cnf_affine_coupling, args, dev, (z, nn_out, sf, m, ldj, z_out, ldj_out))


def affine_coupling_nll_launch(z, nn_out, scaling_factor, mask, z_out, ldj_out, length, neglog, nll, sums, ldj=None,
                               channel_padding_mask=None, sigma=LOGISTIC_SIGMA, log_sigma=LOGISTIC_LOG_SIGMA):
    z, nn_out = _f32(z, "z"), _f32(nn_out, "nn_out")
    dev = z.device
    B, N, D = z.shape
    sf = _opt_f32(scaling_factor, "scaling_factor", dev)
    m, mr, mc = _mask_desc(mask, D, dev)
    pad = _pad2d(channel_padding_mask, B, N, dev)
    ln = _length(length, B, dev)
    lib = _lib.load()
    args = [_ptr(z), _ptr(nn_out), _ptr(sf), _ptr(m), mr, mc, _ptr(ldj), _ptr(z_out), _ptr(ldj_out), _ptr(pad), _ptr(ln),
            _ptr(neglog), _ptr(nll), _ptr(sums), B, N, D, float(sigma), float(log_sigma), _ptr(flag_word(dev)), None]
    return Launch("cnf_affine_coupling_nll", lib.cnf_affine_coupling_nll, args, dev,
                  (z, nn_out, sf, m, ldj, z_out, ldj_out, pad, ln, neglog, nll, sums))


NLL_ACC_SLOTS = 1024          # int64 words of one accumulator (64 used, one per 128-byte line)


def affine_coupling_nll_acc_launch(z, nn_out, scaling_factor, mask, z_out, ldj_out, length, neglog, nll, acc, ldj=None,
                                   channel_padding_mask=None, sigma=LOGISTIC_SIGMA, log_sigma=LOGISTIC_LOG_SIGMA):
    """Pre-bound cnf_affine_coupling_nll_acc: the batch sum is accumulated inside the kernel into `acc` (int64 [64])."""
    z, nn_out = _f32(z, "z"), _f32(nn_out, "nn_out")
    dev = z.device
    B, N, D = z.shape
    if acc.dtype != torch.int64 or acc.numel() < NLL_ACC_SLOTS or not acc.is_contiguous():
        raise ValueError("acc must be a contiguous int64 tensor with at least %d elements" % NLL_ACC_SLOTS)
    sf = _opt_f32(scaling_factor, "scaling_factor", dev)
    m, mr, mc = _mask_desc(mask, D, dev)
    pad = _pad2d(channel_padding_mask, B, N, dev)
    ln = _length(length, B, dev)
    lib = _lib.load()
    args = [_ptr(z), _ptr(nn_out), _ptr(sf), _ptr(m), mr, mc, _ptr(ldj), _ptr(z_out), _ptr(ldj_out), _ptr(pad), _ptr(ln),
            _ptr(neglog), _ptr(nll), _ptr(acc), B, N, D, float(sigma), float(log_sigma), _ptr(flag_word(dev)), None]
    return Launch("cnf_affine_coupling_nll_acc", lib.cnf_affine_coupling_nll_acc, args, dev,
                  (z, nn_out, sf, m, ldj, z_out, ldj_out, pad, ln, neglog, nll, acc))


def nll_acc_read(acc, count, sums=None):
    """(sum, count) in fp64 from the fixed-point partial sums that cnf_affine_coupling_nll_acc accumulated."""
    dev = acc.device
    if sums is None:
        sums = torch.empty(2, dtype=torch.float64, device=dev)
    _launch(dev, "cnf_nll_acc_read", _ptr(acc), int(acc.numel()), float(count), _ptr(sums), _stream(dev))
    return sums


def nll_sum_launch(nll, sums):
    dev = nll.device
    lib = _lib.load()
    return Launch("cnf_nll_sum", lib.cnf_nll_sum, [_ptr(nll), int(nll.numel()), _ptr(sums), None], dev, (nll, sums))


def nll_sum(nll, sums=None):
    """(sum_b nll[b], B) in fp64, fixed summation order."""
    nll = _f32(nll, "nll")
    dev = nll.device
    if sums is None:
        sums = torch.empty(2, dtype=torch.float64, device=dev)
    _launch(dev, "cnf_nll_sum", _ptr(nll), int(nll.numel()), _ptr(sums), _stream(dev))
    return sums


def prior_nll_launch(z, ldj, length, neglog, nll, sums, channel_padding_mask=None,
                     sigma=LOGISTIC_SIGMA, log_sigma=LOGISTIC_LOG_SIGMA):
    z = _f32(z, "z")
    dev = z.device
    B, N, D = z.shape
    pad = _pad2d(channel_padding_mask, B, N, dev)
    ln = _length(length, B, dev)
    lib = _lib.load()
    args = [_ptr(z), _ptr(pad), _ptr(ldj), _ptr(ln), _ptr(neglog), _ptr(nll), _ptr(sums), B, N, D, float(sigma),
            float(log_sigma), None]
    return Launch("cnf_prior_nll", lib.cnf_prior_nll, args, dev, (z, pad, ldj, ln, neglog, nll, sums))


def mixture_coupling_launch(z, nn_out, mask, num_mixtures, z_out, ldj_out, scaling_factor=None,
                            mixture_scaling_factor=None, reverse=False, channel_padding_mask=None):
    z, nn_out = _f32(z, "z"), _f32(nn_out, "nn_out")
    dev = z.device
    B, N, D = z.shape
    sf = _opt_f32(scaling_factor, "scaling_factor", dev)
    msf = _opt_f32(mixture_scaling_factor, "mixture_scaling_factor", dev)
    m, mr, mc = _mask_desc(mask, D, dev)
    act, n_act = _act_list(mask, m, mr, mc, D)
    pad = _pad2d(channel_padding_mask, B, N, dev)
    lib = _lib.load()
    ws = _mixture_workspace(dev, B)
    args = [_ptr(z), _ptr(nn_out), _ptr(sf), _ptr(msf), _ptr(m), mr, mc, act, n_act, _ptr(pad), 1, 1, None,
            _ptr(z_out), _ptr(ldj_out), None, B, N, D, int(num_mixtures), int(bool(reverse)), -1.0, 1.0, 0,
            _ptr(ws), int(ws.numel()), _ptr(flag_word(dev)), None]
    name = "cnf_mixture_coupling_compact" if mixture_layout(nn_out, z, num_mixtures, act, n_act) else "cnf_mixture_coupling_ws"
    return Launch(name, getattr(lib, name), args, dev, (z, nn_out, sf, msf, m, act, pad, z_out, ldj_out, ws))


def mixture_coupling_nll_launch(z, nn_out, mask, num_mixtures, z_out, ldj_out, length, neglog, nll, acc=None,
                                scaling_factor=None, mixture_scaling_factor=None, channel_padding_mask=None,
                                sigma=LOGISTIC_SIGMA, log_sigma=LOGISTIC_LOG_SIGMA):
    z, nn_out = _f32(z, "z"), _f32(nn_out, "nn_out")
    dev = z.device
    B, N, D = z.shape
    sf = _opt_f32(scaling_factor, "scaling_factor", dev)
    msf = _opt_f32(mixture_scaling_factor, "mixture_scaling_factor", dev)
    m, mr, mc = _mask_desc(mask, D, dev)
    act, n_act = _act_list(mask, m, mr, mc, D)
    pad = _pad2d(channel_padding_mask, B, N, dev)
    ln = _length(length, B, dev)
    lib = _lib.load()
    ws = _mixture_workspace(dev, B)
    args = [_ptr(z), _ptr(nn_out), _ptr(sf), _ptr(msf), _ptr(m), mr, mc, act, n_act, _ptr(pad), 1, 1, None,
            _ptr(z_out), _ptr(ldj_out), None, _ptr(ln), _ptr(neglog), _ptr(nll), _ptr(acc),
            B, N, D, int(num_mixtures), -1.0, 1.0, 0, float(sigma), float(log_sigma),
            _ptr(ws), int(ws.numel()), _ptr(flag_word(dev)), None]
    name = "cnf_mixture_coupling_compact_nll" if mixture_layout(nn_out, z, num_mixtures, act, n_act) else "cnf_mixture_coupling_nll"
    return Launch(name, getattr(lib, name), args, dev, (z, nn_out, sf, msf, m, act, pad, ln, z_out, ldj_out, neglog, nll, acc, ws))
