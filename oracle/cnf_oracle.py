"""CPU oracle for the CategoricalNF coupling hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``categoricalnf_amd/`` imports this file; only
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may.

It restates, as plain functions over CPU tensors, the arithmetic of the reference's
eager-PyTorch layers (phlippe/CategoricalNF, mounted at /root/reference when the fixtures were
generated).  The same op order and the same dtypes (fp32 everywhere, fp64 inside the
mixture-CDF transform and inside the logit of the noise sampler) are used so that the outputs are
equal to the reference's on the same inputs.

Parity pin: ``oracle/gen_golden.py`` imported the real reference in the build container, ran it on
seeded inputs and wrote inputs + reference outputs into ``tests/golden/*.npz``.
``tests/test_oracle_golden.py`` checks every function here against those files, so the oracle is
PINNED (not "parity unpinned").  The reference ships no golden files or tests of its own
(SURVEY.md §4); its only known-answer checks are the ``__main__`` demos, which the fixtures
replay (mixture round trip, dequantisation round trip, LU-vs-dense weight).

Each function cites the reference lines (relative to /root/reference) it follows.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

LOGISTIC_SIGMA = 1.0 / 1.81          # layers/flows/distributions.py:95
LOGISTIC_LOG_SIGMA = float(np.log(LOGISTIC_SIGMA))   # distributions.py:99


# --------------------------------------------------------------------------------------------
# masks  (layers/flows/coupling_layer.py:67-74, 101-121 ; general/mutils.py:273-288)
# --------------------------------------------------------------------------------------------

def channel_mask(c_in, ratio=0.5, mask_floor=True):
    """coupling_layer.py:101-112 — first floor(c_in*ratio) channels are 1 (= fed to the subnet)."""
    kept = int(math.floor(c_in * ratio)) if mask_floor else int(math.ceil(c_in * ratio))
    m = torch.zeros(1, c_in)
    m[0, :kept] = 1.0
    return m


def chess_mask(seq_len=2):
    """coupling_layer.py:115-121 — [[1],[0]] for seq_len=2 (alternates along the sequence axis)."""
    assert seq_len > 1
    zeros = seq_len // 2
    ones = seq_len - zeros
    return torch.cat([torch.ones(ones, 1), torch.zeros(zeros, 1)], dim=1).view(-1, 1)


def expand_mask(mask, z):
    """coupling_layer.py:67-74 — broadcastable mask for a [B,N,D] tensor (tile/truncate along N)."""
    m = mask.unsqueeze(0) if z.dim() > mask.dim() else mask
    if 1 < m.size(1) < z.size(1):
        m = m.repeat(1, int(math.ceil(z.size(1) / m.size(1))), 1).contiguous()
    if m.size(1) > z.size(1):
        m = m[:, :z.size(1)]
    return m


def length_mask(length, max_len=None):
    """general/mutils.py:273-288 — [B,N,1] fp32 padding mask from integer lengths."""
    if max_len is None:
        max_len = int(length.max())
    m = (torch.arange(max_len).view(1, max_len) < length.unsqueeze(-1)).to(torch.float32)
    return m.unsqueeze(-1)


# --------------------------------------------------------------------------------------------
# affine coupling  (layers/flows/coupling_layer.py:42-98)
# --------------------------------------------------------------------------------------------

def affine_params(nn_out, mask, scaling_factor=None):
    """coupling_layer.py:76-86 (and :53-60) — interleaved (s,t) split, tanh bound, masking."""
    pairs = nn_out.view(nn_out.shape[:-1] + (nn_out.shape[-1] // 2, 2))
    s, t = pairs[..., 0], pairs[..., 1]
    if scaling_factor is not None:
        fac = scaling_factor.exp().view(1, 1, -1)
        s = torch.tanh(s / fac.clamp(min=1.0)) * fac
    s = s * (1 - mask)
    t = t * (1 - mask)
    return s, t


def affine_transform(z, s, t, reverse=False):
    """coupling_layer.py:88-98."""
    if not reverse:
        return (z + t) * torch.exp(s), s.sum(dim=[1, 2])
    return z * torch.exp(-1 * s) - t, -s.sum(dim=[1, 2])


def affine_coupling(z, nn_out, mask, scaling_factor, reverse=False, ldj=None):
    """coupling_layer.py:42-65 with the subnet output injected."""
    if ldj is None:
        ldj = z.new_zeros(z.size(0))
    m = expand_mask(mask, z)
    s, t = affine_params(nn_out, m, scaling_factor)
    z_out, layer_ldj = affine_transform(z, s, t, reverse=reverse)
    return z_out, ldj + layer_ldj


# --------------------------------------------------------------------------------------------
# logistic-mixture CDF coupling  (layers/flows/mixture_cdf_layer.py)
# --------------------------------------------------------------------------------------------

def _safe_log(x):
    """mixture_cdf_layer.py:197-198."""
    return torch.log(x.clamp(min=1e-22))


def mixture_params(nn_out, mask, num_mixtures, scaling_factor=None, mixture_scaling_factor=None):
    """mixture_cdf_layer.py:145-180 — channel-major blocks [t, log_s, log_pi[K], mu[K], ls[K]]."""
    K = num_mixtures
    P = 2 + 3 * K
    blk = nn_out.reshape(nn_out.shape[:-1] + (nn_out.shape[-1] // P, P))
    t, log_s = blk[..., 0], blk[..., 1]
    log_pi, mixt_t, mixt_log_s = blk[..., 2:2 + K], blk[..., 2 + K:2 + 2 * K], blk[..., 2 + 2 * K:2 + 3 * K]
    if scaling_factor is not None:
        fac = scaling_factor.exp().view(*((1,) * (log_s.dim() - 1) + tuple(scaling_factor.shape)))
        log_s = torch.tanh(log_s / fac.clamp(min=1.0)) * fac
    if mixture_scaling_factor is not None:
        mfac = mixture_scaling_factor.exp().view(
            *((1,) * (mixt_log_s.dim() - 2) + tuple(mixture_scaling_factor.shape)))
        mixt_log_s = torch.tanh(mixt_log_s / mfac.clamp(min=1.0)) * mfac
    if mask is not None:
        keep = 1 - mask
        t, log_s = t * keep, log_s * keep
        keep_k = 1 - mask.unsqueeze(-1)
        log_pi, mixt_t, mixt_log_s = log_pi * keep_k, mixt_t * keep_k, mixt_log_s * keep_k
    return t.double(), log_s.double(), log_pi.double(), mixt_t.double(), mixt_log_s.double()


def _mix_log_cdf(x, log_pi, mu, ls):
    """mixture_cdf_layer.py:209-214, 226-232."""
    zk = (x.unsqueeze(-1) - mu) * torch.exp(-ls)
    return torch.logsumexp(F.log_softmax(log_pi, dim=-1) + F.logsigmoid(zk), dim=-1)


def _mix_log_pdf(x, log_pi, mu, ls):
    """mixture_cdf_layer.py:201-206, 217-223."""
    zk = (x.unsqueeze(-1) - mu) * torch.exp(-ls)
    return torch.logsumexp(F.log_softmax(log_pi, dim=-1) + (zk - ls - 2 * F.softplus(zk)), dim=-1)


def mixture_inv_cdf(y, log_pi, mu, ls, eps=1e-10, max_iters=100):
    """mixture_cdf_layer.py:235-264 — bisection from x=0 with a tensor-wide stopping test."""
    if y.min() <= 0 or y.max() >= 1:
        raise RuntimeError('Inverse logisitic CDF got y outside (0, 1)')
    x = torch.zeros_like(y)
    spread = torch.sum(torch.exp(ls), dim=-1, keepdim=True)
    lb = (mu - 20 * spread).min(dim=-1)[0]
    ub = (mu + 20 * spread).max(dim=-1)[0]
    diff, it = float('inf'), 0
    while diff > eps and it < max_iters:
        cur = torch.exp(_mix_log_cdf(x, log_pi, mu, ls))
        gt = (cur > y).type(y.dtype)
        lt = 1 - gt
        new_x = gt * (x + lb) / 2. + lt * (x + ub) / 2.
        lb = gt * lb + lt * x
        ub = gt * x + lt * ub
        diff = (new_x - x).abs().max()
        x = new_x
        it += 1
    return x


def mixture_transform(z64, t, log_s, log_pi, mu, ls, reverse=False, reg_max=-1, reg_factor=1,
                      mask=None, channel_padding_mask=None, is_training=True):
    """mixture_cdf_layer.py:95-142.  Returns (z_out fp64, ldj[B] fp64, reg_ldj[B,N,D] or None)."""
    change = 1 - mask if mask is not None else torch.ones_like(z64)
    if channel_padding_mask is not None:
        change = change * channel_padding_mask
    reg = None
    if not reverse:
        u = _mix_log_cdf(z64, log_pi, mu, ls).exp()
        if reg_max > 0 and is_training:
            reg = torch.stack([_safe_log(u), _safe_log(1 - u)], dim=-1) / np.log(10)
            reg = (reg.clamp(max=-reg_max) + reg_max).sum(dim=-1) * change
        else:
            reg = torch.zeros_like(u)
        y = -_safe_log(u.reciprocal() - 1.)                       # :267-276 inverse()
        mixt_ldj = -_safe_log(u) - _safe_log(1. - u)
        out = (y + t) * log_s.exp()
        logistic_ldj = _mix_log_pdf(z64, log_pi, mu, ls)
        ldj = (change * (log_s + mixt_ldj + logistic_ldj + reg * reg_factor)).sum(dim=[1, 2])
    else:
        v = z64 * (-log_s).exp() - t
        u = torch.sigmoid(v)
        mixt_ldj = F.softplus(v) + F.softplus(-v)
        u = u.clamp(1e-5, 1. - 1e-5)
        out = mixture_inv_cdf(u, log_pi, mu, ls)
        logistic_ldj = _mix_log_pdf(out, log_pi, mu, ls)
        ldj = -(change * (log_s + mixt_ldj + logistic_ldj)).sum(dim=[1, 2])
    if mask is not None:
        out = out * change + z64 * (1 - change)
    return out, ldj, reg


def mixture_coupling(z, nn_out, mask, num_mixtures, scaling_factor, mixture_scaling_factor,
                     reverse=False, channel_padding_mask=None, reg_max=-1, reg_factor=1,
                     is_training=True):
    """mixture_cdf_layer.py:45-92 with the subnet output injected.

    Returns (z_out fp32, ldj[B] fp32, reg_sum[B] fp32).  ``mask`` may be None (the
    autoregressive variant, autoregressive_coupling.py:25-47, which also passes no padding mask
    into the transform and multiplies the output by it afterwards)."""
    m = expand_mask(mask, z) if mask is not None else None
    if channel_padding_mask is None and mask is not None:
        channel_padding_mask = torch.ones_like(z)
    p = mixture_params(nn_out, m, num_mixtures, scaling_factor, mixture_scaling_factor)
    out, ldj, reg = mixture_transform(z.double(), *p, reverse=reverse, reg_max=reg_max,
                                      reg_factor=reg_factor, mask=m,
                                      channel_padding_mask=channel_padding_mask if mask is not None else None,
                                      is_training=is_training)
    out = out.float()
    if channel_padding_mask is not None:
        out = out * channel_padding_mask
    reg_sum = reg.float().sum(dim=[1, 2]) if reg is not None else None
    return out, ldj.float(), reg_sum


# --------------------------------------------------------------------------------------------
# ActNorm / ExtActNorm  (layers/flows/activation_normalization.py)
# --------------------------------------------------------------------------------------------

def actnorm(z, bias, scales, reverse=False, length=None, channel_padding_mask=None, ldj=None):
    """activation_normalization.py:24-48.  bias/scales are [1,1,D].  Adds into ldj."""
    if ldj is None:
        ldj = z.new_zeros(z.size(0))
    else:
        ldj = ldj.clone()
    if length is None:
        length = z.size(1) if channel_padding_mask is None else channel_padding_mask.squeeze(2).sum(1)
    else:
        length = length.float()
    if not reverse:
        z = (z + bias) * torch.exp(scales)
        ldj += scales.sum(dim=[1, 2]) * length
    else:
        z = z * torch.exp(-scales) - bias
        ldj += (-scales.sum(dim=[1, 2])) * length
    if channel_padding_mask is not None:
        z = z * channel_padding_mask
    return z, ldj


def actnorm_data_init(x, channel_padding_mask=None):
    """activation_normalization.py:55-67 — bias = -mean, scales = -0.5*log(var) over (0,1)."""
    m = channel_padding_mask if channel_padding_mask is not None else x.new_ones(x.shape)
    cnt = m.sum(dim=[0, 1], keepdims=True)
    bias = -(x * m).sum(dim=[0, 1], keepdims=True) / cnt
    var = (((x + bias) ** 2) * m).sum(dim=[0, 1], keepdims=True) / cnt
    return bias, -0.5 * var.log()


def ext_actnorm(z, nn_out, reverse=False, channel_padding_mask=None, ldj=None):
    """activation_normalization.py:116-144 with the predictor output nn_out [B,N,2D] injected."""
    ldj = z.new_zeros(z.size(0)) if ldj is None else ldj.clone()
    pad = 1.0 if channel_padding_mask is None else channel_padding_mask
    bias, scales = nn_out.chunk(2, dim=2)
    scales = torch.tanh(scales)
    if not reverse:
        z = (z + bias) * torch.exp(scales)
        ldj += (scales * pad).sum(dim=[1, 2])
    else:
        z = z * torch.exp(-scales) - bias
        ldj += -(scales * pad).sum(dim=[1, 2])
    return z, ldj


# --------------------------------------------------------------------------------------------
# invertible 1x1 convolution  (layers/flows/permutation_layers.py)
# --------------------------------------------------------------------------------------------

def invconv_weight_lu(p, l, u, log_s, sign_s):
    """permutation_layers.py:61-71 — W = P (L∘tril + I)(U∘triu + diag(sign·e^log_s)), sldj = Σ log_s."""
    D = l.size(0)
    l_mask = torch.tril(torch.ones(D, D), -1)
    eye = torch.eye(D)
    lo = l * l_mask + eye
    up = u * l_mask.transpose(0, 1).contiguous() + torch.diag(sign_s * torch.exp(log_s))
    return torch.matmul(p, torch.matmul(lo, up)), log_s.sum()


def invconv(x, weight, sldj, reverse=False, length=None, channel_padding_mask=None, ldj=None):
    """permutation_layers.py:106-136 — z = x @ W (right multiply); reverse uses inverse(W.double())."""
    if ldj is None:
        ldj = x.new_zeros(x.size(0))
    length = x.size(1) if length is None else length.float()
    w = torch.inverse(weight.double()).float() if reverse else weight
    s = sldj * length
    ldj = ldj - s if reverse else ldj + s
    z = torch.matmul(x, w.unsqueeze(0))
    if channel_padding_mask is not None:
        z = z * channel_padding_mask
    return z, ldj


# --------------------------------------------------------------------------------------------
# logistic prior  (layers/flows/distributions.py:91-185) and NLL assembly
# --------------------------------------------------------------------------------------------

def logistic_log_prob(x, mu=0.0, sigma=LOGISTIC_SIGMA, log_sigma=LOGISTIC_LOG_SIGMA):
    """distributions.py:129-136, 154-163."""
    v = (x - mu) / sigma
    return -(F.softplus(v) + F.softplus(-v) + log_sigma)


def logistic_from_uniform(u, mu=0.0, sigma=LOGISTIC_SIGMA, eps=1e-4):
    """distributions.py:139-145, 117-127 — squeeze, logit in fp64, cast, scale."""
    u = (u * (1 - eps)) + eps / 2
    u = u.double()
    x = -torch.log(u.reciprocal() - 1.)
    return x.float() * sigma + mu


def nll_per_sample(z, ldj, length, channel_padding_mask=None):
    """experiments/set_modeling/task.py:96-118 — (−Σ logp·mask − ldj)/length per sample."""
    lp = logistic_log_prob(z)
    if channel_padding_mask is not None:
        lp = lp * channel_padding_mask
    neglog = -lp.sum(dim=[1, 2])
    return (-ldj) / length.float() + neglog / length.float()


def bits_per_dim(nll):
    """general/task.py:148-149."""
    return float(np.log2(np.exp(1)) * nll)


# --------------------------------------------------------------------------------------------
# mixture-model categorical encoder  (layers/categorical_encoding/linear_encoding.py)
# --------------------------------------------------------------------------------------------

def encoder_forward(categ, eps, table, category_prior_log, beta=1.0, channel_padding_mask=None):
    """linear_encoding.py:59-106,120-133 + 153-174 for the mixture-model flow (one ExtActNorm).

    categ int64 [B,N]; eps fp32 [B*N,1,D] logistic noise; table fp32 [C,2D] = pred_net(embed(c))
    (activation_normalization.py:127-129: first D = bias, last D = pre-tanh scales).
    Returns (z [B,N,D], ldj[B], class_prob_log [B*N])."""
    B, N = categ.shape
    T, C, D = B * N, table.size(0), eps.size(-1)
    c = categ.reshape(T)
    pad = (channel_padding_mask.reshape(T, 1, 1) if channel_padding_mask is not None
           else eps.new_ones(T, 1, 1))
    init_log_p = logistic_log_prob(eps).sum(dim=[1, 2])
    z, ldj_f = ext_actnorm(eps, table[c].view(T, 1, 2 * D))
    log_point = init_log_p - ldj_f + torch.take(category_prior_log, c)
    # all-class reverse flows (linear_encoding.py:153-174)
    z_rep = z.expand(-1, C, -1).reshape(T * C, 1, D)
    cls = torch.arange(C).view(1, C).expand(T, -1).reshape(T * C)
    z_back, ldj_b = ext_actnorm(z_rep, table[cls].view(T * C, 1, 2 * D), reverse=True)
    back = logistic_log_prob(z_back).sum(dim=[1, 2]) + ldj_b
    denom = back.view(T, C) + category_prior_log[None, :]
    onehot = F.one_hot(c, C).to(torch.float32)
    denom = denom * (1 - onehot) + log_point.unsqueeze(-1) * onehot
    class_prob_log = log_point - torch.logsumexp(denom, dim=-1)
    ldj_tok = (beta * class_prob_log - (init_log_p - ldj_f)) * pad.squeeze()
    z = z * pad
    return z.reshape(B, N, D), ldj_tok.reshape(B, N).sum(dim=-1), class_prob_log


def encoder_decode(z, table, category_prior_log):
    """linear_encoding.py:108-118,184-196 — argmax over class-conditional reverse-flow log-probs."""
    B, N, D = z.shape
    T, C = B * N, table.size(0)
    z_rep = z.reshape(T, 1, D).expand(-1, C, -1).reshape(T * C, 1, D)
    cls = torch.arange(C).view(1, C).expand(T, -1).reshape(T * C)
    z_back, ldj_b = ext_actnorm(z_rep, table[cls].view(T * C, 1, 2 * D), reverse=True)
    score = (logistic_log_prob(z_back).sum(dim=[1, 2]) + ldj_b).view(T, C) + category_prior_log[None, :]
    return score.argmax(dim=-1).reshape(B, N), score


# --------------------------------------------------------------------------------------------
# sigmoid / logit flow  (layers/flows/sigmoid_layer.py:24-47)
# --------------------------------------------------------------------------------------------

def sigmoid_flow(z, reverse=False, ldj=None, alpha=1e-5):
    """sigmoid_layer.py:24-47 (reverse already XOR-ed by the caller)."""
    if ldj is None:
        ldj = z.new_zeros(z.size(0))
    if not reverse:
        layer = -z - 2 * F.softplus(-z)
        z = torch.sigmoid(z)
    else:
        z = z * (1 - alpha) + alpha * 0.5
        layer = (-torch.log(z) - torch.log(1 - z) + math.log(1 - alpha))
        z = torch.log(z) - torch.log(1 - z)
    return z, ldj + layer.view(z.size(0), -1).sum(dim=1)
