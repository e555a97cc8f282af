"""Invertible 1x1 convolution (z' = x @ W) on the HIP kernel cnf_invconv.

Interface of layers/flows/permutation_layers.py: LU-parametrised (buffers p, sign_s, l_mask, eye;
parameters l, log_s, u) or dense (`weight`), eval-mode cache of (W, W^-1, sldj) per device string
(:56-103).  The [B,N,D] x [D,D] product, the padding mask and the log-det update are the kernel; building W
from (P, L, U) is D x D parameter preparation — one launch of its own on the device (cnf_invconv_lu_weight: the
reference's chain of ~11 tiny ops and their autograd were 26 launches per flow step of a training pass), the
reference's tensor expression for CPU tensors and D > 16."""
from collections import defaultdict

import numpy as np
import scipy.linalg
import torch
import torch.nn as nn

from ... import functional as Fn
from ... import ops
from .flow_layer import FlowLayer


class InvertibleConv(FlowLayer):

    def __init__(self, c_in, LU_decomposed=True):
        super().__init__()
        self.num_channels = c_in
        self.LU_decomposed = LU_decomposed
        if c_in == 2:
            # a random 2x2 orthogonal matrix is often close to the identity: start from a rotation
            # by 45..135 or 225..315 degrees instead (:19-30)
            r = np.random.uniform()
            angle = (0.25 + 0.5 * (2 * r)) * np.pi if r < 0.5 else (1.25 + 0.5 * (2 * r - 1)) * np.pi
            w_init = np.array([[np.cos(angle), -np.sin(angle)], [np.sin(angle), np.cos(angle)]])
        else:
            w_init = np.linalg.qr(np.random.randn(c_in, c_in))[0].astype(np.float32)
        if not LU_decomposed:
            self.weight = nn.Parameter(torch.from_numpy(w_init.astype(np.float32)), requires_grad=True)
        else:
            np_p, np_l, np_u = scipy.linalg.lu(w_init)
            diag = np.diag(np_u)
            self.register_buffer('p', torch.Tensor(np_p.astype(np.float32)))
            self.register_buffer('sign_s', torch.Tensor(np.sign(diag).astype(np.float32)))
            self.l = nn.Parameter(torch.Tensor(np_l.astype(np.float32)), requires_grad=True)
            self.log_s = nn.Parameter(torch.Tensor(np.log(np.abs(diag)).astype(np.float32)), requires_grad=True)
            self.u = nn.Parameter(torch.Tensor(np.triu(np_u, k=1).astype(np.float32)), requires_grad=True)
            self.register_buffer('l_mask', torch.Tensor(np.tril(np.ones(w_init.shape, dtype=np.float32), -1)))
            self.register_buffer('eye', torch.Tensor(np.eye(*w_init.shape, dtype=np.float32)))
        self.eval_dict = defaultdict(lambda: self._get_default_inner_dict())

    def _get_default_inner_dict(self):
        return {"weight": None, "inv_weight": None, "sldj": None}

    def _build_weight(self):
        if not self.LU_decomposed:
            return self.weight, torch.slogdet(self.weight)[1]
        if self.l.is_cuda and self.num_channels <= ops.LU_WEIGHT_MAX_D and ops.FUSE_LU_WEIGHT:
            # the same matrix and log-det in one launch (and one for the backward): cnf_invconv_lu_weight
            weight, sldj, inverse = Fn.LUWeightFn.apply(self.l, self.u, self.log_s, self.p, self.sign_s)
            if inverse.numel():
                weight._cnf_inverse = inverse             # picked up by the fused groups' backward (functional._known_inverse)
            return weight, sldj
        lower = self.l * self.l_mask + self.eye
        upper = self.u * self.l_mask.transpose(0, 1).contiguous() + torch.diag(self.sign_s * torch.exp(self.log_s))
        return torch.matmul(self.p, torch.matmul(lower, upper)), self.log_s.sum()

    def _get_weight(self, device_name, inverse=False):
        """Train mode: rebuild every call and drop the cache; eval mode: cache per device (:61-89)."""
        if self.training:
            weight, sldj = self._build_weight()
            if not self._is_eval_dict_empty(device_name):
                self._empty_eval_dict(device_name)
            if inverse:
                weight = torch.inverse(weight.double()).float()
            return weight, sldj
        if self._is_eval_dict_empty(device_name):
            weight, sldj = self._build_weight()
            entry = self.eval_dict[device_name]
            entry["weight"] = weight.detach()
            entry["sldj"] = sldj.detach()
            entry["inv_weight"] = torch.inverse(weight.double()).float().detach()
        entry = self.eval_dict[device_name]
        return (entry["inv_weight"] if inverse else entry["weight"]), entry["sldj"]

    def _is_eval_dict_empty(self, device_name=None):
        if device_name is not None:
            return not (device_name in self.eval_dict)
        return len(self.eval_dict) == 0

    def _empty_eval_dict(self, device_name=None):
        if device_name is not None:
            self.eval_dict.pop(device_name)
        else:
            self.eval_dict = defaultdict(lambda: self._get_default_inner_dict())

    def forward(self, x, ldj=None, reverse=False, length=None, channel_padding_mask=None,
                layer_share_dict=None, **kwargs):
        weight, sldj = self._get_weight(device_name=str(x.device), inverse=reverse)
        if Fn.needs_grad(x, weight, sldj, ldj):
            z, ldj_out = Fn.InvConvFn.apply(x, weight, sldj, ldj, length, channel_padding_mask, reverse)
        else:
            z, ldj_out = ops.invconv(x, weight, sldj, reverse=reverse, length=length,
                                     channel_padding_mask=channel_padding_mask, ldj=ldj)
        if layer_share_dict is not None:
            layer_share_dict["t"] = layer_share_dict["t"] * 0.0
            layer_share_dict["log_s"] = layer_share_dict["log_s"] * 0.0
            if "error_decay" in layer_share_dict:
                layer_share_dict["error_decay"] = layer_share_dict["error_decay"] * 0.0
        return z, ldj_out

    def info(self):
        return "Invertible 1x1 Convolution - %i channels %s" % (self.num_channels, "(LU decomposed)" if self.LU_decomposed else "")
