"""Affine coupling layer on the fused HIP kernel (cnf_affine_coupling).

Same constructor, buffers (`mask`), parameters (`scaling_factor`, `nn.*`), statics and info() as
layers/flows/coupling_layer.py of the reference; the eight eager ops after the subnet
(:53-63) are one kernel launch."""
import math

import torch
import torch.nn as nn

from ... import functional as Fn
from ... import ops
from ..networks.help_layers import run_sequential_with_mask
from .flow_layer import FlowLayer


class CouplingLayer(FlowLayer):

    def __init__(self, c_in, mask, model_func, block_type=None, c_out=-1, **kwargs):
        super().__init__()
        self.c_in = c_in
        self.c_out = c_out if c_out > 0 else 2 * c_in
        self.register_buffer('mask', mask)
        self.block_type = block_type
        self.scaling_factor = nn.Parameter(torch.zeros(c_in))
        self.nn = model_func(c_out=self.c_out)

    def run_network(self, x, length=None, **kwargs):
        """coupling_layer.py:28-39 — the subnet stays PyTorch (GEMMs)."""
        if isinstance(self.nn, nn.Sequential):
            out = run_sequential_with_mask(self.nn, x, length=length, **kwargs)
        else:
            out = self.nn(x, length=length, **kwargs)
        if kwargs.get("channel_padding_mask", None) is not None:
            out = out * kwargs["channel_padding_mask"]
        return out

    def _prepare_mask(self, mask, z):
        """coupling_layer.py:67-74."""
        m = self.mask.unsqueeze(dim=0) if len(z.shape) > len(self.mask.shape) else self.mask
        if 1 < m.size(1) < z.size(1):
            m = m.repeat(1, int(math.ceil(z.size(1) / m.size(1))), 1).contiguous()
        if m.size(1) > z.size(1):
            m = m[:, :z.size(1)]
        return m

    def forward(self, z, ldj=None, reverse=False, channel_padding_mask=None, **kwargs):
        # NB: like the reference, padding is ignored by the affine coupling
        nn_out = self.run_network(x=z * self._prepare_mask(self.mask, z), **kwargs)
        if Fn.needs_grad(z, nn_out, self.scaling_factor, ldj):
            return Fn.AffineCouplingFn.apply(z, nn_out, self.scaling_factor, ldj, self.mask, reverse)
        return ops.affine_coupling(z, nn_out, self.scaling_factor, self.mask, reverse=reverse, ldj=ldj)

    @staticmethod
    def get_coup_params(nn_out, mask, scaling_factor=None):
        """coupling_layer.py:76-86 — materialised (s, t)."""
        if Fn.needs_grad(nn_out, scaling_factor):
            return Fn.AffineParamsFn.apply(nn_out, scaling_factor, mask)
        return ops.affine_params(nn_out, mask, scaling_factor)

    @staticmethod
    def run_with_params(orig_z, s, t, reverse=False):
        """coupling_layer.py:88-98."""
        if Fn.needs_grad(orig_z, s, t):
            return Fn.AffineTransformFn.apply(orig_z, s, t, reverse)
        return ops.affine_transform(orig_z, s, t, reverse=reverse)

    @staticmethod
    def create_channel_mask(c_in, ratio=0.5, mask_floor=True):
        """First floor/ceil(c_in*ratio) channels are 1 = fed to the subnet, left unchanged."""
        kept = int(math.floor(c_in * ratio)) if mask_floor else int(math.ceil(c_in * ratio))
        mask = torch.zeros(1, c_in)
        mask[0, :kept] = 1.0
        return mask

    @staticmethod
    def create_chess_mask(seq_len=2):
        assert seq_len > 1
        zeros = int(seq_len // 2)
        ones = seq_len - zeros
        return torch.cat([torch.ones(ones, 1), torch.zeros(zeros, 1)], dim=1).view(-1, 1)

    def info(self):
        is_channel_mask = (self.mask.size(0) == 1)
        s = "Coupling Layer - Input size %i" % (self.c_in)
        if self.block_type is not None:
            s += ", block type %s" % (self.block_type)
        s += ", mask ratio %.2f, %s mask" % ((1 - self.mask).mean().item(), "channel" if is_channel_mask else "chess")
        return s
