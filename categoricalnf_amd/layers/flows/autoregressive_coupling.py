"""Mixture-CDF transform of ALL channels with parameters from an autoregressive subnet.

Interface of layers/flows/autoregressive_coupling.py:9-47 (forward only; reverse raises, as there)."""
import torch
import torch.nn as nn

from ... import functional as Fn
from ... import ops
from .flow_layer import FlowLayer


class AutoregressiveMixtureCDFCoupling(FlowLayer):

    def __init__(self, c_in, model_func, block_type=None, num_mixtures=10):
        super().__init__()
        self.c_in = c_in
        self.num_mixtures = num_mixtures
        self.block_type = block_type
        self.scaling_factor = nn.Parameter(torch.zeros(self.c_in))
        self.mixture_scaling_factor = nn.Parameter(torch.zeros(self.c_in, self.num_mixtures))
        self.nn = model_func(c_out=c_in * (2 + 3 * self.num_mixtures))

    def forward(self, z, ldj=None, reverse=False, **kwargs):
        if reverse:
            raise NotImplementedError
        nn_out = self.nn(x=z, **kwargs)
        if Fn.needs_grad(z, nn_out, self.scaling_factor, self.mixture_scaling_factor, ldj):
            z_out, ldj_out, _ = Fn.MixtureCouplingFn.apply(
                z, nn_out, self.scaling_factor, self.mixture_scaling_factor, ldj, None, kwargs.get("channel_padding_mask", None),
                self.num_mixtures, -1, 1, self.training, False, True)
            return z_out, ldj_out
        # no mask and no padding inside the transform; the output is multiplied by the padding mask
        # afterwards (autoregressive_coupling.py:38-45)
        z_out, ldj_out, _ = ops.mixture_coupling(
            z, nn_out, None, self.num_mixtures, self.scaling_factor, self.mixture_scaling_factor, reverse=False,
            channel_padding_mask=kwargs.get("channel_padding_mask", None), pad_in_transform=False, pad_output=True,
            ldj=ldj, want_reg=False)
        return z_out, ldj_out

    def info(self):
        s = "Autoregressive Mixture CDF Coupling Layer - Input size %i" % (self.c_in)
        if self.block_type is not None:
            s += ", block type %s" % (self.block_type)
        return s
