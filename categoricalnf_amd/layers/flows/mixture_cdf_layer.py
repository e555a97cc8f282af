"""Logistic-mixture CDF coupling (Flow++ style) on the fp64 HIP kernels.

Interface of layers/flows/mixture_cdf_layer.py: constructor (:13-42), forward (:45-92) returning the
3-tuple (z, ldj, {"ldj", "regularizer_ldj"}), and the two statics that other layers call
(get_mixt_params :145-180, run_with_params :95-142)."""
import os
import warnings

import torch
import torch.nn as nn
import torch.nn.functional as F

from ... import functional as Fn
from ... import ops
from .coupling_layer import CouplingLayer

COMPACT_PARAMS = os.environ.get("CNF_COMPACT_PARAMS", "0") == "1"      # default of MixtureCDFCoupling(compact_params=...)


class RowSlicedLinear(nn.Linear):
    """nn.Linear that can apply a contiguous range of its output rows only.  The last Linear of a coupling sub-network is re-classed
    to this (same parameters, same state_dict keys, `isinstance(m, nn.Linear)` still holds); with `rows = (a, b)` its GEMM runs on
    `weight[a:b]` / `bias[a:b]` — views of the parameters, so a reference checkpoint loads unchanged and the optimiser sees the same
    tensors (autograd leaves zero rows in their gradients where the reference's masked blocks give zeros)."""
    rows = None

    def forward(self, x):
        if self.rows is None:
            return F.linear(x, self.weight, self.bias)
        a, b = self.rows
        return F.linear(x, self.weight[a:b], None if self.bias is None else self.bias[a:b])


class MixtureCDFCoupling(CouplingLayer):
    """compact_params (beyond the reference's signature; default from CNF_COMPACT_PARAMS=1): with a channel mask the reference's
    sub-network computes parameter blocks for ALL c_in channels and get_mixt_params multiplies those of the untransformed channels by
    zero (mixture_cdf_layer.py:65-78,163-171).  With compact_params the sub-network's last Linear applies only the rows of the
    transformed channels (RowSlicedLinear) and the kernels take the compact [B, N, n_act * (2 + 3K)] tensor
    (cnf_mixture_coupling_compact*): half the GEMM, half the HBM traffic of forward / inverse, no zero blocks in the backward.  Falls
    back to the reference layout by itself when the mask is not a channel mask, when no final Linear of c_out rows is found, or when the
    sub-network's output does not turn out to be that Linear's (checked once, on the first call, against the full layout)."""

    def __init__(self, c_in, mask, model_func, block_type=None, num_mixtures=10,
                 regularizer_max=-1, regularizer_factor=1, compact_params=None, **kwargs):
        super().__init__(c_in=c_in, mask=mask, model_func=model_func, block_type=block_type,
                         c_out=c_in * (2 + num_mixtures * 3), **kwargs)
        self.num_mixtures = num_mixtures
        self.mixture_scaling_factor = nn.Parameter(torch.zeros(self.c_in, self.num_mixtures))
        self.regularizer_max = regularizer_max
        self.regularizer_factor = regularizer_factor
        self._compact_linear = None
        self._compact_rows = None
        self._compact_checked = False
        if COMPACT_PARAMS if compact_params is None else compact_params:
            self.enable_compact_params()

    def enable_compact_params(self):
        """Switch the layer to the compact parameter layout if its mask and sub-network allow it; returns whether it did."""
        m = self.mask.detach().reshape(-1, self.mask.shape[-1]).cpu()
        P = 2 + 3 * self.num_mixtures
        if m.shape[0] != 1 or m.shape[1] != self.c_in:
            return False                                        # chess masks transform every channel somewhere
        act = [i for i, v in enumerate(m[0].tolist()) if v == 0.0]
        if not act or len(act) == self.c_in or act != list(range(act[0], act[0] + len(act))):
            return False
        last = None
        for mod in self.nn.modules():
            if isinstance(mod, nn.Linear) and mod.out_features == self.c_out:
                last = mod
        if last is None or type(last) not in (nn.Linear, RowSlicedLinear):
            return False
        last.__class__ = RowSlicedLinear
        object.__setattr__(self, "_compact_linear", last)       # a plain reference: not a second registration (state_dict keys stay the reference's)
        self._compact_rows = (act[0] * P, (act[0] + len(act)) * P)
        self._compact_checked = False
        return True

    def disable_compact_params(self):
        if self._compact_linear is not None:
            self._compact_linear.rows = None
        self._compact_linear = None
        self._compact_rows = None

    def run_network(self, x, length=None, **kwargs):
        if self._compact_linear is None:
            return super().run_network(x, length=length, **kwargs)
        lin = self._compact_linear
        if not self._compact_checked:
            # once: the compact output must be the full output's transformed blocks (a sub-network that does anything position-
            # dependent to its last Linear's columns would break that) — otherwise the reference layout stays
            with torch.no_grad():
                full = super().run_network(x, length=length, **kwargs)
                lin.rows = self._compact_rows
                try:
                    comp = super().run_network(x, length=length, **kwargs)
                finally:
                    lin.rows = None
                a, b = self._compact_rows
                ok = (comp.shape[:-1] == full.shape[:-1] and comp.shape[-1] == b - a
                      and torch.allclose(comp, full[..., a:b], rtol=1e-3, atol=1e-4 * (1.0 + float(full.abs().max()))))
            if not ok:
                warnings.warn("MixtureCDFCoupling: compact_params disabled — the sub-network's output is not its last Linear's rows")
                self.disable_compact_params()
                return super().run_network(x, length=length, **kwargs)
            self._compact_checked = True
        lin.rows = self._compact_rows
        try:
            return super().run_network(x, length=length, **kwargs)
        finally:
            lin.rows = None

    def forward(self, z, ldj=None, reverse=False, channel_padding_mask=None, **kwargs):
        nn_out = self.run_network(x=z * self._prepare_mask(self.mask, z), **kwargs)
        # the incoming ldj is ignored on purpose: the reference overwrites it (:63) and the caller sums
        if Fn.needs_grad(z, nn_out, self.scaling_factor, self.mixture_scaling_factor):
            if reverse:
                raise NotImplementedError("the mixture-CDF inverse is not differentiable (the reference never differentiates "
                                          "its bisection either); run it under torch.no_grad()")
            z_out, layer_ldj, reg = Fn.MixtureCouplingFn.apply(
                z, nn_out, self.scaling_factor, self.mixture_scaling_factor, None, self.mask, channel_padding_mask,
                self.num_mixtures, self.regularizer_max, self.regularizer_factor, self.training, True, True)
            return z_out, layer_ldj, {"ldj": layer_ldj, "regularizer_ldj": reg}
        z_out, layer_ldj, reg = ops.mixture_coupling(
            z, nn_out, self.mask, self.num_mixtures, self.scaling_factor, self.mixture_scaling_factor,
            reverse=reverse, channel_padding_mask=channel_padding_mask, reg_max=self.regularizer_max,
            reg_factor=self.regularizer_factor, is_training=self.training, ldj=None)
        detail = {"ldj": layer_ldj}
        if reg is not None and not reverse:       # the reference has no regulariser entry in reverse (:79-80)
            detail["regularizer_ldj"] = reg
        return z_out, layer_ldj, detail

    @staticmethod
    def get_mixt_params(nn_out, mask, num_mixtures, scaling_factor=None, mixture_scaling_factor=None):
        """Five fp64 tensors (t, log_s, log_pi, mixt_t, mixt_log_s), tanh-bounded and masked."""
        if Fn.needs_grad(nn_out, scaling_factor, mixture_scaling_factor):
            return Fn.MixtureParamsFn.apply(nn_out, scaling_factor, mixture_scaling_factor, mask, num_mixtures)
        return ops.mixture_params(nn_out, mask, num_mixtures, scaling_factor, mixture_scaling_factor)

    @staticmethod
    def run_with_params(orig_z, t, log_s, log_pi, mixt_t, mixt_log_s, reverse=False,
                        reg_max=-1, reg_factor=1, mask=None, channel_padding_mask=None,
                        is_training=True, return_reg_ldj=False):
        if Fn.needs_grad(orig_z, t, log_s, log_pi, mixt_t, mixt_log_s):
            if reverse:
                raise NotImplementedError("the mixture-CDF inverse is not differentiable; run it under torch.no_grad()")
            z_out, ldj, reg = Fn.MixtureTransformFn.apply(orig_z, t, log_s, log_pi, mixt_t, mixt_log_s, mask, channel_padding_mask,
                                                          reg_max, reg_factor, is_training)
            return (z_out, ldj, reg) if return_reg_ldj else (z_out, ldj)
        z_out, ldj, reg = ops.mixture_transform(orig_z, t, log_s, log_pi, mixt_t, mixt_log_s, reverse=reverse,
                                                reg_max=reg_max, reg_factor=reg_factor, mask=mask,
                                                channel_padding_mask=channel_padding_mask, is_training=is_training)
        if return_reg_ldj:
            return z_out, ldj, (None if reverse else reg)
        return z_out, ldj

    def info(self):
        is_channel_mask = (self.mask.size(0) == 1)
        s = "Mixture CDF Coupling Layer - Input size %i" % (self.c_in)
        if self.block_type is not None:
            s += ", block type %s" % (self.block_type)
        s += ", %i mixtures" % (self.num_mixtures) + \
             ", mask ratio %.2f, %s mask" % ((1 - self.mask).mean().item(), "channel" if is_channel_mask else "chess")
        return s
