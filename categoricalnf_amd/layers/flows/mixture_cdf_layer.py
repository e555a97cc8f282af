"""Logistic-mixture CDF coupling (Flow++ style) on the fp64 HIP kernels.

Interface of layers/flows/mixture_cdf_layer.py: constructor (:13-42), forward (:45-92) returning the
3-tuple (z, ldj, {"ldj", "regularizer_ldj"}), and the two statics that other layers call
(get_mixt_params :145-180, run_with_params :95-142)."""
import torch
import torch.nn as nn

from ... import functional as Fn
from ... import ops
from .coupling_layer import CouplingLayer


class MixtureCDFCoupling(CouplingLayer):

    def __init__(self, c_in, mask, model_func, block_type=None, num_mixtures=10,
                 regularizer_max=-1, regularizer_factor=1, **kwargs):
        super().__init__(c_in=c_in, mask=mask, model_func=model_func, block_type=block_type,
                         c_out=c_in * (2 + num_mixtures * 3), **kwargs)
        self.num_mixtures = num_mixtures
        self.mixture_scaling_factor = nn.Parameter(torch.zeros(self.c_in, self.num_mixtures))
        self.regularizer_max = regularizer_max
        self.regularizer_factor = regularizer_factor

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
