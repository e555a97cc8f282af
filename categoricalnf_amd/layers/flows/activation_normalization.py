"""ActNorm (per-channel affine) and ExtActNorm (affine predicted from an external input).

Interface of layers/flows/activation_normalization.py: ActNormFlow (:9-77) with `bias`/`scales`
[1,1,D] parameters and data-dependent init, ExtActNormFlow (:80-180) with `pred_net`.
Both add into the caller's `ldj` IN PLACE like the reference (:37,:40,:133,:139)."""
import torch
import torch.nn as nn

from ... import functional as Fn
from ... import ops
from .flow_layer import FlowLayer


class ActNormFlow(FlowLayer):

    def __init__(self, c_in, data_init=True):
        super().__init__()
        self.c_in = c_in
        self.data_init = data_init
        self.bias = nn.Parameter(torch.zeros(1, 1, self.c_in))
        self.scales = nn.Parameter(torch.zeros(1, 1, self.c_in))

    def forward(self, z, ldj=None, reverse=False, length=None, channel_padding_mask=None, **kwargs):
        if Fn.needs_grad(z, self.bias, self.scales, ldj):
            # out of place on `ldj` under autograd (nobody relies on the alias while training)
            return Fn.ActNormFn.apply(z, self.bias, self.scales, ldj, length, channel_padding_mask, reverse)
        return ops.actnorm(z, self.bias, self.scales, reverse=reverse, length=length,
                           channel_padding_mask=channel_padding_mask, ldj=ldj)

    def need_data_init(self):
        return self.data_init

    def data_init_forward(self, input_data, channel_padding_mask=None, **kwargs):
        """bias = -mean, scales = -0.5*log(var) over batch and sequence (:55-67)."""
        bias, scales = ops.actnorm_data_init(input_data, channel_padding_mask)
        self.bias.data = bias
        self.scales.data = scales
        with torch.no_grad():
            out, _ = ops.actnorm(input_data, self.bias, self.scales, channel_padding_mask=channel_padding_mask)
            b2, s2 = ops.actnorm_data_init(out, channel_padding_mask)
        print("[INFO - ActNorm] New mean", (-b2).view(-1))
        print("[INFO - ActNorm] New variance", torch.exp(-s2).view(-1))

    def info(self):
        return "Activation Normalizing Flow (c_in=%i)" % (self.c_in)


class ExtActNormFlow(FlowLayer):

    def __init__(self, c_in, net, zero_init=False, data_init=False, make_unique=False):
        super().__init__()
        self.c_in = c_in
        self.data_init = data_init
        self.make_unique = make_unique
        self.pred_net = net
        if zero_init:
            if hasattr(self.pred_net, "initialize_zeros"):
                self.pred_net.initialize_zeros()
            elif isinstance(self.pred_net, nn.Sequential):
                self.pred_net[-1].weight.data.zero_()
                self.pred_net[-1].bias.data.zero_()

    def _run_nn(self, ext_input):
        if not self.make_unique:
            return self.pred_net(ext_input)
        # run the predictor once per distinct input value and gather (:105-113)
        uniq, inverse = torch.unique(ext_input, return_inverse=True)
        outs = self.pred_net(uniq)
        return outs.index_select(0, inverse.reshape(-1)).reshape(tuple(ext_input.shape) + outs.shape[-1:])

    def forward(self, z, ldj=None, reverse=False, ext_input=None, channel_padding_mask=None,
                layer_share_dict=None, **kwargs):
        if ext_input is None:
            print("[!] WARNING: External input in ExtActNormFlow is None. Using default params...")
            nn_out = z.new_zeros(z.size(0), z.size(1), 2 * z.size(2))
        else:
            nn_out = self._run_nn(ext_input)
        if Fn.needs_grad(z, nn_out, ldj):
            z_out, ldj_out = Fn.ExtActNormFn.apply(z, nn_out, ldj, channel_padding_mask, reverse)
        else:
            z_out, ldj_out = ops.ext_actnorm(z, nn_out, reverse=reverse, channel_padding_mask=channel_padding_mask, ldj=ldj)
        if layer_share_dict is not None and not reverse:
            bias, scales = nn_out.chunk(2, dim=2)
            scales = torch.tanh(scales)
            layer_share_dict["t"] = (layer_share_dict["t"] + bias) * torch.exp(scales)
            layer_share_dict["log_s"] = layer_share_dict["log_s"] + scales
        return z_out, ldj_out

    def need_data_init(self):
        return self.data_init

    def data_init_forward(self, input_data, channel_padding_mask=None, **kwargs):
        """Write (-mean, -0.5 log var) into the predictor's last bias (:151-170)."""
        bias, scales = ops.actnorm_data_init(input_data, channel_padding_mask)
        new_bias = torch.cat([bias, scales], dim=-1).squeeze()
        if isinstance(self.pred_net, nn.Sequential):
            self.pred_net[-1].bias.data = new_bias
        else:
            self.pred_net.set_bias(new_bias)
        print("[INFO - External ActNorm] initialised bias", new_bias)

    def info(self):
        return "External Activation Normalizing Flow (c_in=%i)" % (self.c_in)
