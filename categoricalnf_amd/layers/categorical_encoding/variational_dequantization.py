"""Variational dequantisation baseline (layers/categorical_encoding/variational_dequantization.py):
u ~ U[0,1) -> logit -> (ActNorm, affine coupling conditioned on the class embedding)^n -> sigmoid,
z = category + noise; decode = floor.  Every flow step is one of the HIP layer kernels."""
import torch
import torch.nn as nn

from ... import ops
from ...host_utils import get_param_val
from ..flows.activation_normalization import ActNormFlow
from ..flows.coupling_layer import CouplingLayer
from ..flows.flow_layer import FlowLayer
from ..flows.sigmoid_layer import SigmoidFlow
from .decoder import create_embed_layer


class VariationalDequantization(FlowLayer):

    def __init__(self, flow_config, vocab=None, vocab_size=-1, default_embed_layer_dims=128, **kwargs):
        super().__init__()
        self.embed_layer, self.vocab_size = create_embed_layer(vocab, vocab_size, default_embed_layer_dims)
        self.flow_layers = _create_flows(flow_config, self.embed_layer.weight.shape[1])
        self.sigmoid_flow = SigmoidFlow(reverse=True)

    def forward(self, z, ldj=None, reverse=False, noise=None, **kwargs):
        if ldj is None:
            ldj = z.new_zeros(z.size(0), dtype=torch.float32)
        if not reverse:
            u = noise if noise is not None else torch.rand_like(z, dtype=torch.float32)
            u = u.reshape(z.shape).unsqueeze(dim=-1)                               # [B,N,1] in [0,1)
            u, ldj = self.sigmoid_flow(u, ldj=ldj, reverse=False)                  # -> (-inf, inf)
            u, ldj = self._flow_forward(u, z, ldj, **kwargs)
            u, ldj = self.sigmoid_flow(u, ldj=ldj, reverse=True)                   # -> [0,1]
            if ops._STRICT:
                assert (u < 0.0).sum() == 0 and (u > 1.0).sum() == 0, \
                    "ERROR: Variational Dequantization output is out of bounds."
            z_out = z.to(torch.float32).unsqueeze(dim=-1) + u
        else:
            z_out = torch.floor(z).clamp(min=0, max=self.vocab_size - 1).long().squeeze(dim=-1)
        return z_out, ldj

    def _flow_forward(self, rand_inp, z, ldj, **kwargs):
        embed_features = self.embed_layer(z)
        for flow in self.flow_layers:
            rand_inp, ldj = flow(rand_inp, ldj, ext_input=embed_features, reverse=False, **kwargs)
        return rand_inp, ldj

    def info(self):
        s = "Variational Dequantization with %i flows.\n" % (len(self.flow_layers))
        s += "\n".join(["-> [%i] " % (i + 1) + flow.info() for i, flow in enumerate(self.flow_layers)])
        return s


def _create_flows(config, embed_dims):
    """n x [ActNorm(1, no data init), affine coupling with alternating chess masks] (:75-98)."""
    num_flows = get_param_val(config, "num_flows", 4)
    model_func = get_param_val(config, "model_func", allow_default=False)
    block_type = get_param_val(config, "block_type", None)
    layers = []
    for index in range(num_flows):
        mask = CouplingLayer.create_chess_mask()
        if index % 2 == 0:
            mask = 1 - mask
        layers += [ActNormFlow(c_in=1, data_init=False),
                   CouplingLayer(c_in=1, mask=mask, model_func=model_func, block_type=block_type)]
    return nn.ModuleList(layers)
