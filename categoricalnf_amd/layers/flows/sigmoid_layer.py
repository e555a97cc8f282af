"""Sigmoid / logit flow with log-det (layers/flows/sigmoid_layer.py:12-51) on cnf_sigmoid_flow."""
import torch

from ... import functional as Fn
from ... import ops
from ...host_utils import forbid_grad
from .flow_layer import FlowLayer


class SigmoidFlow(FlowLayer):
    """reverse=True at construction turns the layer into a logit flow."""

    def __init__(self, reverse=False):
        super().__init__()
        self.reverse_layer = reverse

    def forward(self, z, ldj=None, reverse=False, sum_ldj=True, **kwargs):
        direction = (self.reverse_layer != reverse)       # XOR of the two flags (:29)
        if sum_ldj:
            if Fn.needs_grad(z, ldj):
                return Fn.SigmoidFlowFn.apply(z, ldj, direction, 1e-5)
            return ops.sigmoid_flow(z, reverse=direction, ldj=ldj)
        forbid_grad("SigmoidFlow(sum_ldj=False)", z)
        # per-element log-det requested: rows of one element each
        flat = z.reshape(-1, 1)
        out, el = ops.sigmoid_flow(flat, reverse=direction, ldj=None)
        return out.view_as(z), el.view_as(z)

    def info(self):
        return "Sigmoid Flow"
