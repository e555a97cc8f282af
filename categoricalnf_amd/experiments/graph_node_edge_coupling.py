"""Mixture-CDF coupling applied to node AND edge latents of a graph at once, plus the wrapper that
applies a node flow and an edge flow side by side.

Interface of experiments/molecule_generation/graph_node_edge_coupling.py (NodeEdgeCoupling :11-148,
NodeEdgeFlowWrapper :151-180) including parameter / buffer names.  Where the reference calls the two
statics `get_mixt_params` + `run_with_params` (materialising five fp64 tensors per call), this runs
the fused kernel `cnf_mixture_coupling` twice (nodes, edges) on the sub-network's raw output."""
import numpy as np
import torch
import torch.nn as nn

from .. import functional as Fn
from .. import ops
from ..layers.flows.flow_layer import FlowLayer


class NodeEdgeCoupling(FlowLayer):

    def __init__(self, c_in_nodes, c_in_edges, mask_nodes, mask_edges, num_mixtures_nodes, num_mixtures_edges,
                 model_func, regularizer_max=-1, regularizer_factor=1, **kwargs):
        super().__init__()
        self.c_in_nodes, self.c_in_edges = c_in_nodes, c_in_edges
        self.num_mixtures_nodes, self.num_mixtures_edges = num_mixtures_nodes, num_mixtures_edges
        self.regularizer_max, self.regularizer_factor = regularizer_max, regularizer_factor
        self.register_buffer("mask_nodes", mask_nodes)
        self.register_buffer("mask_edges", mask_edges)
        self.c_out_nodes = self.c_in_nodes * (2 + 3 * self.num_mixtures_nodes)
        self.c_out_edges = self.c_in_edges * (2 + 3 * self.num_mixtures_edges)
        self.nn = model_func(c_out_nodes=self.c_out_nodes, c_out_edges=self.c_out_edges)
        self.scaling_factor_nodes = nn.Parameter(torch.zeros(self.c_in_nodes))
        self.scaling_factor_edges = nn.Parameter(torch.zeros(self.c_in_edges))
        self.mixture_scaling_factor_nodes = nn.Parameter(torch.zeros(self.c_in_nodes, self.num_mixtures_nodes))
        self.mixture_scaling_factor_edges = nn.Parameter(torch.zeros(self.c_in_edges, self.num_mixtures_edges))

    def _mask_view(self, name, n):
        """The mask cut to n rows, as ONE tensor object per (buffer, n): ops caches a mask's transformed-channel list per tensor
        object (reading it is a device-to-host copy), and a fresh slice on every call would read it again — under a stream
        capture (graphs.GraphedTraining) that copy is not permitted at all."""
        buf = getattr(self, name)
        key = (name, min(buf.size(0), int(n)))
        views = self.__dict__.setdefault("_mask_views", {})
        hit = views.get(key)
        if hit is None or hit[0] is not buf or hit[1] != buf._version:
            hit = (buf, buf._version, buf[None, :key[1], :])
            views[key] = hit
        return hit[2]

    def forward(self, z_nodes, z_edges, ldj=None, reverse=False, length=None, channel_padding_mask=None,
                mask_valid=None, x_indices=None, binary_adjacency=None, **kwargs):
        if ldj is None:
            ldj = z_nodes.new_zeros(z_nodes.size(0),)
        mask_nodes = self._mask_view("mask_nodes", z_nodes.size(1))
        mask_edges = self._mask_view("mask_edges", z_edges.size(1))
        nn_nodes, nn_edges = self.nn(z_nodes=mask_nodes * z_nodes, z_edges=mask_edges * z_edges, length=length,
                                     channel_padding_mask=channel_padding_mask, x_indices=x_indices,
                                     mask_valid=mask_valid, binary_adjacency=binary_adjacency)
        # padded nodes / invalid edges are never transformed, so the reference's `nn_out * mask` (:65,:78) is moot
        def run(z, nn_out, mask, K, sf, msf, pad):
            if Fn.needs_grad(z, nn_out, sf, msf):
                if reverse:
                    raise NotImplementedError("the mixture-CDF inverse is not differentiable; use torch.no_grad()")
                return Fn.MixtureCouplingFn.apply(z, nn_out, sf, msf, None, mask, pad, K, self.regularizer_max,
                                                  self.regularizer_factor, self.training, True, True)
            return ops.mixture_coupling(z, nn_out, mask, K, sf, msf, reverse=reverse, channel_padding_mask=pad,
                                        reg_max=self.regularizer_max, reg_factor=self.regularizer_factor,
                                        is_training=self.training)

        zn, ldj_n, reg_n = run(z_nodes, nn_nodes, mask_nodes, self.num_mixtures_nodes, self.scaling_factor_nodes,
                               self.mixture_scaling_factor_nodes, channel_padding_mask)
        ze, ldj_e, reg_e = run(z_edges, nn_edges, mask_edges, self.num_mixtures_edges, self.scaling_factor_edges,
                               self.mixture_scaling_factor_edges, mask_valid.unsqueeze(dim=-1))
        ldj = ldj + ldj_n + ldj_e
        detail = {"ldj": ldj}
        if not reverse:
            detail["regularizer_nodes_ldj"] = reg_n
            detail["regularizer_edges_ldj"] = reg_e
        return zn, ze, ldj, detail

    def info(self):
        rn = self.mask_nodes.sum().item() / np.prod(self.mask_nodes.shape)
        re = self.mask_edges.sum().item() / np.prod(self.mask_edges.shape)
        return "Node+Edge Mixture Coupling Layer - Nodes: c_in=%i, num_mixtures=%2i, mask_ratio=%3.2f\n" % (self.c_in_nodes, self.num_mixtures_nodes, rn) + \
               "                                   Edges: c_in=%i, num_mixtures=%2i, mask_ratio=%3.2f" % (self.c_in_edges, self.num_mixtures_edges, re)


class NodeEdgeFlowWrapper(FlowLayer):
    """Runs `node_flow` on the nodes (with `length`) and `edge_flow` on the edges (with the number of valid edges)."""

    def __init__(self, node_flow, edge_flow):
        super().__init__()
        self.node_flow = node_flow
        self.edge_flow = edge_flow

    def forward(self, z_nodes, z_edges, ldj=None, reverse=False, length=None, channel_padding_mask=None,
                mask_valid=None, **kwargs):
        z_nodes, ldj = self.node_flow(z_nodes, ldj=ldj, reverse=reverse, length=length,
                                      channel_padding_mask=channel_padding_mask, **kwargs)
        edge_length = mask_valid.sum(dim=1)
        if len(mask_valid.shape) == 2:
            mask_valid = mask_valid.unsqueeze(dim=-1)
        z_edges, ldj = self.edge_flow(z_edges, ldj=ldj, reverse=reverse, length=edge_length,
                                      channel_padding_mask=mask_valid, **kwargs)
        return z_nodes, z_edges, ldj

    def need_data_init(self):
        return self.node_flow.need_data_init() or self.edge_flow.need_data_init()

    def data_init_forward(self, z_nodes, z_edges, channel_padding_mask=None, mask_valid=None, **kwargs):
        if self.node_flow.need_data_init():
            self.node_flow.data_init_forward(z_nodes, channel_padding_mask=channel_padding_mask)
        if self.edge_flow.need_data_init():
            if len(mask_valid.shape) == 2:
                mask_valid = mask_valid.unsqueeze(dim=-1)
            self.edge_flow.data_init_forward(z_edges, channel_padding_mask=mask_valid)

    def info(self):
        return "FlowWrapper - Node layer: %s\n" % self.node_flow.info() + \
               "              Edge layer: %s" % self.edge_flow.info()
