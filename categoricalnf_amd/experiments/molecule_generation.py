"""Three-stage GraphCNF for molecule generation on the MI355X layers (BASELINE configs[4]).

What the reference's experiments/molecule_generation/graphCNF.py:26-404 computes, restated as a pipeline of three
stages over the drop-in layers (same sub-module names, so its checkpoints load: node_encoding, edge_attr_encoding,
edge_virtual_encoding, edge_virtual_decoder, step1_flows, step2_flows, step3_flows):

  stage 1  nodes:          categorical encoder of the node types, then n1 x (ActNorm, 1x1 conv, mixture-CDF coupling
                           whose sub-network sees the full typed adjacency)                          (:281-294)
  stage 2  edge attributes: the upper-triangular node pairs become an edge list; real edges get their type encoded,
                           then n2 x (ActNorm, 1x1 conv, mixture-CDF coupling) on nodes AND edges at once (:296-316)
  stage 3  virtual edges:  absent edges get a one-class latent, a small decoder scores edge / no edge, then
                           n3 x the same node+edge flow steps; finally the prior log-prob of ALL edge latents joins
                           the log-det (:318-347, :250-253)

and the whole thing backwards for sampling (:255-273).  Every layer call goes through the HIP kernels of this
package (ActNorm, 1x1 conv, mixture-CDF coupling, encoders); the coupling sub-networks are PyTorch modules the caller
may provide: `node_subnet(c_out)` for stage 1 and `edge_subnet(stage, c_out_nodes, c_out_edges)` for stages 2 / 3.  The
defaults are this package's RGCNNet and Edge-GNN (layers/networks/edge_gnn.py: the reference's architecture and parameter
names — graphCNF.py:125-160 — as dense masked attention), built with the reference's hyper-parameters
(coupling_hidden_size_nodes 256, coupling_hidden_size_edges 128, coupling_hidden_layers 4: graphCNF.py:80-91)."""
import numpy as np
import torch
import torch.nn as nn

from ..host_utils import create_channel_mask, get_param_val
from ..layers.categorical_encoding.decoder import DecoderLinear
from ..layers.categorical_encoding.linear_encoding import LinearCategoricalEncoding
from ..layers.categorical_encoding.mutils import create_encoding
from ..layers.flows.activation_normalization import ActNormFlow
from ..layers.flows.coupling_layer import CouplingLayer
from ..layers.flows.distributions import create_prior_distribution
from ..layers.flows.flow_layer import FlowLayer
from ..layers.flows.flow_model import FlowModel
from ..layers.flows.mixture_cdf_layer import MixtureCDFCoupling
from ..layers.flows.permutation_layers import InvertibleConv
from .graph_node_edge_coupling import NodeEdgeCoupling, NodeEdgeFlowWrapper


def pair_indices(num_nodes, device):
    """(i, j) of every unordered node pair i < j, row-major: the order of the reference's edge list."""
    idx = torch.triu_indices(num_nodes, num_nodes, offset=1, device=device)
    return idx[0], idx[1]


def get_adjacency_indices(num_nodes, length):
    i, j = pair_indices(num_nodes, length.device)
    valid = ((i[None, :] < length[:, None]) & (j[None, :] < length[:, None])).float()
    return valid, (i, j)


def adjacency2pairs(adjacency, length):
    """[B, N, N] typed adjacency -> ([B, E] edge types of the pairs, (i, j), [B, E] validity mask), E = N (N-1) / 2."""
    n = adjacency.shape[1]
    valid, (i, j) = get_adjacency_indices(n, length)
    return adjacency[:, j, i], (i, j), valid


def pairs2adjacency(num_nodes, pairs, length, x_indices):
    """[B, E] per-pair values -> symmetric int64 [B, N, N] (one vectorised scatter instead of the per-sample loop)."""
    i, j = x_indices
    adj = pairs.new_zeros(pairs.size(0), num_nodes, num_nodes)
    adj[:, j, i] = pairs
    return (adj + adj.transpose(1, 2)).long()


def _run(layer, *z, ldj, detail=None, **kwargs):
    """Call a flow layer, add its log-det to the running one; returns (outputs..., ldj).  Layers return 2-, 3- or
    4-tuples: tensors..., layer log-det[, detail]."""
    out = layer(*z, **kwargs) if len(z) == 1 else layer(z_nodes=z[0], z_edges=z[1], **kwargs)
    n_t = len(z)
    tensors, layer_ldj = out[:n_t], out[n_t]
    if detail is not None:
        detail.append(out[n_t + 1] if len(out) > n_t + 1 else layer_ldj)
    return (*tensors, ldj + layer_ldj)


class GraphCNF(FlowModel):

    def __init__(self, model_params, dataset_class, node_subnet=None, edge_subnet=None, **kwargs):
        super().__init__(layers=None, name="GraphCNF")
        self.model_params = model_params
        self.dataset_class = dataset_class
        ds = dataset_class
        self.max_num_nodes, self.num_node_types = ds.max_num_nodes(), ds.num_node_types()
        self.num_edge_types, self.num_max_neighbours = ds.num_edge_types(), ds.num_max_neighbours()
        p = model_params
        quiet = dict(warning_if_default=False)
        self.prior_distribution = create_prior_distribution(get_param_val(p, "prior_distribution", default_val=dict(), **quiet))

        # encoders: node types, edge types (without the "no edge" class), and the single-class latent of absent edges
        self.node_encoding = create_encoding(p["categ_encoding_nodes"], dataset_class=ds, vocab_size=self.num_node_types,
                                             category_prior=ds.get_node_prior(data_root="data/"))
        self.edge_attr_encoding = create_encoding(p["categ_encoding_edges"], dataset_class=ds, vocab_size=self.num_edge_types,
                                                  category_prior=ds.get_edge_prior(data_root="data/"))
        dn, de = self.node_encoding.D, self.edge_attr_encoding.D
        self.encoding_dim_nodes, self.encoding_dim_edges = dn, de
        self.edge_virtual_encoding = LinearCategoricalEncoding(
            num_dimensions=de, flow_config={"num_flows": p["encoding_virtual_num_flows"], "hidden_layers": 2, "hidden_size": 128},
            dataset_class=ds, vocab_size=1)
        # molecules are sparse: ~10 % of the pairs are bonded
        self.edge_virtual_decoder = DecoderLinear(num_categories=2, embed_dim=de, hidden_size=128, num_layers=2,
                                                  class_prior_log=np.log(np.array([0.9, 0.1])))

        hidden_nodes = get_param_val(p, "coupling_hidden_size_nodes", default_val=256, **quiet)
        n_flows = [int(k) for k in str(get_param_val(p, "coupling_num_flows", default_val="4,6,6", **quiet)).split(",")]
        layers = get_param_val(p, "coupling_hidden_layers", default_val=4, **quiet)
        layers = [int(v) for v in layers.split(",")] if isinstance(layers, str) and "," in layers else [int(layers)] * 3
        k_nodes = get_param_val(p, "coupling_num_mixtures_nodes", default_val=16, **quiet)
        k_edges = get_param_val(p, "coupling_num_mixtures_edges", default_val=16, **quiet)
        ratio = get_param_val(p, "coupling_mask_ratio", default_val=0.5, **quiet)
        dropout = get_param_val(p, "coupling_dropout", default_val=0.0, **quiet)
        mask_n = CouplingLayer.create_channel_mask(dn, ratio=ratio)
        mask_e = CouplingLayer.create_channel_mask(de, ratio=ratio)

        if node_subnet is None:
            from ..layers.networks.graph_layers import RGCNNet, RelationGraphConv

            def node_subnet(c_out):
                return RGCNNet(c_in=dn, c_out=c_out, num_edges=self.num_edge_types, num_layers=layers[0], hidden_size=hidden_nodes,
                               max_neighbours=self.num_max_neighbours, dp_rate=dropout, rgc_layer_fun=RelationGraphConv)
        if edge_subnet is None:
            from ..layers.networks.edge_gnn import (EdgeGNN, EdgeGNNLayer, Edge2NodeAttnLayer, Edge2NodeQKVAttnLayer,
                                                     Node2EdgePlainLayer)
            hidden_edges = get_param_val(p, "coupling_hidden_size_edges", default_val=128, **quiet)

            def edge_subnet(stage, c_out_nodes, c_out_edges):
                # graphCNF.py:132-155: the edge-attribute stage attends with pair-only (sigmoid) weights, the virtual-edge
                # stage with query-key-value attention; highway skips everywhere
                node_update = Edge2NodeAttnLayer if stage == 1 else Edge2NodeQKVAttnLayer
                return EdgeGNN(c_in_nodes=dn, c_in_edges=de, c_out_nodes=c_out_nodes, c_out_edges=c_out_edges,
                               edge_gnn_layer_func=lambda: EdgeGNNLayer(
                                   edge2node_layer_func=lambda: node_update(hidden_size_nodes=hidden_nodes, hidden_size_edges=hidden_edges, skip_config=2),
                                   node2edge_layer_func=lambda: Node2EdgePlainLayer(hidden_size_nodes=hidden_nodes, hidden_size_edges=hidden_edges, skip_config=2)),
                               max_neighbours=self.num_max_neighbours, num_layers=layers[stage])

        def node_step():
            return [ActNormFlow(dn), InvertibleConv(dn),
                    MixtureCDFCoupling(c_in=dn, mask=mask_n, model_func=node_subnet, block_type="RelationGraphConv",
                                       num_mixtures=k_nodes, regularizer_max=3.5, regularizer_factor=2)]

        def pair_step(stage):
            return [NodeEdgeFlowWrapper(node_flow=ActNormFlow(c_in=dn), edge_flow=ActNormFlow(c_in=de)),
                    NodeEdgeFlowWrapper(node_flow=InvertibleConv(c_in=dn), edge_flow=InvertibleConv(c_in=de)),
                    NodeEdgeCoupling(c_in_nodes=dn, c_in_edges=de, mask_nodes=mask_n, mask_edges=mask_e,
                                     num_mixtures_nodes=k_nodes, num_mixtures_edges=k_edges,
                                     model_func=lambda c_out_nodes, c_out_edges: edge_subnet(stage, c_out_nodes, c_out_edges),
                                     regularizer_max=3.5, regularizer_factor=2)]

        self.step1_flows = nn.ModuleList([l for _ in range(n_flows[0]) for l in node_step()])
        self.step2_flows = nn.ModuleList([l for _ in range(n_flows[1]) for l in pair_step(1)])
        self.step3_flows = nn.ModuleList([l for _ in range(n_flows[2]) for l in pair_step(2)])
        self.print_overview()

    # ---- the three stages, each in both directions ------------------------------------------------------------
    def _stage1(self, z_nodes, adjacency, ldj, reverse, detail, noise=None, **kw):
        order = [self.node_encoding] + list(self.step1_flows)
        for layer in (reversed(order) if reverse else order):
            extra = dict(adjacency=adjacency) if layer is not self.node_encoding else (dict(noise=noise) if noise is not None else {})
            z_nodes, ldj = _run(layer, z_nodes, ldj=ldj, detail=detail, reverse=reverse, **extra, **kw)
        return z_nodes, ldj

    def _stage2(self, z_nodes, z_edges, ldj, reverse, detail, noise=None, **kw):
        enc_kw = dict(kw, channel_padding_mask=kw["mask_valid"].unsqueeze(dim=-1))
        if not reverse:
            attr = (z_edges - 1).clamp(min=0)          # edge type without the "no edge" class; absent edges are masked out
            if noise is not None:
                enc_kw["noise"] = noise
            z_edges, ldj = _run(self.edge_attr_encoding, attr, ldj=ldj, detail=detail, reverse=False, **enc_kw)
            for layer in self.step2_flows:
                z_nodes, z_edges, ldj = _run(layer, z_nodes, z_edges, ldj=ldj, detail=detail, reverse=False, **kw)
            return z_nodes, z_edges, ldj
        for layer in reversed(self.step2_flows):
            z_nodes, z_edges, ldj = _run(layer, z_nodes, z_edges, ldj=ldj, detail=detail, reverse=True, **kw)
        z_edges, ldj = _run(self.edge_attr_encoding, z_edges, ldj=ldj, detail=detail, reverse=True, **enc_kw)
        return z_nodes, (z_edges + 1) * kw["mask_valid"].long(), ldj       # 0 = no edge

    def _stage3(self, z_nodes, z_edges, ldj, reverse, detail, virtual_mask=None, noise=None, **kw):
        if not reverse:
            enc_kw = dict(kw, channel_padding_mask=virtual_mask.unsqueeze(dim=-1))
            if noise is not None:
                enc_kw["noise"] = noise
            zeros = z_edges.new_zeros(z_edges.shape[:-1], dtype=torch.long)
            z_virtual, ldj = _run(self.edge_virtual_encoding, zeros, ldj=ldj, detail=detail, reverse=False, **enc_kw)
            z_edges = torch.where(virtual_mask.unsqueeze(dim=-1) == 1, z_virtual, z_edges)
            # edge / no-edge decoder on every valid pair: class 0 = absent, class 1 = present
            log_probs = self.edge_virtual_decoder(z_edges)
            edge_ldj = torch.where(virtual_mask == 1, log_probs[..., 0], log_probs[..., 1] * kw["mask_valid"]).sum(dim=-1)
            ldj = ldj + edge_ldj * (kw["beta"] if "beta" in kw else 1.0)
            if detail is not None:
                with torch.no_grad():
                    detail.append({"virtual_edges_bpd": np.log2(np.exp(1)) * edge_ldj / kw["mask_valid"].sum(dim=-1)})
            for layer in self.step3_flows:
                z_nodes, z_edges, ldj = _run(layer, z_nodes, z_edges, ldj=ldj, detail=detail, reverse=False, **kw)
            return z_nodes, z_edges, ldj
        for layer in reversed(self.step3_flows):
            z_nodes, z_edges, ldj = _run(layer, z_nodes, z_edges, ldj=ldj, detail=detail, reverse=True, **kw)
        present = self.edge_virtual_decoder(z_edges).argmax(dim=-1) == 1
        return z_nodes, z_edges, ldj, kw["mask_valid"] * present.float()

    def forward(self, z, adjacency=None, ldj=None, reverse=False, get_ldj_per_layer=False, length=None, sample_temp=1.0,
                noise=None, edge_latents=None, **kwargs):
        """forward: z = int64 node types [B, N], adjacency = int64 typed adjacency [B, N, N] -> (node latents, ldj).
        reverse: z = node latents -> ((node types, adjacency), ldj).  `noise` = (u_nodes, u_edge_attr, u_virtual)
        uniform draws for the three encoders and `edge_latents` the stage-3 edge latents of the reverse pass (parity
        runs; drawn on the fly otherwise)."""
        if ldj is None:
            ldj = z.new_zeros(z.size(0), dtype=torch.float32)
        if length is not None:
            kwargs["length"] = length
            kwargs["channel_padding_mask"] = create_channel_mask(length, max_len=z.size(1))
        u_nodes, u_attr, u_virtual = noise if noise is not None else (None, None, None)
        detail = []
        if not reverse:
            z_nodes, ldj = self._stage1(z, adjacency, ldj, False, detail, noise=u_nodes, **kwargs)
            pairs, x_indices, valid = adjacency2pairs(adjacency=adjacency, length=length)
            real = valid * (pairs != 0).to(valid.dtype)
            z_nodes, z_edges, ldj = self._stage2(z_nodes, pairs, ldj, False, detail, noise=u_attr, mask_valid=real, x_indices=x_indices,
                                                 binary_adjacency=(adjacency > 0).long(), **kwargs)
            z_nodes, z_edges, ldj = self._stage3(z_nodes, z_edges, ldj, False, detail, virtual_mask=valid * (pairs == 0).float(),
                                                 noise=u_virtual, mask_valid=valid, x_indices=x_indices, **kwargs)
            # the edge latents end here: their prior log-prob is part of the model's log-likelihood (the task only adds
            # the node prior)
            edge_log_prob = (self.prior_distribution.log_prob(z_edges) * valid.unsqueeze(dim=-1)).sum(dim=[1, 2])
            ldj = ldj + edge_log_prob
            detail.append({"adjacency_log_prob": edge_log_prob})
            out = z_nodes
        else:
            n = z.size(1)
            valid, x_indices = get_adjacency_indices(num_nodes=n, length=length)
            if edge_latents is None:
                edge_latents = self.prior_distribution.sample(shape=(z.size(0), valid.size(1), self.encoding_dim_edges),
                                                              temp=sample_temp).to(z.device)
            z_nodes, z_edges, ldj, present = self._stage3(z, edge_latents, ldj, True, detail, mask_valid=valid, x_indices=x_indices, **kwargs)
            binary = pairs2adjacency(num_nodes=n, pairs=present, length=length, x_indices=x_indices)
            z_nodes, edge_types, ldj = self._stage2(z_nodes, z_edges, ldj, True, detail, mask_valid=present, x_indices=x_indices,
                                                    binary_adjacency=binary, **kwargs)
            adjacency = pairs2adjacency(num_nodes=n, pairs=edge_types, length=length, x_indices=x_indices)
            node_types, ldj = self._stage1(z_nodes, adjacency, ldj, True, detail, **kwargs)
            out = (node_types, adjacency)
        if z.is_cuda:
            from .. import ops
            ops.check_flags(z.device, "Flow: %s" % self.name)
        return (out, ldj, detail) if get_ldj_per_layer else (out, ldj)

    # ---- data-dependent initialisation (:349-401): stage by stage, every batch pushed through the stage before ----
    def initialize_data_dependent(self, batch_list):
        with torch.no_grad():
            for z, kw in batch_list:
                kw["channel_padding_mask"] = create_channel_mask(kw["length"], max_len=z.shape[1])
            for layer in [self.node_encoding] + list(self.step1_flows):
                batch_list = FlowModel.run_data_init_layer(batch_list, layer)
            staged = []
            for z_nodes, kw in batch_list:
                pairs, x_indices, valid = adjacency2pairs(adjacency=kw["adjacency"], length=kw["length"])
                real = valid * (pairs != 0).to(valid.dtype)
                z_edges = self.edge_attr_encoding((pairs - 1).clamp(min=0), reverse=False, channel_padding_mask=real.unsqueeze(dim=-1))[0]
                kw = dict(kw, _pairs=pairs, _valid=valid, binary_adjacency=(kw["adjacency"] > 0).long(), mask_valid=real, x_indices=x_indices)
                staged.append(([z_nodes, z_edges], kw))
            for layer in self.step2_flows:
                staged = FlowModel.run_data_init_layer(staged, layer)
            final = []
            for (z_nodes, z_edges), kw in staged:
                absent = kw["_valid"] * (kw["_pairs"] == 0).float()
                z_virtual = self.edge_virtual_encoding(torch.zeros_like(kw["_pairs"]), reverse=False,
                                                       channel_padding_mask=absent.unsqueeze(dim=-1))[0]
                z_edges = z_edges * (1 - absent)[..., None] + z_virtual * absent[..., None]
                kw = {k: v for k, v in kw.items() if k != "binary_adjacency"}
                kw["mask_valid"] = kw["_valid"]
                final.append(([z_nodes, z_edges], kw))
            for layer in self.step3_flows:
                final = FlowModel.run_data_init_layer(final, layer)

    def need_data_init(self):
        return True

    def print_overview(self):
        if not hasattr(self, "step3_flows"):
            return
        print("=" * 60)
        print("GraphCNF: %i + %i + %i flow layers (nodes | edge attributes | virtual edges)"
              % (len(self.step1_flows), len(self.step2_flows), len(self.step3_flows)))
        for name, mods in (("node encoding", [self.node_encoding]), ("step 1", self.step1_flows), ("edge attribute encoding", [self.edge_attr_encoding]),
                           ("step 2", self.step2_flows), ("virtual edge encoding", [self.edge_virtual_encoding]), ("step 3", self.step3_flows)):
            print("-- %s" % name)
            for i, layer in enumerate(mods):
                print("(%2i) %s" % (i + 1, layer.info()))
        print("=" * 60)
