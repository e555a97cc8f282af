"""Model assembly of the graph-colouring flow (node-based GraphCNF): categorical encoder + n x (ActNorm,
invertible 1x1 conv, mixture-CDF coupling with an RGCN-attention sub-network, CDF regulariser 3.5 x 2) + a
final ActNorm.  Interface and parameter names of experiments/graph_coloring/graph_node_flow.py:17-117."""
import torch
import torch.nn as nn

from ..host_utils import create_channel_mask, create_transformer_mask, get_param_val
from ..layers.categorical_encoding.mutils import create_encoding
from ..layers.flows.activation_normalization import ActNormFlow
from ..layers.flows.coupling_layer import CouplingLayer
from ..layers.flows.flow_model import FlowModel
from ..layers.flows.mixture_cdf_layer import MixtureCDFCoupling
from ..layers.flows.permutation_layers import InvertibleConv
from ..layers.networks.graph_layers import RGCNNet, RelationGraphAttention


class GraphNodeFlow(FlowModel):

    def __init__(self, model_params, dataset_class, **kwargs):
        super().__init__(layers=None, name="GraphCNF (node based)")
        self.model_params = model_params
        self.dataset_class = dataset_class
        self._create_layers()
        self.print_overview()

    def _create_layers(self):
        self.num_node_types = self.dataset_class.num_node_types()
        self.node_embed_flow = create_encoding(self.model_params["categ_encoding"], dataset_class=self.dataset_class,
                                               vocab_size=self.num_node_types)
        self.embed_dim = self.node_embed_flow.D
        self.flow_layers = nn.ModuleList([self.node_embed_flow] + self._create_node_flow_layers())

    def _create_node_flow_layers(self):
        p = self.model_params
        quiet = dict(warning_if_default=False)
        num_flows = get_param_val(p, "coupling_num_flows", default_val=8, **quiet)
        hidden_size = get_param_val(p, "coupling_hidden_size", default_val=384, **quiet)
        hidden_layers = get_param_val(p, "coupling_hidden_layers", default_val=4, **quiet)
        num_mixtures = get_param_val(p, "coupling_num_mixtures", default_val=16, **quiet)
        mask = CouplingLayer.create_channel_mask(self.embed_dim, ratio=get_param_val(p, "coupling_mask_ratio", default_val=0.5, **quiet))
        dropout = get_param_val(p, "coupling_dropout", default_val=0.0, **quiet)
        model_func = lambda c_out: RGCNNet(c_in=self.embed_dim, c_out=c_out, num_edges=1, num_layers=hidden_layers,
                                           hidden_size=hidden_size, dp_rate=dropout, rgc_layer_fun=RelationGraphAttention)
        layers = []
        for _ in range(num_flows):
            layers += [ActNormFlow(self.embed_dim), InvertibleConv(self.embed_dim),
                       MixtureCDFCoupling(c_in=self.embed_dim, mask=mask, model_func=model_func, block_type="GraphAttentionNet",
                                          num_mixtures=num_mixtures, regularizer_max=3.5, regularizer_factor=2)]
        return layers + [ActNormFlow(c_in=self.embed_dim)]

    def forward(self, z, adjacency, ldj=None, reverse=False, length=None, **kwargs):
        if length is not None:
            kwargs["src_key_padding_mask"] = create_transformer_mask(length, max_len=z.size(1))
            kwargs["channel_padding_mask"] = create_channel_mask(length, max_len=z.size(1))
        return super().forward(z, adjacency=adjacency, ldj=ldj, reverse=reverse, length=length, **kwargs)

    def initialize_data_dependent(self, batch_list):
        with torch.no_grad():
            for batch, kwargs in batch_list:
                kwargs["src_key_padding_mask"] = create_transformer_mask(kwargs["length"], max_len=batch.shape[1])
                kwargs["channel_padding_mask"] = create_channel_mask(kwargs["length"], max_len=batch.shape[1])
            for layer in self.flow_layers:
                batch_list = FlowModel.run_data_init_layer(batch_list, layer)

    def need_data_init(self):
        return True

    def _masks(self, z, length):
        return dict(length=length, src_key_padding_mask=create_transformer_mask(length, max_len=z.size(1)),
                    channel_padding_mask=create_channel_mask(length, max_len=z.size(1)))

    def test_reversibility(self, z, adjacency, length, tol_z=1e-2, tol_ldj=1e-1):
        """Encode, push through all coupling steps and back (graph_node_flow.py:161-222): the reference's tolerances."""
        with torch.no_grad():
            kw = self._masks(z, length)
            z0, ldj0, _ = self.node_embed_flow(z, reverse=False, adjacency=adjacency, **kw)
            zf, ldj = z0, ldj0
            for flow in self.flow_layers[1:]:
                res = flow(zf, reverse=False, adjacency=adjacency, **kw)
                zf, ldj = res[0], ldj + res[1]
            zr, ldjr = zf, ldj
            for flow in reversed(self.flow_layers[1:]):
                res = flow(zr, reverse=True, adjacency=adjacency, **kw)
                zr, ldjr = res[0], ldjr + res[1]
        return bool(((zr - z0).abs() > tol_z).sum() == 0 and ((ldjr - ldj0).abs() > tol_ldj).sum() == 0)

    def test_permutation(self, z, adjacency, length, tol_z=1e-4, tol_ldj=1e-3):
        """Relabelling the nodes of a graph permutes the latents and leaves the log-det unchanged (:120-158)."""
        with torch.no_grad():
            kw = self._masks(z, length)
            z0, ldj0, _ = self.node_embed_flow(z, reverse=False, adjacency=adjacency, **kw)
            full = (kw["channel_padding_mask"].sum(dim=[0, 2]) == z.size(0))       # positions valid in every sample
            noise = (torch.rand(z.size(1), device=z.device) - 2) * full.float() + 0.001 * torch.arange(z.size(1), device=z.device)
            perm = noise.sort(dim=0)[1]
            inv = perm.sort(dim=0)[1]
            za, la, zb, lb = z0, ldj0, z0[:, perm], ldj0
            adj_p = adjacency[:, perm][:, :, perm]
            for flow in self.flow_layers[1:]:
                ra = flow(za, reverse=False, adjacency=adjacency, **kw)
                rb = flow(zb, reverse=False, adjacency=adj_p, **kw)
                za, la, zb, lb = ra[0], la + ra[1], rb[0], lb + rb[1]
        return bool(((za - zb[:, inv]).abs() > tol_z).sum() == 0 and ((la - lb).abs() > tol_ldj).sum() == 0)


def flow_nll(model, prior, nodes, adjacency, length, beta=1.0):
    """Per-graph negative log-likelihood per node, task.py:84-130 (`_train_batch_flow` / `_eval_batch_flow`):
    (-sum over valid nodes of log p(z) - ldj) / length.  Returns ([B] nll, per-layer log-det list)."""
    z, ldj, ldj_per_layer = model(nodes, adjacency, reverse=False, get_ldj_per_layer=True, beta=beta, length=length)
    pad = create_channel_mask(length, max_len=nodes.size(1))
    neglog = -(prior.log_prob(z) * pad).sum(dim=[1, 2])
    return (neglog - ldj) / length.float(), ldj_per_layer


def sample_colorings(model, prior, adjacency, length, temp=1.0, noise=None):
    """One generation pass, task.py:170-190: latents from the prior for every node of the given graphs, the flow
    backwards, the encoder's arg-max decode.  Returns int64 [B, N] colours (padding positions are whatever the decode
    gives there; the validity count looks at the first `length` nodes only)."""
    shape = (adjacency.shape[0], adjacency.shape[1], model.embed_dim)
    with torch.no_grad():
        if noise is not None:
            z = prior.sample(shape=shape, device=adjacency.device, uniform=noise.reshape(shape))
        else:
            z = prior.sample(shape=shape, device=adjacency.device, temp=temp)
        nodes, _ = model(z, adjacency=adjacency, length=length, reverse=True)
    return nodes


def generation_validity(model, prior, batches, dataset_class):
    """`_eval_finalize_metrics` (task.py:194-215) without the host round trips: sample a colouring for every graph of
    `batches` (iterable of (nodes, adjacency, length) on the model's device) and count the valid ones where they lie.
    Returns {"valid_ratio": ..., "num_graphs": ...}."""
    valid, total = 0.0, 0
    for _, adjacency, length in batches:
        nodes = sample_colorings(model, prior, adjacency, length)
        ratio = dataset_class.evaluate_generations(nodes=nodes, adjacency=adjacency, length=length)["valid_ratio"]
        valid += ratio * adjacency.shape[0]
        total += adjacency.shape[0]
    return {"valid_ratio": valid / max(total, 1), "num_graphs": total}
