"""Node-based graph sub-networks used as coupling networks in the graph-colouring flow: relational graph
convolution / relational graph attention layers and the RGCN stack.

Same constructor arguments, parameter names and mathematics as the reference's
layers/networks/graph_layers.py (RelationGraphConv :15-50, RelationGraphAttention :53-154,
RGCNNet :157-232, GNNSkipConnection :702-735), so its checkpoints load.  These are dense GEMM /
attention sub-networks and stay PyTorch-ROCm (hipBLASLt / MFMA).  The attention is written MI355X-first as
dense masked attention — `[B,H,N,N] @ [B,H,N,C]` batched GEMMs per edge type — instead of the reference's
gather into a padded neighbour list (index_select / masked_select with a host sync on `max_neighbours`);
the result is the same softmax over each node's neighbours."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ...host_utils import one_hot



class _LinearSplitK(torch.autograd.Function):
    """y = x W^T + b with the weight gradient as a split-K batched GEMM.  The pair list of a batch is 45 000 rows of 128
    features (64 graphs x 703 pairs): `grad_out^T @ x` is then a [128 x 45 000] x [45 000 x 128] product — one output tile
    per 32 x 32 block and a serial K loop, 120-265 us on hipBLASLt (5-8 TFLOP/s; 16 % of a training step in
    profiles/r05_mfma_util_molecule.txt).  Cut into S = 16 slabs of rows it is a batched GEMM of 16 x more workgroups and
    a 16-term sum: 25-53 us for the same shapes (tools/molecule_train_probe.py: 5.55 -> 6.1 steps/s eager).  Same
    mathematics; the fp32 sum over rows is associated per slab."""

    @staticmethod
    def forward(ctx, x, weight, bias, slabs):
        ctx.save_for_backward(x, weight)
        ctx.slabs = slabs
        return F.linear(x, weight, bias)

    @staticmethod
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        g2, x2 = g.reshape(-1, g.size(-1)), x.reshape(-1, x.size(-1))
        gx = g.matmul(weight) if ctx.needs_input_grad[0] else None
        gw = gb = None
        if ctx.needs_input_grad[1]:
            S = ctx.slabs
            gw = torch.bmm(g2.view(S, -1, g2.size(1)).transpose(1, 2), x2.view(S, -1, x2.size(1))).sum(dim=0)
        if ctx.needs_input_grad[2]:
            gb = g2.sum(dim=0)
        return gx, gw, gb, None


class SplitKLinear(nn.Linear):
    """nn.Linear (same parameters, same state_dict) whose backward takes the split-K weight gradient on long inputs."""
    _MIN_ROWS = 8192

    def forward(self, x):
        rows = x.numel() // max(x.size(-1), 1)
        if x.is_cuda and rows >= self._MIN_ROWS and torch.is_grad_enabled() and self.weight.requires_grad:
            for slabs in (16, 8, 4):
                if rows % slabs == 0:
                    return _LinearSplitK.apply(x, self.weight, self.bias, slabs)
        return F.linear(x, self.weight, self.bias)

class RelationGraphConv(nn.Module):
    """h_i = W_s x_i + (1/|N_i|) sum_{j in N_i} W_{r(i,j)} x_j on layer-normed features."""

    def __init__(self, c_in, c_out, num_edges, **kwargs):
        super().__init__()
        self.c_in, self.c_out, self.num_edges = c_in, c_out, num_edges
        self.norm_layer = nn.LayerNorm(self.c_in)
        self.linear_hs = nn.Linear(self.c_in, self.c_out)
        self.linear_hr = nn.Linear(self.c_in, self.c_out * self.num_edges)

    def forward(self, x, adjacency, num_neighbours=None):
        B, N = x.size(0), x.size(1)
        if num_neighbours is None:
            num_neighbours = adjacency.sum(dim=[1, 3])
        x = self.norm_layer(x)
        hr_all = self.linear_hr(x).view(B, N, self.num_edges, self.c_out)
        # sum over source nodes and edge types, exactly the reference's (hr_all * adjacency).sum(dim=[1,3])
        hr = torch.einsum("bjec,bjie->bic", hr_all, adjacency)
        return self.linear_hs(x) + hr / num_neighbours.unsqueeze(dim=-1).clamp(min=1e-5)


class RelationGraphAttention(nn.Module):
    """Multi-head graph attention with one value / key projection per edge type (+ self connection)."""

    def __init__(self, c_in, c_out, num_edges, num_heads=4, **kwargs):
        super().__init__()
        self.c_in, self.c_out, self.num_edges, self.num_heads = c_in, c_out, num_edges, num_heads
        self.c_out_per_head = self.c_out * 2 // self.num_heads
        self.norm_layer = nn.LayerNorm(self.c_in)
        self.linear_hs = nn.Linear(self.c_in, self.c_out_per_head * self.num_heads)
        self.linear_hr = nn.Linear(self.c_in, self.c_out_per_head * self.num_heads * (self.num_edges + 1))
        self.attn_weight = nn.Parameter(torch.zeros(self.num_heads, 2, self.c_out_per_head), requires_grad=True)
        nn.init.xavier_uniform_(self.attn_weight.data, gain=1.414)
        self.output_projection = nn.Sequential(nn.GELU(), nn.Linear(self.c_out_per_head * self.num_heads, self.c_out))
        self.leaky_relu = nn.LeakyReLU(0.2)

    def forward(self, x, adjacency, **kwargs):
        B, N = x.size(0), x.size(1)
        H, C, E1 = self.num_heads, self.c_out_per_head, self.num_edges + 1
        x = self.norm_layer(x)
        hs = self.linear_hs(x).reshape(B, N, H, C)
        hr_all = self.linear_hr(x).reshape(B, N, E1, H, C)
        # contractions, not broadcast-multiply-and-sum: the gradient of attn_weight is then a batched GEMM instead of a sum over
        # B*N rows (PyTorch's two-pass reduction, whose memset node a captured training step must not contain: graphs.py)
        hs_attn = torch.einsum("bnhc,hc->bnh", hs, self.attn_weight[:, 0])                   # [B,N,H]   (query side)
        hr_attn = torch.einsum("bnehc,hc->bneh", hr_all, self.attn_weight[:, 1])             # [B,N,E1,H] (key side)
        with torch.no_grad():
            eye = torch.eye(N, device=x.device, dtype=adjacency.dtype).view(1, N, N, 1).expand(B, -1, -1, -1)
            adj = torch.cat([adjacency, eye], dim=-1)                                         # self connection = last type
            connected = adj.sum(dim=-1) > 0                                                   # [B,N,N]
        # key logit of neighbour j as seen from i: the projection of the edge type that links them
        key = torch.einsum("bije,bjeh->bijh", adj, hr_attn)
        logits = self.leaky_relu(hs_attn.unsqueeze(dim=2) + key)
        logits = logits.masked_fill(~connected.unsqueeze(dim=-1), -9e15)
        probs = torch.softmax(logits, dim=2)                                                  # over neighbours j
        # values: sum_e (probs * adj_e) @ hr_all[:, :, e]  — one batched GEMM per edge type
        out = x.new_zeros(B, H, N, C)
        for e in range(E1):
            out = out + torch.matmul((probs * adj[..., e:e + 1]).permute(0, 3, 1, 2), hr_all[:, :, e].permute(0, 2, 1, 3))
        out = out.permute(0, 2, 1, 3).reshape(B, N, H * C)
        return self.output_projection(out)


class GNNSkipConnection(nn.Module):
    """0: residual, 1: gated residual, 2: highway combination of the block input and the block output."""

    def __init__(self, hidden_size, config=0, input_size=-1, dp_rate=0.0):
        super().__init__()
        self.hidden_size = hidden_size
        self.input_size = input_size if input_size > 0 else hidden_size
        self.config = config
        self.dp_rate = dp_rate
        assert config in (0, 1, 2), "[!] ERROR: Unknown skip connection config \"%s\"" % str(config)
        self.skip_layer = SplitKLinear(self.input_size, self.hidden_size * (1 if config == 0 else 2))     # also runs on pair lists
        if self.dp_rate > 0.0:
            self.skip_layer = nn.Sequential(nn.Dropout(self.dp_rate), self.skip_layer)

    def forward(self, orig, feat):
        if self.config == 0:
            return orig + self.skip_layer(feat)
        val, gate_logits = self.skip_layer(feat).chunk(2, dim=-1)
        gate = torch.sigmoid(gate_logits)
        return orig + val * gate if self.config == 1 else orig * (1 - gate) + val * gate


class RGCNNet(nn.Module):
    """Input MLP (+ embedding of the clamped neighbour count), `num_layers` x [graph layer, GELU, dropout, skip], output MLP."""

    def __init__(self, c_in, c_out, num_edges, num_layers, hidden_size, dp_rate=0.0, max_neighbours=4,
                 skip_config=2, rgc_layer_fun=RelationGraphConv, **kwargs):
        super().__init__()
        self.c_in, self.c_out, self.num_edges, self.num_layers = c_in, c_out, num_edges, num_layers
        self.hidden_size, self.dp_rate, self.max_neighbours = hidden_size, dp_rate, max_neighbours
        if self.max_neighbours > 0:
            neighbour_embed_size = int(hidden_size // 4)
            self.neighbour_embed = nn.Linear(max_neighbours + 1, neighbour_embed_size)
        else:
            neighbour_embed_size = 0
        self.act_fn = nn.GELU()
        self.dropout = nn.Dropout(dp_rate)
        self.layers = nn.ModuleList([
            nn.ModuleList([rgc_layer_fun(c_in=hidden_size, c_out=hidden_size, num_edges=num_edges), self.act_fn, self.dropout,
                           GNNSkipConnection(hidden_size=hidden_size, config=skip_config)])
            for _ in range(num_layers)])
        self.input_layer = nn.Sequential(nn.Linear(c_in, hidden_size), self.act_fn,
                                         nn.Linear(hidden_size, hidden_size - neighbour_embed_size))
        self.output_layer = nn.Sequential(nn.LayerNorm(hidden_size), nn.Linear(hidden_size, hidden_size), self.act_fn,
                                          nn.Linear(hidden_size, c_out))

    def forward(self, x, adjacency, channel_padding_mask=None, embed_ext_input=None, **kwargs):
        adj_one_hot = one_hot(adjacency, num_classes=self.num_edges + 1)[..., 1:]       # type 0 = no edge
        num_neighbours = adj_one_hot.sum(dim=[1, 3])
        x = self.input_layer(x)
        if self.max_neighbours > 0:
            num_neighbours = num_neighbours.clamp(max=self.max_neighbours)
            x = torch.cat([x, self.neighbour_embed(one_hot(num_neighbours.long(), num_classes=self.max_neighbours + 1))], dim=-1)
        for block in self.layers:
            block_in = x
            for layer in block:
                if isinstance(layer, (RelationGraphConv, RelationGraphAttention)):
                    x = layer(x, adjacency=adj_one_hot, num_neighbours=num_neighbours)
                elif isinstance(layer, GNNSkipConnection):
                    x = layer(orig=block_in, feat=x)
                else:
                    x = layer(x)
        x = self.output_layer(x)
        if channel_padding_mask is not None:       # the adjacency is already zero for padded nodes
            x = x * channel_padding_mask
        return x


_EDGE_GNN_NAMES = ("EdgeGNN", "EdgeGNNLayer", "Node2EdgePlainLayer", "Edge2NodeQKVAttnLayer", "Edge2NodeAttnLayer")


def __getattr__(name):
    """The molecule flow's Edge-GNN lives in edge_gnn.py (which imports GNNSkipConnection from here); the reference keeps all
    of these classes in this one file, so `layers.networks.graph_layers.EdgeGNN` resolves too."""
    if name in _EDGE_GNN_NAMES:
        from . import edge_gnn
        return getattr(edge_gnn, name)
    raise AttributeError("module %r has no attribute %r" % (__name__, name))
