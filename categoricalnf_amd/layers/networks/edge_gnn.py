"""Edge-GNN: the coupling sub-network of the molecule flow's edge stages (GraphCNF steps 2 and 3), which carries features
on nodes AND on node pairs.  Drop-in for the reference's `EdgeGNN` and its layer classes (layers/networks/graph_layers.py:
EdgeGNNLayer :242, Node2EdgePlainLayer :297, Edge2NodeQKVAttnLayer :388, Edge2NodeAttnLayer :561, EdgeGNN :737; wired in
experiments/molecule_generation/graphCNF.py:125-160): same class names, constructor arguments, forward keywords and
parameter names, so the reference's `state_dict`s load strictly.

Written from the mathematics, not from the reference's code, and MI355X-first.  The reference gathers every node's
neighbours into padded lists (`topk` over the adjacency with a host sync on the largest degree, `index_select` chains,
a sort of the doubled pair list, compaction of the valid pairs to a flat list and back); here a graph of V nodes is DENSE:

  * pair features live in the pair list [B, E, C] (E = V (V - 1) / 2, pair p = (i < j)); a [V, V] table of pair indices
    (built once per call from `x_indices`) turns a per-pair quantity into a symmetric [B, V, V, C] view with one gather;
  * "node i attends over its neighbours j" is a masked [V, V] attention per head: `softmax(Q K^T / sqrt(d) + A)` and
    `P @ V` are batched GEMMs (hipBLASLt / MFMA), the per-pair terms `sum_j P_ij E_ij` a scaling of the pair values in pair
    space followed by two batched GEMMs with the pair list's constant incidence matrices (all heads at once);
  * invalid pairs / padded nodes are masked, never compacted: no data-dependent shapes, no host syncs, capturable in a
    HIP graph.

Who is a neighbour: `binary_adjacency` [B, V, V] when the caller passes it (the reference then takes its sparse path,
which attends over exactly those nodes), otherwise the valid pairs `mask_valid` [B, E] (its dense path).

The layers (h = node features [B, V, Hn], e = pair features [B, E, He], LN = LayerNorm, hw = highway skip
`x (1 - g) + v g` with `[v, g] = W f`, GNNSkipConnection config 2):
  node update, edge-attention form (step 2):   [s, c] = W_n LN(h);  m_ij = W_e LN(e_ij);  a_ij = sigmoid(w_l . LN(e_ij));
        h_i' = hw(h_i, gelu(s_i + sum_j a_ij / max(sum_j a_ij, 1e-5) (m_ij + c_j)))                       (per head)
  node update, query-key-value form (step 3):  [q, k, v] = W LN(h);  m_ij = W_e LN(e_ij);  b_ij = w_b . LN(e_ij);
        P_i. = softmax_j(q_i . k_j / sqrt(d) + b_ij);   h_i' = hw(h_i, gelu(W_o [LN(h_i), sum_j P_ij (v_j + m_ij)]))
  pair update:                                  e_ij' = hw(e_ij, gelu(W_e LN(e_ij) + W_n LN(h_i) + W_n LN(h_j)))  on valid pairs, else 0.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .graph_layers import GNNSkipConnection, SplitKLinear as _Linear

_NEG = -9e15        # the reference's masking constant (its softmax rows without a neighbour are zeroed afterwards)


def pair_table(x_indices, num_nodes):
    """[V, V] long: entry (i, j) = index of pair {i, j} in the pair list, 0 on the diagonal (masked by every user)."""
    x1, x2 = x_indices
    table = x1.new_zeros(num_nodes, num_nodes)
    p = torch.arange(x1.numel(), device=x1.device, dtype=x1.dtype)
    table[x1, x2] = p
    table[x2, x1] = p
    return table


def pairs_to_dense(values, table):
    """[B, E, ...] per-pair values -> [B, V, V, ...] (symmetric; the diagonal holds pair 0's values and must be masked)."""
    V = table.size(0)
    return values.index_select(1, table.reshape(-1)).reshape((values.size(0), V, V) + tuple(values.shape[2:]))


def neighbour_mask(mask_valid, table, binary_adjacency=None):
    """[B, V, V] bool: j is a neighbour of i."""
    V = table.size(0)
    if binary_adjacency is not None:
        m = binary_adjacency > 0
    else:
        m = pairs_to_dense(mask_valid, table) > 0
    return m & ~torch.eye(V, dtype=torch.bool, device=m.device)


class _GraphContext:
    """What every layer of one EdgeGNN call shares: the pair table, the neighbour mask, the valid-pair mask, and the pair
    list's incidence matrices (which node is a pair's first / second end) with the positions of (i, j) and (j, i) in a
    flattened [V, V] matrix."""

    def __init__(self, x_indices, mask_valid, num_nodes, binary_adjacency=None):
        self.x1, self.x2 = x_indices
        self.table = pair_table(x_indices, num_nodes)
        self.neigh = neighbour_mask(mask_valid, self.table, binary_adjacency)          # [B, V, V]
        self.has_neigh = self.neigh.any(dim=-1)                                         # [B, V]
        self.valid = mask_valid > 0                                                     # [B, E]
        nodes = torch.arange(num_nodes, device=self.x1.device).unsqueeze(1)
        dtype = mask_valid.dtype if mask_valid.is_floating_point() else torch.float32
        self.inc1 = (nodes == self.x1.unsqueeze(0)).to(dtype)                           # [V, E]: inc1[i, p] = 1 iff x1[p] == i
        self.inc2 = (nodes == self.x2.unsqueeze(0)).to(dtype)
        self.inc12 = torch.cat([self.inc1, self.inc2], dim=1)                           # [V, 2E]
        self.flat12 = self.x1 * num_nodes + self.x2                                     # [E]: position of (x1, x2) in [V * V]
        self.flat21 = self.x2 * num_nodes + self.x1


def _context(kwargs, x_indices, mask_valid, num_nodes):
    ctx = kwargs.get("graph_context")
    if ctx is None:
        ctx = _GraphContext(x_indices, mask_valid, num_nodes, kwargs.get("binary_adjacency"))
    return ctx


class Node2EdgePlainLayer(nn.Module):
    """Pair update from the two end nodes (graph_layers.py:297-331)."""

    def __init__(self, hidden_size_nodes, hidden_size_edges, skip_config=0, dp_rate=0.0, act_fn=nn.GELU):
        super().__init__()
        self.hidden_size_nodes, self.hidden_size_edges = hidden_size_nodes, hidden_size_edges
        self.node_feat_layer = nn.Sequential(nn.LayerNorm(hidden_size_nodes), _Linear(hidden_size_nodes, hidden_size_edges))
        self.edge_feat_layer = nn.Sequential(nn.LayerNorm(hidden_size_edges), _Linear(hidden_size_edges, hidden_size_edges))
        self.skip_layer = GNNSkipConnection(hidden_size_edges, config=skip_config, dp_rate=dp_rate)
        self.dropout = nn.Dropout(dp_rate)
        self.act_fn = act_fn()

    def forward(self, node_feat, edge_feat, x_indices, mask_valid, **kwargs):
        ctx = _context(kwargs, x_indices, mask_valid, node_feat.size(1))
        from_nodes = self.node_feat_layer(self.dropout(node_feat))                      # [B, V, He]
        ends = from_nodes.index_select(1, ctx.x1) + from_nodes.index_select(1, ctx.x2)  # [B, E, He]
        comb = self.act_fn(self.dropout(self.edge_feat_layer(self.dropout(edge_feat)) + ends))
        out = self.skip_layer(orig=edge_feat, feat=comb)
        return torch.where(ctx.valid.unsqueeze(-1), out, torch.zeros_like(out))


class _Edge2NodeBase(nn.Module):
    def _heads(self, t, B, n):
        return t.reshape(B, n, self.num_heads, self.hidden_size_per_head)

    def _pair_terms(self, probs, pair_vals, ctx):
        """sum_j P[b,h,i,j] * m[b, pair(i,j), h, :] -> [B, V, H, d], in PAIR space: pair p = (i, j) hands P[i,j] m_p to node i
        and P[j,i] m_p to node j, and "hand to the end node" is a product with the constant incidence matrix — one batched
        [V, 2E] x [2E, H d] GEMM (MFMA) for all heads and both ends at once.  (Round 4 gathered a dense [B, V, V, d] copy of the pair values
        per head in a Python loop and contracted it with one-row matrix products: 16x16 / 32x32 GEMM tiles at 2-5 % of the
        MFMA peak, profiles/r04_mfma_util_molecule.txt.)  The incidence products add exact zeros for every other pair."""
        B, H, V, _ = probs.shape
        E = pair_vals.size(1)
        flat = probs.reshape(B, H, V * V)
        w12 = flat.index_select(2, ctx.flat12).transpose(1, 2).unsqueeze(-1)           # [B, E, H, 1]: P[x1, x2]
        w21 = flat.index_select(2, ctx.flat21).transpose(1, 2).unsqueeze(-1)           #               P[x2, x1]
        both = torch.stack([w12, w21], dim=1) * pair_vals.unsqueeze(1)                  # [B, 2, E, H, d]: one pass over the pair values
        out = torch.matmul(ctx.inc12.to(pair_vals.dtype), both.reshape(B, 2 * E, -1))   # [V, 2E] x [B, 2E, H d]: one GEMM, K = 2E
        return out.reshape(B, V, H, -1)


class Edge2NodeQKVAttnLayer(_Edge2NodeBase):
    """Node update: dot-product attention over the neighbours with a per-pair logit bias and per-pair value terms
    (graph_layers.py:388-558; its dense and its neighbour-list path compute this same function)."""

    def __init__(self, hidden_size_nodes, hidden_size_edges, num_heads=4, dp_rate=0.0, act_fn=nn.GELU, skip_config=2):
        super().__init__()
        self.hidden_size_nodes, self.hidden_size_edges, self.num_heads = hidden_size_nodes, hidden_size_edges, num_heads
        self.hidden_size_per_head = hidden_size_nodes // num_heads
        self.dot_prod_scaling = float(self.hidden_size_per_head) ** -0.5
        inner = num_heads * self.hidden_size_per_head
        self.node_query_key_val_layer = _Linear(hidden_size_nodes, inner * 3)
        self.edge_val_layer = _Linear(hidden_size_edges, inner)
        self.edge_adj_layer = _Linear(hidden_size_edges, num_heads)
        self.output_projection = _Linear(inner + hidden_size_nodes, hidden_size_nodes)
        self.skip_layer = GNNSkipConnection(hidden_size_nodes, config=skip_config, input_size=hidden_size_nodes, dp_rate=dp_rate)
        self.dropout = nn.Dropout(dp_rate)
        self.act_fn = act_fn()
        self.node_normalization = nn.LayerNorm(hidden_size_nodes)
        self.edge_normalization = nn.LayerNorm(hidden_size_edges)

    def forward(self, node_feat, edge_feat, x_indices, mask_valid, **kwargs):
        B, V = node_feat.size(0), node_feat.size(1)
        ctx = _context(kwargs, x_indices, mask_valid, V)
        h_in, e_in = self.node_normalization(node_feat), self.edge_normalization(edge_feat)
        q, k, v = self.node_query_key_val_layer(self.dropout(h_in)).chunk(3, dim=-1)
        q, k, v = (self._heads(t, B, V).permute(0, 2, 1, 3) for t in (q, k, v))          # [B, H, V, d]
        pair_val = self._heads(self.edge_val_layer(e_in), B, edge_feat.size(1))          # [B, E, H, d]
        pair_bias = pairs_to_dense(self.edge_adj_layer(e_in), ctx.table).permute(0, 3, 1, 2)       # [B, H, V, V]
        logits = torch.matmul(q, k.transpose(-1, -2)) * self.dot_prod_scaling + pair_bias
        logits = logits.masked_fill(~ctx.neigh.unsqueeze(1), _NEG)
        probs = torch.softmax(logits, dim=-1) * ctx.has_neigh[:, None, :, None].to(logits.dtype)
        attn = torch.matmul(probs, v).permute(0, 2, 1, 3) + self._pair_terms(probs, pair_val, ctx)     # [B, V, H, d]
        comb = self.act_fn(self.dropout(self.output_projection(torch.cat([h_in, attn.reshape(B, V, -1)], dim=-1))))
        return self.skip_layer(orig=node_feat, feat=comb)


class Edge2NodeAttnLayer(_Edge2NodeBase):
    """Node update whose attention weights come from the pairs alone, normalised sigmoids instead of a softmax
    (graph_layers.py:561-699)."""

    def __init__(self, hidden_size_nodes, hidden_size_edges, skip_config=2, num_heads=4, dp_rate=0.0, act_fn=nn.GELU):
        super().__init__()
        self.hidden_size_nodes, self.hidden_size_edges, self.num_heads = hidden_size_nodes, hidden_size_edges, num_heads
        self.hidden_size_per_head = int(hidden_size_nodes // num_heads)
        self.hidden_size_output = self.hidden_size_per_head * num_heads
        self.node_feat_layer = _Linear(hidden_size_nodes, self.hidden_size_output * 2)
        self.edge_feat_layer = _Linear(hidden_size_edges, self.hidden_size_output)
        self.edge_logits_layer = _Linear(hidden_size_edges, num_heads)
        self.skip_layer = GNNSkipConnection(hidden_size_nodes, config=skip_config, input_size=self.hidden_size_output)
        self.dropout = nn.Dropout(dp_rate)
        self.act_fn = act_fn()
        self.node_normalization = nn.LayerNorm(hidden_size_nodes)
        self.edge_normalization = nn.LayerNorm(hidden_size_edges)

    def forward(self, node_feat, edge_feat, x_indices, mask_valid, **kwargs):
        B, V = node_feat.size(0), node_feat.size(1)
        ctx = _context(kwargs, x_indices, mask_valid, V)
        own, context = self.node_feat_layer(self.node_normalization(node_feat)).chunk(2, dim=-1)
        e_in = self.edge_normalization(edge_feat)
        pair_val = self._heads(self.edge_feat_layer(e_in), B, edge_feat.size(1))         # [B, E, H, d]
        gate = torch.sigmoid(pairs_to_dense(self.edge_logits_layer(e_in), ctx.table)).permute(0, 3, 1, 2)   # [B, H, V, V]
        gate = gate * ctx.neigh.unsqueeze(1).to(gate.dtype)
        probs = gate / gate.sum(dim=-1, keepdim=True).clamp(min=1e-5)
        context = self._heads(context, B, V).permute(0, 2, 1, 3)                          # [B, H, V, d]
        attn = torch.matmul(probs, context).permute(0, 2, 1, 3) + self._pair_terms(probs, pair_val, ctx)
        comb = self.act_fn(self.dropout(own + attn.reshape(B, V, self.hidden_size_output)))
        return self.skip_layer(orig=node_feat, feat=comb)


class EdgeGNNLayer(nn.Module):
    """Nodes first (from the pairs as they are), then the pairs (from the updated nodes) (graph_layers.py:242-263)."""

    def __init__(self, edge2node_layer_func, node2edge_layer_func):
        super().__init__()
        self.node2edge_layer = node2edge_layer_func()
        self.edge2node_layer = edge2node_layer_func()

    def forward(self, node_feat, edge_feat, x_indices, mask_valid, **kwargs):
        node_feat = self.edge2node_layer(node_feat=node_feat, edge_feat=edge_feat, x_indices=x_indices, mask_valid=mask_valid, **kwargs)
        edge_feat = self.node2edge_layer(node_feat=node_feat, edge_feat=edge_feat, x_indices=x_indices, mask_valid=mask_valid, **kwargs)
        # (graph_layers.py:262 multiplies by mask_valid here; Node2EdgePlainLayer has just zeroed the invalid pairs — a 0/1
        # mask applied twice — so only a foreign node2edge layer still needs it)
        if not isinstance(self.node2edge_layer, Node2EdgePlainLayer):
            edge_feat = edge_feat * mask_valid.unsqueeze(dim=-1)
        return node_feat, edge_feat


class EdgeGNN(nn.Module):
    """Input MLPs (+ an embedding of the clamped neighbour count), `num_layers` Edge-GNN layers, output MLPs; the outputs
    of padded nodes and invalid pairs are zero (graph_layers.py:737-815)."""

    def __init__(self, c_in_nodes, c_in_edges, c_out_nodes, c_out_edges, edge_gnn_layer_func, num_layers=4, max_neighbours=-1):
        super().__init__()
        self.c_in_nodes, self.c_in_edges, self.c_out_nodes, self.c_out_edges = c_in_nodes, c_in_edges, c_out_nodes, c_out_edges
        self.layers = nn.ModuleList([edge_gnn_layer_func() for _ in range(num_layers)])
        hidden_edges = self.layers[0].node2edge_layer.hidden_size_edges
        hidden_nodes = self.layers[0].node2edge_layer.hidden_size_nodes

        def mlp_in(c_in, hidden):
            return nn.Sequential(_Linear(c_in, hidden), nn.GELU(), _Linear(hidden, hidden))

        def mlp_out(hidden, c_out):
            return nn.Sequential(nn.LayerNorm(hidden), _Linear(hidden, hidden), nn.GELU(), _Linear(hidden, c_out))

        self.input_layer_edges, self.input_layer_nodes = mlp_in(c_in_edges, hidden_edges), mlp_in(c_in_nodes, hidden_nodes)
        self.out_layer_edges, self.out_layer_nodes = mlp_out(hidden_edges, c_out_edges), mlp_out(hidden_nodes, c_out_nodes)
        if max_neighbours > 0:
            self.max_neighbours = max_neighbours
            self.node_neighbour_embed = _Linear(max_neighbours + 1, hidden_nodes)

    def forward(self, z_nodes, z_edges, length, x_indices, mask_valid, channel_padding_mask=None, binary_adjacency=None, **kwargs):
        nodes, edges = self.input_layer_nodes(z_nodes), self.input_layer_edges(z_edges)
        if binary_adjacency is not None and hasattr(self, "node_neighbour_embed"):
            degree = binary_adjacency.sum(dim=-1).long().clamp(max=self.max_neighbours)
            # a one-hot through the Linear = a column of its weight + the bias
            nodes = nodes + F.embedding(degree, self.node_neighbour_embed.weight.t()) + self.node_neighbour_embed.bias
        ctx = _GraphContext(x_indices, mask_valid, z_nodes.size(1), binary_adjacency)
        for layer in self.layers:
            nodes, edges = layer(node_feat=nodes, edge_feat=edges, x_indices=x_indices, mask_valid=mask_valid,
                                 graph_context=ctx, channel_padding_mask=channel_padding_mask, binary_adjacency=binary_adjacency)
        nodes_out = self.out_layer_nodes(nodes)
        edges_out = self.out_layer_edges(edges)
        if channel_padding_mask is not None:
            nodes_out = nodes_out * channel_padding_mask
        edges_out = torch.where(ctx.valid.unsqueeze(-1), edges_out, torch.zeros_like(edges_out))
        return nodes_out, edges_out
