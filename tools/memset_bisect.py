"""Which torch op of the graph-colouring sub-network puts memset nodes into a captured training step?  Captures pieces of the
RGCN forward + backward one at a time and prints the census of each captured graph."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from categoricalnf_amd.graphs import GraphedTrainStep
from categoricalnf_amd.layers.networks.graph_layers import RGCNNet, RelationGraphConv, GNNSkipConnection
dev = torch.device("cuda", 0)
B, N, H = 128, 20, 384
torch.manual_seed(0)
adj = (torch.rand(B, N, N, device=dev) < 0.2).long()
adj = ((adj + adj.transpose(1, 2)) > 0).long()
x = torch.randn(B, N, 2, device=dev)
h = torch.randn(B, N, H, device=dev, requires_grad=True)
w = torch.randn(B, N, H, device=dev)


def census(name, fn):
    try:
        g = GraphedTrainStep(fn, dev, allow_memset_nodes=True)
        print("%-40s %s" % (name, g.nodes), flush=True)
    except Exception as e:
        print("%-40s FAILED %s" % (name, str(e)[:200]), flush=True)


def fb(mod, *inp):
    ps = [p for p in mod.parameters()]
    go = [None]
    def f():
        out = mod(*inp)
        if go[0] is None:
            go[0] = torch.randn_like(out)
        return torch.autograd.grad(out, ps, grad_outputs=go[0], allow_unused=True)      # no full-tensor .sum(): that is a two-pass reduction itself
    return f


net = RGCNNet(c_in=2, c_out=2 * (2 + 3 * 8), num_edges=1, num_layers=3, hidden_size=H).to(dev)
census("RGCNNet fwd+bwd", fb(net, x, adj))
from categoricalnf_amd.host_utils import one_hot
a1 = one_hot(adj, 2)[..., 1:]
conv = RelationGraphConv(H, H, 1).to(dev)
census("RelationGraphConv", fb(conv, h, a1))
census("LayerNorm", fb(torch.nn.LayerNorm(H).to(dev), h))
census("Linear", fb(torch.nn.Linear(H, H).to(dev), h))
skip = GNNSkipConnection(H, config=2).to(dev)
census("GNNSkipConnection", lambda: torch.autograd.grad(skip(h, h * 2), list(skip.parameters()), grad_outputs=w))
census("einsum", lambda: torch.autograd.grad(torch.einsum("bjec,bjie->bic", h.view(B, N, 1, H), a1), [h], grad_outputs=w))
census("GELU", lambda: torch.autograd.grad(torch.nn.functional.gelu(h), [h], grad_outputs=w))
census("input_layer", fb(net.input_layer, x))
census("output_layer", fb(net.output_layer, h))
census("neighbour_embed", fb(net.neighbour_embed, one_hot(a1.sum(dim=[1, 3]).clamp(max=4).long(), 5)))
census("one_hot + sum", lambda: one_hot(adj, 2)[..., 1:].sum(dim=[1, 3]))
census("x * mask", lambda: torch.autograd.grad(h * w[..., :1], [h], grad_outputs=w))
