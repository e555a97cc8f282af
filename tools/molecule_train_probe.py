"""configs[4] end to end on the device at its real sizes: the three-stage molecule GraphCNF (38 nodes, 703 node pairs,
D = 6 / 2, K = 16 / 8, 9 node types, flows 4,6,6) on this package's HIP layers, its stage-1 sub-network the RGCN and its
stage-2 / 3 sub-network the Edge-GNN (layers/networks/edge_gnn.py: the reference's architecture, graphCNF.py:125-160, hidden
sizes 256 / 128, 4 layers), on the synthetic molecule-like graphs of experiments/molecule_data.py:
data-dependent init, `--steps` training steps (Adam, gradient clipping), the per-node NLL curve, steps per second, and one
sampling pass whose graphs are checked for shape / symmetry / padding.

    python tools/molecule_train_probe.py [--steps 60] [--batch 64] [--flows 4,6,6]"""
import argparse, contextlib, io, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from categoricalnf_amd import functional as Fn
from categoricalnf_amd.experiments.molecule_data import MAX_NODES, NUM_NODE_TYPES, TYPE_PROBS, random_molecule_like_graph
from categoricalnf_amd.experiments.molecule_generation import GraphCNF
from categoricalnf_amd.host_utils import create_channel_mask

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=60)
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--flows", default="4,6,6")
ap.add_argument("--hidden_nodes", type=int, default=256)
ap.add_argument("--hidden_edges", type=int, default=128)
ap.add_argument("--layers", type=int, default=4)
ap.add_argument("--graphs", type=int, default=4096)
ap.add_argument("--graph_step", action="store_true", help="capture the training step in a hipGraph (categoricalnf_amd.graphs.GraphedTraining)")
args = ap.parse_args()
dev = torch.device("cuda", 0)
torch.manual_seed(0); np.random.seed(0)
rng = np.random.RandomState(0)
nodes = np.zeros((args.graphs, MAX_NODES), dtype=np.int64)
adjacency = np.zeros((args.graphs, MAX_NODES, MAX_NODES), dtype=np.int64)
length = np.zeros(args.graphs, dtype=np.int64)
for g in range(args.graphs):
    n = rng.randint(8, MAX_NODES + 1)
    nodes[g, :n], adjacency[g, :n, :n] = random_molecule_like_graph(rng, n)
    length[g] = n
edge_counts = np.bincount(adjacency[adjacency > 0].ravel(), minlength=4)[1:4].astype(np.float64)


class Molecules:
    max_num_nodes = staticmethod(lambda: MAX_NODES)
    num_node_types = staticmethod(lambda: NUM_NODE_TYPES)
    num_edge_types = staticmethod(lambda: 3)
    num_max_neighbours = staticmethod(lambda: 5)
    get_node_prior = staticmethod(lambda data_root=None: (TYPE_PROBS / TYPE_PROBS.sum()).astype(np.float32))
    get_edge_prior = staticmethod(lambda data_root=None: ((edge_counts + 1) / (edge_counts + 1).sum()).astype(np.float32))


enc = lambda d: {"use_dequantization": False, "use_variational": False, "use_decoder": False, "num_dimensions": d,
                 "flow_config": {"num_flows": 0, "hidden_layers": 2, "hidden_size": 128}, "decoder_config": {"num_layers": 1, "hidden_size": 64}}
params = {"categ_encoding_nodes": enc(6), "categ_encoding_edges": enc(2), "encoding_virtual_num_flows": 0,
          "coupling_hidden_size_nodes": args.hidden_nodes, "coupling_hidden_size_edges": args.hidden_edges, "coupling_num_flows": args.flows,
          "coupling_hidden_layers": args.layers, "coupling_num_mixtures_nodes": 16, "coupling_num_mixtures_edges": 8,
          "coupling_mask_ratio": 0.5, "coupling_dropout": 0.0}
with contextlib.redirect_stdout(io.StringIO()):
    model = GraphCNF(params, Molecules).to(dev)
n_par = sum(p.numel() for p in model.parameters())


def batch(idx):
    return (torch.from_numpy(nodes[idx]).to(dev), torch.from_numpy(adjacency[idx]).to(dev), torch.from_numpy(length[idx]).to(dev))


def nll_per_node(x, adj, ln, beta=1.0):
    z, ldj = model(x, adjacency=adj, reverse=False, length=ln, beta=beta)
    pad = create_channel_mask(ln, max_len=x.size(1))
    return Fn.PriorNllFn.apply(z, ldj, ln.float(), pad)


with contextlib.redirect_stdout(io.StringIO()):
    init = []
    for _ in range(4):
        x, adj, ln = batch(rng.randint(0, args.graphs, size=args.batch))
        init.append((x, {"adjacency": adj, "length": ln}))
    model.initialize_data_dependent(init)
model.train()
opt = torch.optim.Adam(model.parameters(), lr=5e-4, fused=True)
graphed = None
if args.graph_step:
    # shapes are static (every batch is padded to MAX_NODES): the whole step — ~10 000 kernel launches of the Edge-GNN stages,
    # their backward, clipping, Adam — becomes one hipGraph replay
    from categoricalnf_amd.graphs import GraphedTraining
    sx, sadj, sln = (t.clone() for t in batch(rng.randint(0, args.graphs, size=args.batch)))
    t_cap = time.time()
    graphed = GraphedTraining(model, lambda: nll_per_node(sx, sadj, sln).mean(), dev, 0.25, lr=5e-4, eager_optimizer=None,
                              optimizer_cls=torch.optim.Adam, allow_unverified=True)
    print("captured training step: hipGraph nodes %s, %.1f s" % (graphed.nodes, time.time() - t_cap), flush=True)
print("GraphCNF at configs[4]'s sizes: flows %s, %d layers, %.1f M parameters, batch %d x %d nodes / %d pairs" % (
    args.flows, len(model.step1_flows) + len(model.step2_flows) + len(model.step3_flows), n_par / 1e6, args.batch, MAX_NODES, MAX_NODES * (MAX_NODES - 1) // 2), flush=True)
hist, t0 = [], None
for step in range(1, args.steps + 1):
    if step == 6:
        torch.cuda.synchronize(); t0 = time.time()
    x, adj, ln = batch(rng.randint(0, args.graphs, size=args.batch))
    if graphed is not None:
        sx.copy_(x); sadj.copy_(adj); sln.copy_(ln)
        loss = graphed()
    else:
        loss = nll_per_node(x, adj, ln).mean()
        opt.zero_grad(set_to_none=True)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 0.25)
        opt.step()
    hist.append(float(loss))
    if step % 10 == 0 or step == 1:
        print("step %4d | NLL per node %.4f (%.3f bits)" % (step, hist[-1], hist[-1] * np.log2(np.e)), flush=True)
torch.cuda.synchronize()
rate = (args.steps - 5) / (time.time() - t0) if t0 else float("nan")
first, last = float(np.mean(hist[:5])), float(np.mean(hist[-5:]))
print("training: NLL per node %.4f -> %.4f over %d steps, %.2f steps/s" % (first, last, args.steps, rate), flush=True)
assert np.isfinite(hist).all() and last < first, "the loss did not fall"
if graphed is not None:
    graphed.drop_weight_caches()
model.eval()
with torch.no_grad():
    x, adj, ln = batch(np.arange(32))
    val = nll_per_node(x, adj, ln)
    z = model.prior_distribution.sample(shape=(32, MAX_NODES, 6), device=dev) * create_channel_mask(ln, max_len=MAX_NODES)
    (s_nodes, s_adj), _ = model(z, reverse=True, length=ln)
valid = torch.arange(MAX_NODES, device=dev)[None, :] < ln[:, None]
assert s_nodes.shape == (32, MAX_NODES) and s_adj.shape == (32, MAX_NODES, MAX_NODES)
assert int(s_nodes.min()) >= 0 and int(s_nodes.max()) < NUM_NODE_TYPES and int(s_adj.min()) >= 0 and int(s_adj.max()) <= 3
assert torch.equal(s_adj, s_adj.transpose(1, 2)) and int(s_adj.diagonal(dim1=1, dim2=2).abs().sum()) == 0
assert int((s_adj * (~(valid[:, :, None] & valid[:, None, :])).long()).abs().sum()) == 0, "bonds between padding nodes"
print("evaluation NLL per node %.4f; sampled 32 graphs: node types 0..%d, bond orders 0..%d, symmetric, no bonds on padding, %.1f bonds per graph"
      % (float(val.mean()), int(s_nodes.max()), int(s_adj.max()), float((s_adj > 0).sum()) / 64), flush=True)
print("MOLECULE PROBE OK")
