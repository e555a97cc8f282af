"""Training step of the reference's default set-modelling flow (8 flow steps, Transformer sub-network hidden 256 x 2
layers, D=4, K=8, |S|=16): forward + NLL + backward + RAdam.  Run under `rocprofv3 --kernel-trace --stats` to see how
the step splits between this library's kernels (namespace cnf::) and the PyTorch-ROCm sub-networks.
    python tools/bench_train_step.py <batch> <steps> [flat | graph]
`graph`: the whole step captured once by categoricalnf_amd.graphs.GraphedTrainStep and replayed (round 3); its line also carries
the census of the captured graph's nodes.  CNF_FUSE_TRAINING=0: one autograd Function per layer and a separate NLL assembly
(the training path of rounds 1-3) instead of the fused groups of round 4."""
import os, sys, time, io, contextlib
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from categoricalnf_amd.experiments.set_modeling import FlowSetModeling, SetShufflingDataset
from categoricalnf_amd import functional as Fn, ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
params = {"set_size": 16, "coupling_hidden_layers": 2, "coupling_hidden_size": 256, "coupling_num_flows": 8, "coupling_mask_ratio": 0.5,
          "coupling_num_mixtures": 8, "categ_encoding": {"use_dequantization": False, "use_variational": False, "use_decoder": False,
                                                         "num_dimensions": 4, "flow_config": {"num_flows": 0}, "decoder_config": {}}}
torch.manual_seed(0)
with contextlib.redirect_stdout(io.StringIO()):
    model = FlowSetModeling(params, SetShufflingDataset).cuda().train()
rng = np.random.RandomState(0)
draw = lambda n: torch.from_numpy(np.stack([rng.permutation(16) for _ in range(n)])).long().cuda()
ln = torch.full((B,), 16, dtype=torch.long, device="cuda")
with contextlib.redirect_stdout(io.StringIO()):
    model.initialize_data_dependent([(draw(B), {"length": ln}) for _ in range(2)])
FLAT = len(sys.argv) > 3 and sys.argv[3] == "flat"             # optimiser / clipping / zero_grad on one flat buffer
flat = None
if FLAT:
    from categoricalnf_amd.host_utils import FlatParameters
    flat = FlatParameters(model)
opt = torch.optim.RAdam(flat.parameters() if FLAT else model.parameters(), lr=7.5e-4)        # the reference's default optimiser
xs = [draw(B) for _ in range(4)]

def loss_of(x, **kw):
    if ops.FUSE_TRAINING:                        # the NLL assembly inside the last coupling layer's Function
        return model.nll_loss(x, length=ln, beta=1, **kw)[2].mean()
    z, ldj = model(x, reverse=False, length=ln, beta=1, **kw)
    return Fn.PriorNllFn.apply(z, ldj, ln, None).mean()


def step(i):
    loss = loss_of(xs[i % 4])
    if FLAT:
        flat.zero_grad()
    else:
        opt.zero_grad(set_to_none=True)
    loss.backward()
    torch.nn.utils.clip_grad_norm_(flat.parameters() if FLAT else model.parameters(), 0.25)
    opt.step()
    return loss

GRAPH = len(sys.argv) > 3 and sys.argv[3] == "graph"
if GRAPH:
    from categoricalnf_amd.graphs import GraphedTrainStep
    opt = torch.optim.RAdam(model.parameters(), lr=torch.tensor(7.5e-4, device="cuda"), capturable=True)
    static_x = xs[0].clone()
    static_noise = torch.rand(B * 16, 1, 4, device="cuda")
    plist = [p for p in model.parameters() if p.requires_grad]

    def graph_step():
        loss = loss_of(static_x, noise=static_noise)
        for p, g in zip(plist, torch.autograd.grad(loss, plist, allow_unused=True)):
            p.grad = g
        torch.nn.utils.clip_grad_norm_(plist, 0.25, foreach=True)
        opt.step()
        return loss.detach()
    for p in plist:
        p.grad = None
    graphed = GraphedTrainStep(graph_step, torch.device("cuda", 0))

    def step(i):                                   # what a training loop does between replays: new data and noise into the static inputs
        static_x.copy_(xs[i % 4], non_blocking=True)
        static_noise.uniform_()
        return graphed()

for i in range(5):
    step(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(steps):
    loss = step(i)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print("batch %d %s%s: %.2f ms / training step (%.0f sets/s), loss %.4f%s" % (
    B, "HIP graph replay" if GRAPH else ("eager (flat optimiser)" if FLAT else "eager"), "" if ops.FUSE_TRAINING else " [one Function per layer]",
    dt * 1e3, B / dt, float(loss.detach()), (", graph nodes %s" % (graphed.nodes,)) if GRAPH else ""))
