"""Soak test for a HIP-graph-captured TRAINING step: NOT part of the package or of the test
suite — the acceptance experiment a graph-captured step has to pass before it returns to the package.

Two identical copies of the set-modelling flow are trained on the same data and the same encoder noise: copy A eagerly,
copy B by replaying one captured step.  Every `--check_every` steps the parameters are compared; the script prints the first
step at which they differ by more than `--tol` (relative to the parameter's scale) and the largest difference seen, so a
wrong replay is located to within `--check_every` steps instead of showing up as "stopped learning" after thousands.

    python tools/graph_train_soak.py --steps 3000                      # the acceptance run
    python tools/graph_train_soak.py --steps 1000 --optimizer sgd      # bisect: no adaptive optimiser state
    python tools/graph_train_soak.py --steps 1000 --no_clip            # bisect: no gradient clipping
    python tools/graph_train_soak.py --steps 1000 --sync_each          # bisect: never more than one replay in flight
    python tools/graph_train_soak.py --steps 1000 --capture_stream warmup   # capture on the stream the warm-up ran on

Rules the captured step follows (what round 2's removed implementation had established): encoder noise is drawn OUTSIDE
the graph into a static buffer; gradients come from torch.autograd.grad (no AccumulateGrad nodes bound to another stream);
the optimiser is `capturable` with its learning rate in a device tensor; nothing but copies into the static inputs happens
between replays.  The eager copy runs the SAME Python function, so any difference is the capture's."""
import argparse
import contextlib
import copy
import io
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from categoricalnf_amd import functional as Fn                                   # noqa: E402
from categoricalnf_amd import ops                                                # noqa: E402
from categoricalnf_amd.experiments.set_modeling import FlowSetModeling, SetShufflingDataset   # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=3000)
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--check_every", type=int, default=100)
ap.add_argument("--tol", type=float, default=1e-4)
ap.add_argument("--optimizer", default="radam", choices=["radam", "adam", "sgd"])
ap.add_argument("--no_clip", action="store_true")
ap.add_argument("--sync_each", action="store_true")
ap.add_argument("--capture_stream", default="fresh", choices=["fresh", "warmup"])
ap.add_argument("--flows", type=int, default=8)
ap.add_argument("--hidden", type=int, default=256)
args = ap.parse_args()

dev = torch.device("cuda", 0)
torch.manual_seed(0)
np.random.seed(0)
params = {"set_size": 16, "coupling_hidden_layers": 2, "coupling_hidden_size": args.hidden, "coupling_num_flows": args.flows,
          "coupling_mask_ratio": 0.5, "coupling_num_mixtures": 8,
          "categ_encoding": {"use_dequantization": False, "use_variational": False, "use_decoder": False, "num_dimensions": 4,
                             "flow_config": {"num_flows": 0}, "decoder_config": {}}}
with contextlib.redirect_stdout(io.StringIO()):
    model_a = FlowSetModeling(params, SetShufflingDataset).to(dev).train()
rng = np.random.RandomState(1)
B, S, D = args.batch, 16, 4


def draw():
    return torch.from_numpy(np.stack([rng.permutation(S) for _ in range(B)])).long().to(dev)


ln = torch.full((B,), S, dtype=torch.long, device=dev)
with contextlib.redirect_stdout(io.StringIO()):
    model_a.initialize_data_dependent([(draw(), {"length": ln}) for _ in range(4)])
model_b = copy.deepcopy(model_a)


def make_optimizer(model):
    lr = torch.tensor(7.5e-4, device=dev)
    if args.optimizer == "radam":
        return torch.optim.RAdam(model.parameters(), lr=lr, capturable=True)
    if args.optimizer == "adam":
        return torch.optim.Adam(model.parameters(), lr=lr, capturable=True)
    return torch.optim.SGD(model.parameters(), lr=1e-3)


def train_step(model, optimizer, x, noise):
    """One step; the same function runs eagerly (copy A) and under capture (copy B)."""
    z, ldj = model(x, reverse=False, length=ln, beta=1, noise=noise)
    loss = Fn.PriorNllFn.apply(z, ldj, ln, None).mean()
    plist = [p for p in model.parameters() if p.requires_grad]
    grads = torch.autograd.grad(loss, plist, allow_unused=True)
    for p, g in zip(plist, grads):
        p.grad = g
    if not args.no_clip:
        torch.nn.utils.clip_grad_norm_(plist, 0.25, foreach=True)
    optimizer.step()
    return loss.detach()


opt_a, opt_b = make_optimizer(model_a), make_optimizer(model_b)
static_x = draw()
static_noise = torch.rand(B * S, 1, D, device=dev)
gen = torch.Generator(device=dev).manual_seed(5)

# warm-up of BOTH copies with identical inputs (optimiser state, lazy caches, allocator pools), then capture copy B's step
main = torch.cuda.current_stream(dev)
side = torch.cuda.Stream(device=dev)
for _ in range(3):
    x = draw()
    static_x.copy_(x)
    static_noise.uniform_(generator=gen)
    train_step(model_a, opt_a, static_x, static_noise)
    side.wait_stream(main)                                # the inputs above are written on the main stream
    with torch.cuda.stream(side):
        ops.CAPTURING = True
        try:
            train_step(model_b, opt_b, static_x, static_noise)
        finally:
            ops.CAPTURING = False
    main.wait_stream(side)
torch.cuda.synchronize(dev)
ops.check_flags(dev, "soak warm-up")
graph = torch.cuda.CUDAGraph()
for p in model_b.parameters():
    p.grad = None
ops.CAPTURING = True
try:
    with torch.cuda.graph(graph, **({"stream": side} if args.capture_stream == "warmup" else {})):
        static_loss = train_step(model_b, opt_b, static_x, static_noise)
finally:
    ops.CAPTURING = False
# the capture executed nothing: copy A is one step behind unless it repeats the captured step's inputs once
graph.replay()
train_step(model_a, opt_a, static_x, static_noise)
torch.cuda.synchronize(dev)


def compare():
    worst, where = 0.0, None
    for (name, pa), pb in zip(model_a.named_parameters(), model_b.parameters()):
        scale = float(pa.detach().abs().max()) + 1e-12
        d = float((pa.detach() - pb.detach()).abs().max()) / scale
        if d > worst:
            worst, where = d, name
    return worst, where


first_bad, worst_seen = None, 0.0
w0, _ = compare()
print("after capture: max relative parameter difference %.3g" % w0, flush=True)
t0 = time.time()
for step in range(1, args.steps + 1):
    x = draw()
    static_x.copy_(x)
    static_noise.uniform_(generator=gen)
    loss_a = train_step(model_a, opt_a, static_x, static_noise)
    graph.replay()
    if args.sync_each:
        torch.cuda.synchronize(dev)
    if step % args.check_every == 0:
        torch.cuda.synchronize(dev)
        worst, where = compare()
        worst_seen = max(worst_seen, worst)
        print("step %6d | eager loss %.4f graph loss %.4f | max relative parameter difference %.3g (%s) | %.1f steps/s"
              % (step, float(loss_a), float(static_loss), worst, where, step / (time.time() - t0)), flush=True)
        if worst > args.tol and first_bad is None:
            first_bad = step
ops.check_flags(dev, "soak")
if first_bad is None:
    print("SOAK OK: %d replays, largest relative parameter difference %.3g" % (args.steps, worst_seen))
else:
    print("SOAK FAILED: parameters differ by more than %.1g from step <= %d on (largest %.3g)" % (args.tol, first_bad, worst_seen))
    sys.exit(1)
