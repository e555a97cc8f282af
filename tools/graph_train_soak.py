"""Soak test for a HIP-graph-captured TRAINING step (profiles/HISTORY.md section 8, item 1): NOT part of the package or of the test
suite — the acceptance experiment a graph-captured step has to pass before it returns to the package.

Two identical copies of the set-modelling flow are trained on the same data and the same encoder noise: copy A eagerly,
copy B by replaying one captured step.  Every `--check_every` steps the parameters are compared; the script prints the first
step at which they differ by more than `--tol` (relative to the parameter's scale) and the largest difference seen, so a
wrong replay is located to within `--check_every` steps instead of showing up as "stopped learning" after thousands.

    python tools/graph_train_soak.py --steps 3000                      # the acceptance run
    python tools/graph_train_soak.py --steps 1000 --optimizer sgd      # bisect: no adaptive optimiser state
    python tools/graph_train_soak.py --steps 1000 --no_clip            # bisect: no gradient clipping
    python tools/graph_train_soak.py --steps 1000 --sync_each          # bisect: never more than one replay in flight
    python tools/graph_train_soak.py --steps 600 --plain_linear    # the failure of round 2, for the record (see graphs.py)

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
from categoricalnf_amd.graphs import capture_safe_linear                         # noqa: E402
from categoricalnf_amd.experiments.set_modeling import FlowSetModeling, SetShufflingDataset   # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=3000)
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--check_every", type=int, default=100)
ap.add_argument("--tol", type=float, default=1e-4)
ap.add_argument("--optimizer", default="radam", choices=["radam", "adam", "sgd"])
ap.add_argument("--no_clip", action="store_true")
ap.add_argument("--sync_each", action="store_true")
ap.add_argument("--plain_linear", action="store_true", help="reproduce the failure: PyTorch's own Linear backward (memset node) in the captured step")
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
model_c = copy.deepcopy(model_a)          # control: a SECOND eager copy — its distance from copy A is the run-to-run noise of eager


def make_optimizer(model):
    lr = torch.tensor(7.5e-4, device=dev)
    if args.optimizer == "radam":
        return torch.optim.RAdam(model.parameters(), lr=lr, capturable=True)
    if args.optimizer == "adam":
        return torch.optim.Adam(model.parameters(), lr=lr, capturable=True)
    return torch.optim.SGD(model.parameters(), lr=1e-3)


def train_step(model, optimizer, x, noise):
    """One step; the same function runs eagerly (copies A, C) and under capture (copy B)."""
    with (contextlib.nullcontext() if args.plain_linear else capture_safe_linear()):
        return _train_step(model, optimizer, x, noise)


def _train_step(model, optimizer, x, noise):
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


opt_a, opt_b, opt_c = make_optimizer(model_a), make_optimizer(model_b), make_optimizer(model_c)
static_x = draw()
static_noise = torch.rand(B * S, 1, D, device=dev)
gen = torch.Generator(device=dev).manual_seed(5)

# copy B's step is captured by the package's GraphedTrainStep (3 warm-up steps on a side stream, then the capture, all on the
# static inputs as they are now); copies A and C take the same 3 steps eagerly so that all three start the soak equal
from categoricalnf_amd.graphs import GraphedTrainStep                          # noqa: E402
for p in model_b.parameters():
    p.grad = None
for _ in range(3):
    train_step(model_a, opt_a, static_x, static_noise)
    train_step(model_c, opt_c, static_x, static_noise)
if args.plain_linear:
    import categoricalnf_amd.graphs as _g
    _g.capture_safe_linear = contextlib.nullcontext          # the failure, for the record: PyTorch's own bias gradient under replay
graph = GraphedTrainStep(lambda: _train_step(model_b, opt_b, static_x, static_noise), dev, warmup=3, allow_memset_nodes=args.plain_linear)
print("captured graph: %s" % graph.nodes, flush=True)
static_loss = graph.static_out
# the capture executed nothing: the first replay is the step copies A and C take now
graph()
train_step(model_a, opt_a, static_x, static_noise)
train_step(model_c, opt_c, static_x, static_noise)
torch.cuda.synchronize(dev)


def compare(other):
    """Distance of `other` from copy A: the largest parameter difference relative to the tensor's own scale, over tensors
    that are not near zero (a bias that is still ~0 has no scale), and the difference of the whole parameter vectors
    relative to its norm."""
    worst, where, num, den = 0.0, None, 0.0, 0.0
    for (name, pa), pb in zip(model_a.named_parameters(), other.parameters()):
        a, b = pa.detach().double(), pb.detach().double()
        num += float(((a - b) ** 2).sum())
        den += float((a ** 2).sum())
        scale = float(a.abs().max())
        if scale < 1e-3:
            continue
        d = float((a - b).abs().max()) / scale
        if d > worst:
            worst, where = d, name
    return worst, where, (num / max(den, 1e-300)) ** 0.5


first_bad, worst_seen, worst_ctrl = None, 0.0, 0.0
w0, _, _ = compare(model_b)
print("after capture: max relative parameter difference %.3g" % w0, flush=True)
t0 = time.time()
loss_hist = []
for step in range(1, args.steps + 1):
    x = draw()
    static_x.copy_(x)
    static_noise.uniform_(generator=gen)
    loss_a = train_step(model_a, opt_a, static_x, static_noise)
    loss_c = train_step(model_c, opt_c, static_x, static_noise)
    graph()
    if args.sync_each:
        torch.cuda.synchronize(dev)
    if step % args.check_every == 0:
        torch.cuda.synchronize(dev)
        worst, where, vec = compare(model_b)
        ctrl, _, vec_c = compare(model_c)
        worst_seen, worst_ctrl = max(worst_seen, worst), max(worst_ctrl, ctrl)
        loss_hist.append((float(loss_a), float(static_loss), float(loss_c)))
        print("step %6d | loss eager %.4f graph %.4f eager twin %.4f | graph vs eager: worst tensor %.3g (%s), whole vector %.3g | "
              "eager twin vs eager: worst tensor %.3g, whole vector %.3g | %.1f steps/s"
              % (step, float(loss_a), float(static_loss), float(loss_c), worst, where, vec, ctrl, vec_c, step / (time.time() - t0)), flush=True)
        # the graph copy must stay as close to the eager copy as a second eager copy does (x4, plus the tolerance)
        if worst > max(args.tol, 4.0 * ctrl) and first_bad is None:
            first_bad = step
ops.check_flags(dev, "soak")
la = sum(h[0] for h in loss_hist[-5:]) / max(len(loss_hist[-5:]), 1)
lb = sum(h[1] for h in loss_hist[-5:]) / max(len(loss_hist[-5:]), 1)
if first_bad is None and abs(la - lb) < 0.02:
    print("SOAK OK: %d replays, largest relative parameter difference graph vs eager %.3g (eager twin vs eager: %.3g), final losses %.4f / %.4f"
          % (args.steps, worst_seen, worst_ctrl, la, lb))
else:
    print("SOAK FAILED: graph copy leaves the eager copy from step <= %s on (largest difference %.3g, eager twin %.3g; final losses %.4f / %.4f)"
          % (first_bad, worst_seen, worst_ctrl, la, lb))
    sys.exit(1)
