"""Diagnostic for the captured TRAINING step (profiles/HISTORY.md section 8 item 1): with the parameters held fixed, the loss and
every parameter gradient of one step replayed from a HIP graph against the same step run eagerly, on several fresh
inputs.  Any difference is the capture's (no optimiser in the loop, so nothing can drift).  Prints the parameters whose
gradients differ, largest first, and the first flow layer whose OUTPUT differs between the two paths.

    python tools/graph_grad_diag.py [--batch 64] [--flows 8] [--hidden 256] [--rounds 4]"""
import argparse, contextlib, copy, io, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from categoricalnf_amd import functional as Fn, ops
from categoricalnf_amd.experiments.set_modeling import FlowSetModeling, SetShufflingDataset

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--flows", type=int, default=8)
ap.add_argument("--hidden", type=int, default=256)
ap.add_argument("--rounds", type=int, default=4)
ap.add_argument("--safe_linear", action="store_true", help="graphs.capture_safe_linear around both paths (the workaround)")
args = ap.parse_args()
dev = torch.device("cuda", 0)
torch.manual_seed(0); np.random.seed(0)
params = {"set_size": 16, "coupling_hidden_layers": 2, "coupling_hidden_size": args.hidden, "coupling_num_flows": args.flows,
          "coupling_mask_ratio": 0.5, "coupling_num_mixtures": 8,
          "categ_encoding": {"use_dequantization": False, "use_variational": False, "use_decoder": False, "num_dimensions": 4,
                             "flow_config": {"num_flows": 0}, "decoder_config": {}}}
with contextlib.redirect_stdout(io.StringIO()):
    model_a = FlowSetModeling(params, SetShufflingDataset).to(dev).train()
rng = np.random.RandomState(1)
B, S, D = args.batch, 16, 4
draw = lambda: torch.from_numpy(np.stack([rng.permutation(S) for _ in range(B)])).long().to(dev)
ln = torch.full((B,), S, dtype=torch.long, device=dev)
with contextlib.redirect_stdout(io.StringIO()):
    model_a.initialize_data_dependent([(draw(), {"length": ln}) for _ in range(4)])
for p in model_a.parameters():                       # away from the symmetric start (zero-initialised last layers)
    p.data = p.data + 0.02 * torch.randn_like(p)
model_b = copy.deepcopy(model_a)
names = [n for n, _ in model_a.named_parameters()]


def grads_of(model, x, noise, taps=None):
    from categoricalnf_amd.graphs import capture_safe_linear
    with (capture_safe_linear() if args.safe_linear else contextlib.nullcontext()):
        return _grads_of(model, x, noise, taps)


def _grads_of(model, x, noise, taps=None):
    hooks = []
    if taps is not None:
        for i, layer in enumerate(model.flow_layers):
            hooks.append(layer.register_forward_hook(lambda m, inp, out, i=i: taps.__setitem__(i, (out[0].detach().clone(), out[1].detach().clone()))))
    z, ldj = model(x, reverse=False, length=ln, beta=1, noise=noise)
    for h in hooks:
        h.remove()
    loss = Fn.PriorNllFn.apply(z, ldj, ln, None).mean()
    plist = [p for p in model.parameters() if p.requires_grad]
    g = torch.autograd.grad(loss, plist, allow_unused=True)
    return loss.detach(), [None if t is None else t.detach() for t in g], z.detach(), ldj.detach()


static_x, static_noise = draw(), torch.rand(B * S, 1, D, device=dev)
gen = torch.Generator(device=dev).manual_seed(5)
main, side = torch.cuda.current_stream(dev), torch.cuda.Stream(device=dev)
for _ in range(3):
    side.wait_stream(main)
    with torch.cuda.stream(side):
        ops.CAPTURING = True
        try:
            grads_of(model_b, static_x, static_noise)
        finally:
            ops.CAPTURING = False
    main.wait_stream(side)
torch.cuda.synchronize(dev)
graph = torch.cuda.CUDAGraph()
taps_b = {}
ops.CAPTURING = True
try:
    with torch.cuda.graph(graph):
        loss_b, grads_b, z_b, ldj_b = grads_of(model_b, static_x, static_noise, taps_b)
finally:
    ops.CAPTURING = False
print("captured; parameters: %d tensors, flow layers: %d" % (len(names), len(model_b.flow_layers)), flush=True)
for r in range(args.rounds):
    static_x.copy_(draw())
    static_noise.uniform_(generator=gen)
    taps_a = {}
    loss_a, grads_a, z_a, ldj_a = grads_of(model_a, static_x, static_noise, taps_a)
    graph.replay()
    torch.cuda.synchronize(dev)
    rows = []
    for n, ga, gb in zip(names, grads_a, grads_b):
        if (ga is None) != (gb is None):
            rows.append((float("inf"), n, "one side has no gradient"))
            continue
        if ga is None:
            continue
        scale = float(ga.abs().max()) + 1e-20
        d = float((ga - gb).abs().max()) / scale
        rows.append((d, n, "max |g| %.3e" % scale))
    rows.sort(key=lambda t: -t[0])
    bad = [t for t in rows if t[0] > 1e-3]
    first_layer = None
    for i in sorted(taps_a):
        dz = float((taps_a[i][0] - taps_b[i][0]).abs().max())
        dl = float((taps_a[i][1] - taps_b[i][1]).abs().max())
        if dz > 1e-4 or dl > 1e-3:
            first_layer = (i, type(model_a.flow_layers[i]).__name__, dz, dl)
            break
    print("round %d: loss eager %.6f graph %.6f | z diff %.2e ldj diff %.2e | gradients differing by > 1e-3 of their scale: %d of %d | first layer whose output differs: %s"
          % (r, float(loss_a), float(loss_b), float((z_a - z_b).abs().max()), float((ldj_a - ldj_b).abs().max()), len(bad), len(rows), first_layer), flush=True)
    for d, n, note in rows[:6]:
        print("    %-70s rel diff %.3e  (%s)" % (n, d, note))
ops.check_flags(dev, "diag")
