"""Host cost of a whole differentiable FLOW pass against the GPU time of its kernels, at the benchmark shape: L flow steps of
[ActNorm, 1x1 conv, affine coupling] with the coupling network replaced by a fixed tensor (so that every kernel of the pass is this
library's), forward + NLL + backward through torch.autograd with the fused training groups (default) or one Function per layer
(CNF_FUSE_TRAINING=0).  Prints the wall time per pass; run it under `rocprofv3 --kernel-trace --stats` and divide the summed kernel
time by the printed number of passes for the kernels' share (tools/refresh_profiles.sh does, -> profiles/r04_flow_autograd_overhead.txt).
    python tools/flow_autograd_overhead.py [--steps 8] [--passes 60] [--B 16384]"""
import argparse, contextlib, io, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from categoricalnf_amd import ops
from categoricalnf_amd.layers.flows.activation_normalization import ActNormFlow
from categoricalnf_amd.layers.flows.coupling_layer import CouplingLayer
from categoricalnf_amd.layers.flows.flow_model import FlowModel
from categoricalnf_amd.layers.flows.permutation_layers import InvertibleConv
ap = argparse.ArgumentParser()
ap.add_argument("--single_thread", action="store_true", help="torch.autograd.set_multithreading_enabled(False), as the drivers set it")
ap.add_argument("--steps", type=int, default=8); ap.add_argument("--passes", type=int, default=60); ap.add_argument("--B", type=int, default=16384)
args = ap.parse_args()
if args.single_thread:
    torch.autograd.set_multithreading_enabled(False)
dev = torch.device("cuda:0")
B, N, D = args.B, 64, 6
g = torch.Generator(device=dev).manual_seed(0)


class Fixed(torch.nn.Module):
    """stands in for the coupling network: a parameter-scaled fixed tensor (differentiable, one elementwise kernel)"""
    def __init__(self):
        super().__init__()
        self.out = 0.5 * torch.randn(B, N, 2 * D, generator=g, device=dev)
        self.gain = torch.nn.Parameter(torch.ones((), device=dev))

    def forward(self, x, **kw):
        return self.out * self.gain


layers = []
for i in range(args.steps):
    layers += [ActNormFlow(D), InvertibleConv(D), CouplingLayer(c_in=D, mask=CouplingLayer.create_channel_mask(D), model_func=lambda c_out: Fixed())]
with contextlib.redirect_stdout(io.StringIO()):
    flow = FlowModel(layers).to(dev).train()
for m in flow.modules():
    if isinstance(m, ActNormFlow):
        m.data_init = False
zs = [torch.randn(B, N, D, generator=g, device=dev) for _ in range(2)]
ln = torch.full((B,), N, dtype=torch.long, device=dev)
plist = [p for p in flow.parameters() if p.requires_grad]


def one(i):
    z = zs[i % 2].detach().requires_grad_(True)
    nll = flow.nll_loss(z, length=ln)[2]
    torch.autograd.grad(nll.mean(), plist + [z], allow_unused=True)


for i in range(6):
    one(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(args.passes):
    one(i)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("flow of %d x [ActNorm, 1x1 conv, affine coupling] at B=%d N=%d D=%d, %s: %.1f us wall per forward + NLL + backward pass "
      "(host enqueue %.1f us), %d timed passes (+ 6 warm-up passes)"
      % (args.steps, B, N, D, ("fused training groups" if ops.FUSE_TRAINING else "one Function per layer") + (", single-threaded autograd engine" if args.single_thread else ""), (t2 - t0) / args.passes * 1e6,
         (t1 - t0) / args.passes * 1e6, args.passes), flush=True)
