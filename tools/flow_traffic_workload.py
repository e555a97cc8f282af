"""Workload for the whole-flow HBM traffic table: an evaluation pass (FlowModel.nll) of a
set-modelling style flow — 8 x (ActNorm, 1x1 conv, mixture-CDF coupling) at configs[1]'s shape B=16384, N=16, D=4, K=8 —
with the coupling sub-networks replaced by a stub that hands back a pre-drawn nn_out (so that only this library's
kernels move data), REP passes, layer fusion on (default) or off (argv[1] == "unfused")."""
import os, sys
import torch
import torch.nn as nn
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from categoricalnf_amd import ops
from categoricalnf_amd.layers.flows.flow_model import FlowModel
from categoricalnf_amd.layers.flows.activation_normalization import ActNormFlow
from categoricalnf_amd.layers.flows.permutation_layers import InvertibleConv
from categoricalnf_amd.layers.flows.mixture_cdf_layer import MixtureCDFCoupling
from categoricalnf_amd.layers.flows.coupling_layer import CouplingLayer
import contextlib, io

mode = sys.argv[1] if len(sys.argv) > 1 else "fused"
ops.FUSE_LAYERS = mode != "unfused"
B, N, D, K, STEPS, REP = 16384, 16, 4, 8, 8, 6
dev = torch.device("cuda:0")
torch.manual_seed(0)


class Stub(nn.Module):
    def __init__(self, c_out):
        super().__init__()
        self.out = 0.5 * torch.randn(B, N, c_out, device=dev)

    def forward(self, x=None, **kw):
        return self.out


layers = []
mask = CouplingLayer.create_channel_mask(D)
for i in range(STEPS):
    layers += [ActNormFlow(D), InvertibleConv(D),
               MixtureCDFCoupling(D, mask if i % 2 == 0 else 1 - mask, model_func=lambda c_out: Stub(c_out), num_mixtures=K)]
with contextlib.redirect_stdout(io.StringIO()):
    model = FlowModel(layers).to(dev).eval()
z = torch.randn(B, N, D, device=dev)
ln = torch.full((B,), N, dtype=torch.long, device=dev)
with torch.no_grad():
    for _ in range(REP):
        if mode == "unfused":
            zo, ldj = model(z, length=ln)
            ops.prior_nll(zo, ldj, ln)
        else:
            model.nll(z, length=ln)
torch.cuda.synchronize()
print("done", mode)
