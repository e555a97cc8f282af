"""Soak test of GraphedFlow (inference): thousands of replays on fresh inputs, each N-th compared with the eager pass."""
import os, sys, io, contextlib
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from categoricalnf_amd.experiments.set_modeling import FlowSetModeling, SetShufflingDataset
from categoricalnf_amd.graphs import GraphedFlow
params = {"set_size": 16, "coupling_hidden_layers": 2, "coupling_hidden_size": 256, "coupling_num_flows": 8, "coupling_mask_ratio": 0.5,
          "coupling_num_mixtures": 8, "categ_encoding": {"use_dequantization": False, "use_variational": False, "use_decoder": False,
                                                         "num_dimensions": 4, "flow_config": {"num_flows": 0}, "decoder_config": {}}}
torch.manual_seed(0); np.random.seed(0)
with contextlib.redirect_stdout(io.StringIO()):
    model = FlowSetModeling(params, SetShufflingDataset).cuda().eval()
B = 256
ln = torch.full((B,), 16, dtype=torch.long, device="cuda")
u = torch.rand(B * 16, 1, 4, device="cuda")
draw = lambda: torch.rand(B, 16, device="cuda").argsort(dim=1)
gf = GraphedFlow(model, draw(), reverse=False, length=ln, noise=u)
worst = 0.0
N = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
for i in range(N):
    x = draw()
    z, ldj = gf(x, check=False)
    if (i + 1) % 250 == 0:
        with torch.no_grad():
            ze, le = model(x, reverse=False, length=ln, noise=u)
        worst = max(worst, (z - ze).abs().max().item(), (ldj - le).abs().max().item())
print("GraphedFlow soak: %d replays, worst |graph - eager| over the sampled replays = %.3e" % (N, worst))
