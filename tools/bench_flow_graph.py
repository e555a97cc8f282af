"""Whole-flow latency, eager vs HIP-graph replay: the reference's default set-modelling flow (8 flow steps,
Transformer sub-network hidden 256 x 2 layers, D=4, K=8, |S|=16) forward (log-likelihood) and reverse (sampling)."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from categoricalnf_amd.experiments.set_modeling import FlowSetModeling, SetShufflingDataset
from categoricalnf_amd.graphs import GraphedFlow
import io, contextlib
params = {"set_size": 16, "coupling_hidden_layers": 2, "coupling_hidden_size": 256, "coupling_num_flows": 8, "coupling_mask_ratio": 0.5,
          "coupling_num_mixtures": 8, "categ_encoding": {"use_dequantization": False, "use_variational": False, "use_decoder": False,
                                                         "num_dimensions": 4, "flow_config": {"num_flows": 0}, "decoder_config": {}}}
with contextlib.redirect_stdout(io.StringIO()):
    model = FlowSetModeling(params, SetShufflingDataset).cuda().eval()
rng = np.random.RandomState(0)
print("%6s | %12s %12s %8s | %12s %12s %8s" % ("batch", "fwd eager us", "fwd graph us", "speedup", "rev eager us", "rev graph us", "speedup"))
for B in (64, 256, 1024, 4096):
    x = torch.from_numpy(np.stack([rng.permutation(16) for _ in range(B)])).long().cuda()
    ln = torch.full((B,), 16, dtype=torch.long, device="cuda")

    def t(fn, reps=20):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e6
    with torch.no_grad():
        z, _ = model(x, reverse=False, length=ln)
        fe = t(lambda: model(x, reverse=False, length=ln))
        re_ = t(lambda: model(z, reverse=True, length=ln))
    gf = GraphedFlow(model, x, reverse=False, length=ln)
    gr = GraphedFlow(model, z, reverse=True, length=ln)
    fg = t(lambda: gf(x, check=False))
    rg = t(lambda: gr(z, check=False))
    print("%6d | %12.0f %12.0f %7.2fx | %12.0f %12.0f %7.2fx" % (B, fe, fg, fe / fg, re_, rg, re_ / rg), flush=True)
