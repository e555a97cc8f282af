"""bits/dim golden for set shuffling on a model trained to the sharp-mixture regime (VERDICT r1 next #4b).

The round-1 golden (gen_set_shuffling_golden.py) trains 4000 CPU iterations with the reference and stops at 3.59 bpd.
This one takes a checkpoint that the drop-in's own driver trained on an MI355X,

    python -m categoricalnf_amd.experiments.run_set_modeling --dataset shuffling --max_iterations 50000 --batch_size 256 \
        --coupling_hidden_size 64 --coupling_hidden_layers 2 --coupling_num_flows 4 --eval_freq 10000 --save_freq 100000 \
        --checkpoint_path gpurun_out/ckpt_shuffle_small --learning_rate 1e-3 --lr_decay_factor 0.99995      (2.94 bpd)

loads it into the REFERENCE's FlowSetModeling (reference checkpoint format, reference parameter names) and evaluates it
with the REFERENCE on the CPU: validation bits/dim on the 32768 fixed sets, and for the first 256 sets the injected
uniform noise, z, ldj, per-sample NLL and the decoded sets.  Runs only in the build container:

    PYTHONPATH=/root/reference MPLBACKEND=Agg PYTHONDONTWRITEBYTECODE=1 \
        python oracle/gen_set_shuffling_trained_golden.py gpurun_out/ckpt_shuffle_small/checkpoint_0050000.tar
"""
import contextlib
import io
import json
import os
import sys

import numpy as np
import torch

REF = os.environ.get("CNF_REFERENCE", "/root/reference")
sys.path.insert(0, REF)
os.environ.setdefault("MPLBACKEND", "Agg")
with contextlib.redirect_stdout(io.StringIO()):
    from experiments.set_modeling.flow_model import FlowSetModeling
    from experiments.set_modeling.datasets.set_shuffling import SetShufflingDataset, calc_optimum
    from layers.flows.distributions import LogisticDistribution

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "set_shuffling_trained.npz")
SET_SIZE, D, HIDDEN, TLAYERS, FLOWS, K = 16, 4, 64, 2, 4, 8


def main(ckpt):
    torch.set_num_threads(8)
    params = {"set_size": SET_SIZE, "coupling_hidden_layers": TLAYERS, "coupling_hidden_size": HIDDEN,
              "coupling_num_flows": FLOWS, "coupling_mask_ratio": 0.5, "coupling_num_mixtures": K,
              "categ_encoding": {"use_dequantization": False, "use_variational": False, "use_decoder": False,
                                 "num_dimensions": D, "flow_config": {"num_flows": 0, "hidden_layers": 2, "hidden_size": 128},
                                 "decoder_config": {"num_layers": 1, "hidden_size": 64}}}
    with contextlib.redirect_stdout(io.StringIO()):
        model = FlowSetModeling(params, SetShufflingDataset)
    blob = torch.load(ckpt, map_location="cpu", weights_only=False)
    model.load_state_dict(blob["model_state_dict"], strict=True)
    model.eval()
    prior = LogisticDistribution()
    length = lambda n: torch.full((n,), SET_SIZE, dtype=torch.long)

    def nll_of(x):
        z, ldj = model(x, reverse=False, length=length(x.size(0)), beta=1)
        neglog = -prior.log_prob(z).sum(dim=[1, 2])
        return (-ldj) / SET_SIZE + neglog / SET_SIZE, z, ldj

    val = torch.from_numpy(SetShufflingDataset(SET_SIZE, train=False, val=True).shuffle_set).long()
    torch.manual_seed(2025)
    total = 0.0
    with torch.no_grad():
        for i in range(0, val.size(0), 2048):
            total += float(nll_of(val[i:i + 2048])[0].double().sum())
        val_nll = total / val.size(0)
        x256 = val[:256]
        torch.manual_seed(4242)
        u = torch.rand(256 * SET_SIZE, 1, D)
        torch.manual_seed(4242)
        nll256, z256, ldj256 = nll_of(x256)
        dec256, _ = model(z256, reverse=True, length=length(256))
    val_bpd = float(np.log2(np.e) * val_nll)
    print("reference validation bpd %.4f (optimum %.4f; drop-in driver reported %s), decode accuracy %.4f"
          % (val_bpd, calc_optimum(SET_SIZE), blob.get("evaluation_dict", {}), float((dec256 == x256).float().mean())))
    flat = {"sd_" + k: v.detach().numpy() for k, v in model.state_dict().items()}
    meta = dict(set_size=SET_SIZE, D=D, hidden=HIDDEN, transformer_layers=TLAYERS, flows=FLOWS, K=K, iters=int(blob.get("iteration", -1)),
                trained_with="categoricalnf_amd.experiments.run_set_modeling on MI355X", val_nll=val_nll, val_bpd=val_bpd,
                optimum_bpd=float(calc_optimum(SET_SIZE)), val_seed=123, infos=[l.info() for l in model.flow_layers])
    flat.update(meta=np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8), x256=x256.numpy(), u256=u.numpy(),
                z256=z256.numpy(), ldj256=ldj256.numpy(), nll256=nll256.numpy(), dec256=dec256.numpy())
    np.savez_compressed(OUT, **flat)
    print("wrote %s (%.0f kB)" % (OUT, os.path.getsize(OUT) / 1e3))


if __name__ == "__main__":
    main(sys.argv[1])
