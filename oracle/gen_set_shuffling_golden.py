"""bits/dim golden for set shuffling (BASELINE.json: "bits/dim within +-0.01 of reference").

Runs only in the build container (needs /root/reference).  Trains a REDUCED FlowSetModeling with the
reference's own modules on the CPU (own loop: the reference's general/train.py needs tensorboard), then
evaluates it with the reference on the deterministic validation set (np.random.seed(123), 32768
permutations of 16 — experiments/set_modeling/datasets/set_shuffling.py:23-26) and stores

  * the trained state_dict (reference parameter names),
  * the reference's validation NLL / bits-per-dim,
  * for the first 256 validation sets: the injected uniform noise, z, ldj and per-sample NLL.

The committed tests/golden/set_shuffling_model.npz (meta: iters 6000, val_bpd 3.5907) was made with

    CNF_TRAIN_ITERS=6000 PYTHONPATH=/root/reference MPLBACKEND=Agg PYTHONDONTWRITEBYTECODE=1 python oracle/gen_set_shuffling_golden.py

(the default below, 4000 iterations, is the quicker run of the first draft).  The training loop is a multi-threaded CPU run and
not bit-reproducible: a re-run gives another state_dict of the same quality, and the fixture stays a valid pin because it
stores the state_dict TOGETHER WITH the reference's outputs on it.
"""
import contextlib
import io
import json
import os
import sys
import time

import numpy as np
import torch

REF = os.environ.get("CNF_REFERENCE", "/root/reference")
sys.path.insert(0, REF)
os.environ.setdefault("MPLBACKEND", "Agg")
with contextlib.redirect_stdout(io.StringIO()):
    from experiments.set_modeling.flow_model import FlowSetModeling
    from experiments.set_modeling.datasets.set_shuffling import SetShufflingDataset, calc_optimum
    from layers.flows.distributions import LogisticDistribution

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "set_shuffling_model.npz")

SET_SIZE, D, HIDDEN, TLAYERS, FLOWS, K = 16, 4, 32, 1, 4, 8
ITERS, BATCH, LR = int(os.environ.get("CNF_TRAIN_ITERS", "4000")), 128, 7.5e-4


def model_params():
    return {"set_size": SET_SIZE, "coupling_hidden_layers": TLAYERS, "coupling_hidden_size": HIDDEN,
            "coupling_num_flows": FLOWS, "coupling_mask_ratio": 0.5, "coupling_num_mixtures": K,
            "categ_encoding": {"use_dequantization": False, "use_variational": False, "use_decoder": False,
                               "num_dimensions": D, "flow_config": {"num_flows": 0, "hidden_layers": 2, "hidden_size": 128},
                               "decoder_config": {"num_layers": 1, "hidden_size": 64}}}


def nll_of(model, prior, x, length):
    z, ldj = model(x, reverse=False, length=length, beta=1)
    neglog = -prior.log_prob(z).sum(dim=[1, 2])
    return (-ldj) / length.float() + neglog / length.float(), z, ldj


def main():
    torch.manual_seed(42)
    np.random.seed(42)
    torch.set_num_threads(8)
    with contextlib.redirect_stdout(io.StringIO()):
        model = FlowSetModeling(model_params(), SetShufflingDataset)
    prior = LogisticDistribution()
    rng = np.random.RandomState(7)
    draw = lambda n: torch.from_numpy(np.stack([rng.permutation(SET_SIZE) for _ in range(n)])).long()
    length = lambda n: torch.full((n,), SET_SIZE, dtype=torch.long)

    # data-dependent init of the ActNorm layers (reference: 16 batches)
    with contextlib.redirect_stdout(io.StringIO()):
        model.initialize_data_dependent([(draw(BATCH), {"length": length(BATCH)}) for _ in range(16)])

    opt = torch.optim.Adam(model.parameters(), lr=LR)
    sched = torch.optim.lr_scheduler.StepLR(opt, step_size=1, gamma=0.99975 ** 1)
    model.train()
    t0 = time.time()
    for it in range(ITERS):
        x = draw(BATCH)
        nll, _, _ = nll_of(model, prior, x, length(BATCH))
        loss = nll.mean()
        opt.zero_grad()
        loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 0.25)
        opt.step()
        sched.step()
        if it % 250 == 0 or it == ITERS - 1:
            print("iter %5d  bpd %.4f  (%.0fs)" % (it, loss.item() * np.log2(np.e), time.time() - t0), flush=True)

    # evaluation with the reference on the deterministic validation set
    model.eval()
    val = torch.from_numpy(SetShufflingDataset(SET_SIZE, train=False, val=True).shuffle_set).long()
    torch.manual_seed(2024)
    total, z256 = 0.0, None
    with torch.no_grad():
        for i in range(0, val.size(0), 1024):
            x = val[i:i + 1024]
            nll, _, _ = nll_of(model, prior, x, length(x.size(0)))
            total += float(nll.double().sum())
        val_nll = total / val.size(0)
        # first 256 sets with a recorded noise draw
        x256 = val[:256]
        torch.manual_seed(31337)
        u = torch.rand(256 * SET_SIZE, 1, D)
        torch.manual_seed(31337)
        nll256, z256, ldj256 = nll_of(model, prior, x256, length(256))
        dec256, _ = model(z256, reverse=True, length=length(256))
    val_bpd = float(np.log2(np.e) * val_nll)
    print("reference validation bpd %.4f (optimum %.4f), decode accuracy on 256 sets %.4f"
          % (val_bpd, calc_optimum(SET_SIZE), float((dec256 == x256).float().mean())))

    flat = {"sd_" + k: v.detach().numpy() for k, v in model.state_dict().items()}
    meta = dict(set_size=SET_SIZE, D=D, hidden=HIDDEN, transformer_layers=TLAYERS, flows=FLOWS, K=K, iters=ITERS,
                val_nll=val_nll, val_bpd=val_bpd, optimum_bpd=float(calc_optimum(SET_SIZE)), val_seed=123,
                infos=[l.info() for l in model.flow_layers])
    flat.update(meta=np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8), x256=x256.numpy(), u256=u.numpy(),
                z256=z256.numpy(), ldj256=ldj256.numpy(), nll256=nll256.numpy(), dec256=dec256.numpy())
    np.savez_compressed(OUT, **flat)
    print("wrote %s (%.0f kB)" % (OUT, os.path.getsize(OUT) / 1e3))


if __name__ == "__main__":
    main()
