"""Train / evaluate the set-modelling flow on MI355X: the host side of the reference's
`experiments/set_modeling/train.py` + `general/train.py` + `general/task.py` reduced to what this experiment needs.

    python -m categoricalnf_amd.experiments.run_set_modeling --dataset shuffling --max_iterations 20000 \
        --checkpoint_path checkpoints/shuffling
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 -m \
        categoricalnf_amd.experiments.run_set_modeling --dataset summation ...           # one process per GPU

Same hyper-parameter names and defaults as the reference's CLI (experiments/set_modeling/train.py:67-76,
general/train.py:331-358: RAdam, lr 7.5e-4 decayed by 0.999975 per step, gradient norm clipped at 0.25, batch 64,
evaluation every 2000 iterations on the 32768 fixed validation sets) and the same checkpoint files
(`checkpoint_%07d.tar` holding `model_state_dict` [+ optimizer / scheduler state], `iteration`, `best_save_dict`,
`evaluation_dict`; general/train.py:257-276), so checkpoints move between the two code bases.
`param_config.pik` (the pickled argument namespace, general/train.py:428-432) is written beside them and read back by
`--only_eval`, which then rebuilds the model from the run's own hyper-parameters.
Not reproduced: tensorboard summaries, the discrete-flow baseline.

Multi-GPU replaces `nn.DataParallel` (general/train.py:36-44) by one process per GPU: every rank draws its own
training batches of `batch_size / world` sets, DistributedDataParallel all-reduces the gradients over RCCL, and
the validation sets are sharded by rank with ONE all-reduce of (sum nll, count) per evaluation."""
import argparse
import contextlib
import glob
import io
import json
import os
import pickle
import time

import numpy as np
import torch

from .. import functional as Fn
from ..distributed import allreduce_nll, init_process_group, shard_bounds, wrap_ddp
from ..host_utils import FlatParameters
from .set_modeling import FlowSetModeling, SetShufflingDataset, SetSummationDataset

LOG2E = float(np.log2(np.e))


def parse(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--dataset", default="shuffling", choices=["shuffling", "summation"])
    p.add_argument("--set_size", type=int, default=16)
    p.add_argument("--max_iterations", type=int, default=100000)
    p.add_argument("--batch_size", type=int, default=64)
    p.add_argument("--eval_freq", type=int, default=2000)
    p.add_argument("--save_freq", type=int, default=10000)
    p.add_argument("--print_freq", type=int, default=250)
    p.add_argument("--eval_batch_size", type=int, default=4096)
    p.add_argument("--seed", type=int, default=42)
    p.add_argument("--checkpoint_path", default=None)
    p.add_argument("--only_eval", action="store_true")
    p.add_argument("--load_best_model", action="store_true",
                   help="with --only_eval: evaluate the best-validation checkpoint instead of the newest")
    p.add_argument("--learning_rate", type=float, default=7.5e-4)
    p.add_argument("--lr_decay_factor", type=float, default=0.999975)
    p.add_argument("--lr_decay_step", type=int, default=1)
    p.add_argument("--lr_minimum", type=float, default=0.0)
    p.add_argument("--max_gradient_norm", type=float, default=0.25)
    p.add_argument("--encoding_dim", type=int, default=4)
    p.add_argument("--coupling_hidden_size", type=int, default=256)
    p.add_argument("--coupling_hidden_layers", type=int, default=2)
    p.add_argument("--coupling_num_flows", type=int, default=8)
    p.add_argument("--coupling_mask_ratio", type=float, default=0.5)
    p.add_argument("--coupling_num_mixtures", type=int, default=8)
    p.add_argument("--compact_params", action="store_true",
                   help="the mixture couplings' sub-networks emit the transformed channels' parameter blocks only (the last Linear applies "
                        "just those rows; same parameters and checkpoints): half the parameter traffic, no zero blocks in the backward")
    p.add_argument("--backend", default=None, help="torch.distributed backend (default: nccl = RCCL)")
    p.add_argument("--graph_step", action="store_true",
                   help="single process only: the whole training step (forward, HIP backward kernels, clipping, RAdam) is captured "
                        "once in a HIP graph and replayed (categoricalnf_amd.graphs.GraphedTrainStep); between replays only the "
                        "batch, the encoder noise and the learning rate are written into static device buffers")
    p.add_argument("--graph_step_unverified", action="store_true",
                   help="with --graph_step: accept a captured step whose hipGraph could not be inspected for memset nodes "
                        "(a torch without CUDAGraph(keep_graph=True)); without it such a step is refused")
    p.add_argument("--flat_optimizer", action="store_true",
                   help="single process only: optimiser, clipping and zero_grad on ONE flat parameter buffer "
                        "(host_utils.FlatParameters: 23.0 -> 21.8 ms per step at batch 64); checkpoints "
                        "then carry no optimiser moments (the model and the schedule position resume)")
    return p.parse_args(argv)


def model_params(args):
    return {"set_size": args.set_size, "coupling_hidden_layers": args.coupling_hidden_layers,
            "coupling_hidden_size": args.coupling_hidden_size, "coupling_num_flows": args.coupling_num_flows,
            "coupling_mask_ratio": args.coupling_mask_ratio, "coupling_num_mixtures": args.coupling_num_mixtures,
            "categ_encoding": {"use_dequantization": False, "use_variational": False, "use_decoder": False,
                               "num_dimensions": args.encoding_dim,
                               "flow_config": {"num_flows": 0, "hidden_layers": 2, "hidden_size": 128},
                               "decoder_config": {"num_layers": 1, "hidden_size": 64}}}


def checkpoint_file(path, iteration):
    return os.path.join(path, "checkpoint_" + str(iteration).zfill(7) + ".tar")


def save_checkpoint(path, iteration, model, optimizer=None, scheduler=None, **extra):
    os.makedirs(path, exist_ok=True)
    inner = model.module if hasattr(model, "module") else model
    blob = {"model_state_dict": inner.state_dict(), "iteration": iteration}
    if optimizer is not None:
        blob["optimizer_state_dict"] = optimizer.state_dict()
    if scheduler is not None:
        blob["scheduler_state_dict"] = scheduler.state_dict()
    blob.update(extra)
    torch.save(blob, checkpoint_file(path, iteration))
    return checkpoint_file(path, iteration)


PARAM_CONFIG_FILE = "param_config.pik"          # general/mutils.py:15


def save_args(path, args):
    """The run's argument namespace beside its checkpoints (general/train.py:428-432), so that evaluation can rebuild
    the model without repeating the command line."""
    os.makedirs(path, exist_ok=True)
    with open(os.path.join(path, PARAM_CONFIG_FILE), "wb") as f:
        pickle.dump(args, f)


def load_args(path):
    """general/mutils.py:123-133: `path` = the checkpoint directory or a file in it."""
    if os.path.isfile(path):
        path = os.path.dirname(path)
    with open(os.path.join(path, PARAM_CONFIG_FILE), "rb") as f:
        return pickle.load(f)


def load_checkpoint(path, model=None, optimizer=None, scheduler=None, device="cpu", load_best_model=False):
    """`path` = a checkpoint file or a directory (its newest `*.tar`, or with `load_best_model` the file its
    `best_save_dict` names); returns the non-state entries (general/mutils.py:66-112)."""
    was_dir = os.path.isdir(path)
    if was_dir:
        files = sorted(glob.glob(os.path.join(path, "*.tar")))
        if not files:
            print("No checkpoint files found at", path)
            return {}
        path = files[-1]
    blob = torch.load(path, map_location=device, weights_only=False)
    if was_dir and load_best_model:
        best_file = (blob.get("best_save_dict") or {}).get("file")
        if best_file and not os.path.isfile(best_file):         # the directory may have moved since the run
            best_file = os.path.join(os.path.dirname(path), os.path.basename(best_file))
        if best_file and os.path.isfile(best_file):
            return load_checkpoint(best_file, model, optimizer, scheduler, device=device)
        print("[!] WARNING: Best save dict file is listed as \"%s\", but file could not been found. Using default one..."
              % str((blob.get("best_save_dict") or {}).get("file")))
    if model is not None:
        inner = model.module if hasattr(model, "module") else model
        state = inner.state_dict()
        state.update(blob["model_state_dict"])
        inner.load_state_dict(state)
    if optimizer is not None and "optimizer_state_dict" in blob:
        optimizer.load_state_dict(blob["optimizer_state_dict"])
    if scheduler is not None and "scheduler_state_dict" in blob:
        scheduler.load_state_dict(blob["scheduler_state_dict"])
    elif scheduler is not None and blob.get("iteration", 0) > 0:
        # a best-validation checkpoint holds the model only: continue the learning-rate schedule where the run stopped
        # (the optimiser's moment estimates start afresh) instead of silently jumping back to the initial rate
        print("[#] WARNING: %s has no optimizer / scheduler state; the schedule is advanced to iteration %d and the "
              "optimizer moments restart" % (path, blob["iteration"]))
        advance_schedule(scheduler, blob["iteration"])
    return {k: v for k, v in blob.items() if "state_dict" not in k}


def advance_schedule(scheduler, iteration):
    """Put a LambdaLR where it stands after `iteration` steps (its optimiser's learning rates included)."""
    scheduler.last_epoch = int(iteration)
    for group, base, fn in zip(scheduler.optimizer.param_groups, scheduler.base_lrs, scheduler.lr_lambdas):
        group["lr"] = base * fn(scheduler.last_epoch)
    scheduler._last_lr = [g["lr"] for g in scheduler.optimizer.param_groups]


@torch.no_grad()
def evaluate(model, sets, device, rank=0, world=1, batch_size=4096):
    """Mean NLL per element and bits/dim over `sets` (int64 numpy [M, set_size]); rows sharded over the ranks,
    (sum nll, count) accumulated on the device and all-reduced once."""
    inner = model.module if hasattr(model, "module") else model
    was_training = inner.training
    inner.eval()
    lo, hi = shard_bounds(len(sets), rank, world)
    total = torch.zeros(2, dtype=torch.float64, device=device)
    part = torch.zeros(2, dtype=torch.float64, device=device)
    for i in range(lo, hi, batch_size):
        x = torch.from_numpy(sets[i:min(i + batch_size, hi)]).long().to(device)
        ln = torch.full((x.size(0),), x.size(1), dtype=torch.long, device=device)
        inner.nll(x, length=ln, beta=1, sums=part)            # (sum nll, count) of this batch, on the device
        total += part
    mean_nll, bpd = allreduce_nll(total)
    inner.train(was_training)
    return mean_nll, bpd


MODEL_ARGS = ("dataset", "set_size", "encoding_dim", "coupling_hidden_size", "coupling_hidden_layers",
              "coupling_num_flows", "coupling_mask_ratio", "coupling_num_mixtures")


def main(argv=None):
    """Runs the driver; the process-wide switches it sets (autograd threading, cudnn / MIOpen) are restored when it returns."""
    threading, cudnn = torch.autograd.is_multithreading_enabled(), torch.backends.cudnn.enabled
    try:
        return _main(argv)
    finally:
        torch.autograd.set_multithreading_enabled(threading)
        torch.backends.cudnn.enabled = cudnn


def _main(argv=None):
    args = parse(argv)
    # one process per GPU: backward() runs on the calling thread instead of being handed to the autograd engine's device
    # thread and waited for (two thread wake-ups per call; tools/autograd_overhead.py --single_thread)
    torch.autograd.set_multithreading_enabled(False)
    if args.only_eval and args.checkpoint_path and os.path.isfile(
            os.path.join(args.checkpoint_path if os.path.isdir(args.checkpoint_path) else os.path.dirname(args.checkpoint_path),
                         PARAM_CONFIG_FILE)):
        saved = load_args(args.checkpoint_path)          # the model is the one the run trained, whatever the flags say
        for name in MODEL_ARGS:
            if hasattr(saved, name):
                setattr(args, name, getattr(saved, name))
    rank, local_rank, world = init_process_group(args.backend)
    device = torch.device("cuda", local_rank if world > 1 else 0)
    torch.cuda.set_device(device)
    torch.manual_seed(args.seed)
    np.random.seed(args.seed)
    say = (lambda *a, **k: print(*a, flush=True, **k)) if rank == 0 else (lambda *a, **k: None)     # flushed: logs survive a kill

    data_cls = SetShufflingDataset if args.dataset == "shuffling" else SetSummationDataset
    train_set = data_cls(args.set_size, train=True)
    val_sets = data_cls(args.set_size, train=False, val=True).eval_sets()
    test_sets = data_cls(args.set_size, train=False, test=True).eval_sets()
    optimum = (SetShufflingDataset.optimum_bpd(args.set_size) if args.dataset == "shuffling" else train_set.optimum_bpd())
    with (contextlib.nullcontext() if rank == 0 else contextlib.redirect_stdout(io.StringIO())):
        model = FlowSetModeling(model_params(args), data_cls).to(device)
        if getattr(args, "compact_params", False):
            n = sum(bool(layer.enable_compact_params()) for layer in model.flow_layers if hasattr(layer, "enable_compact_params"))
            say("[#] --compact_params: %d mixture couplings on the compact parameter layout" % n)
    rng = np.random.RandomState(args.seed + 1000 * rank)           # every rank draws different training sets
    per_rank = max(1, args.batch_size // world)

    def batch():
        x = torch.from_numpy(train_set.sample(per_rank, rng)).long().to(device)
        return x, torch.full((x.size(0),), x.size(1), dtype=torch.long, device=device)

    optimizer = torch.optim.RAdam(model.parameters(), lr=args.learning_rate)
    floor = args.lr_minimum / args.learning_rate
    scheduler = torch.optim.lr_scheduler.LambdaLR(
        optimizer, lambda step: max(floor, args.lr_decay_factor ** (step // max(1, args.lr_decay_step))))
    state = {"iteration": 0, "best_save_dict": {"file": None, "metric": 1e6, "detailed_metrics": None, "test": None},
             "evaluation_dict": {}}
    if args.checkpoint_path and os.path.exists(args.checkpoint_path):
        state.update(load_checkpoint(args.checkpoint_path, model, optimizer, scheduler, device=device,
                                     load_best_model=args.only_eval and args.load_best_model))
    if state["iteration"] == 0 and not args.only_eval:
        # data-dependent ActNorm initialisation on 16 batches (general/task.py: initialize); every rank uses the
        # same sets so the replicas start identical
        init_rng = np.random.RandomState(args.seed)
        with contextlib.redirect_stdout(io.StringIO()):
            model.initialize_data_dependent([
                (torch.from_numpy(train_set.sample(args.batch_size, init_rng)).long().to(device),
                 {"length": torch.full((args.batch_size,), args.set_size, dtype=torch.long, device=device)})
                for _ in range(16)])
    ddp = wrap_ddp(model, device)

    if not args.only_eval and args.checkpoint_path and rank == 0:
        save_args(args.checkpoint_path, args)
    if args.only_eval:
        _, val_bpd = evaluate(ddp, val_sets, device, rank, world, args.eval_batch_size)
        _, test_bpd = evaluate(ddp, test_sets, device, rank, world, args.eval_batch_size)
        say("validation %.4f bpd, test %.4f bpd (optimum %.4f)" % (val_bpd, test_bpd, optimum))
        return {"val_bpd": val_bpd, "test_bpd": test_bpd, "optimum_bpd": optimum}

    flat = None
    if args.flat_optimizer and world > 1:
        say("[#] --flat_optimizer ignored: DistributedDataParallel's gradient buckets own the .grad views")
    elif args.flat_optimizer:
        flat = FlatParameters(model)                      # after the data-dependent init, which re-binds ActNorm's .data
        optimizer = torch.optim.RAdam(flat.parameters(), lr=args.learning_rate)
        scheduler = torch.optim.lr_scheduler.LambdaLR(
            optimizer, lambda step: max(floor, args.lr_decay_factor ** (step // max(1, args.lr_decay_step))))
        advance_schedule(scheduler, state["iteration"])
    graphed = None
    if args.graph_step and (world > 1 or flat is not None):
        say("[#] --graph_step ignored: it is a single-process mode without --flat_optimizer")
    elif args.graph_step:
        from ..graphs import GraphedTraining
        lr_of = lambda step: args.learning_rate * max(floor, args.lr_decay_factor ** (step // max(1, args.lr_decay_step)))
        model.train()
        static_x, static_ln = batch()
        static_noise = torch.rand(static_x.numel(), 1, args.encoding_dim, device=device)
        # the NLL assembly rides in the last coupling layer's kernel
        graphed = GraphedTraining(model, lambda: model(static_x, reverse=False, length=static_ln, beta=1, noise=static_noise,
                                                       _nll=model.nll_request(length=static_ln))[2].mean(),
                                  device, args.max_gradient_norm, lr=lr_of(state["iteration"]), eager_optimizer=optimizer, allow_unverified=args.graph_step_unverified)
        optimizer = graphed.optimizer
        say("[#] --graph_step: captured training step, hipGraph nodes %s" % (graphed.nodes,))
    ddp.train()
    best = state["best_save_dict"]
    periodic = set()          # full-state checkpoints written at save_freq steps (kept when a better validation file appears)
    periodic_list = os.path.join(args.checkpoint_path, "periodic_checkpoints.json") if args.checkpoint_path else None
    if args.checkpoint_path and rank == 0 and os.path.isdir(args.checkpoint_path):
        # resumed: the full-state files already in the directory stay protected (a pre-resume periodic file that is also the best
        # one must not be removed when a better validation file appears).  Their names are kept in a sidecar file; only a
        # directory written before the sidecar existed is scanned by unpickling every checkpoint once
        import glob
        listed = None
        if os.path.isfile(periodic_list):
            try:
                listed = [os.path.join(args.checkpoint_path, n) for n in json.load(open(periodic_list))]
            except (ValueError, OSError):
                listed = None          # a truncated sidecar (a crash during its write): fall back to the scan
        if listed is not None:
            periodic.update(f for f in listed if os.path.isfile(f))
        else:
            for f in glob.glob(os.path.join(args.checkpoint_path, "checkpoint_*.tar")):
                try:
                    if "optimizer_state_dict" in torch.load(f, map_location="cpu", weights_only=False):
                        periodic.add(f)
                except Exception:
                    pass
    t0, run_loss, seen = time.time(), torch.zeros((), device=device), 0       # the loss stays on the device between prints
    for it in range(state["iteration"], args.max_iterations):
        x, ln = batch()
        if graphed is not None:
            static_x.copy_(x, non_blocking=True)
            static_noise.uniform_()
            loss = graphed(lr_of(it))
            scheduler.last_epoch, scheduler._last_lr = it + 1, [lr_of(it + 1)]      # what the checkpoint stores of the schedule
        else:
            # the NLL assembly rides in the last coupling layer's kernel
            loss = ddp(x, reverse=False, length=ln, beta=1, _nll=model.nll_request(length=ln))[2].mean()
            if flat is not None:
                flat.zero_grad()
            else:
                optimizer.zero_grad(set_to_none=True)
            loss.backward()
            torch.nn.utils.clip_grad_norm_(flat.parameters() if flat is not None else ddp.parameters(), args.max_gradient_norm)
            optimizer.step()
            scheduler.step()
        run_loss += loss.detach()
        seen += 1
        step = it + 1
        if step % args.print_freq == 0:
            say("iteration %7d | train %.4f bpd | %.1f it/s" % (step, float(run_loss) / seen * LOG2E, seen / (time.time() - t0)))
            t0, seen = time.time(), 0
            run_loss.zero_()
        if step % args.eval_freq == 0 or step == args.max_iterations:
            if graphed is not None:
                graphed.drop_weight_caches()
            val_nll, val_bpd = evaluate(ddp, val_sets, device, rank, world, args.eval_batch_size)
            state["evaluation_dict"][step] = val_nll
            say("iteration %7d | validation %.4f bpd (optimum %.4f)" % (step, val_bpd, optimum))
            if val_nll < best["metric"] and args.checkpoint_path and rank == 0:
                # the previous best file goes, unless it is also a periodic full-state checkpoint (optimizer and scheduler
                # included; the isfile guard of general/train.py:215 kept those too): a resume needs the newest of them
                if best["file"] and os.path.isfile(best["file"]) and best["file"] not in periodic:
                    os.remove(best["file"])
                best.update(file=checkpoint_file(args.checkpoint_path, step), metric=val_nll,
                            detailed_metrics={"val_bpd": val_bpd})
                save_checkpoint(args.checkpoint_path, step, ddp, best_save_dict=best, evaluation_dict=state["evaluation_dict"])
        if step % args.save_freq == 0 and args.checkpoint_path and rank == 0:
            # always the full state (a best-validation file of the same step is a subset of it and is replaced)
            # (a graphed run's file stores what an eager run stores after scheduler.step(): the NEXT step's learning rate as a
            # number, and no capturable flag)
            with (graphed.checkpoint_groups(lr_of(step)) if graphed is not None else contextlib.nullcontext()):
                save_checkpoint(args.checkpoint_path, step, ddp, optimizer if flat is None else None, scheduler,
                                best_save_dict=best, evaluation_dict=state["evaluation_dict"])
            periodic.add(checkpoint_file(args.checkpoint_path, step))
            # written to a temporary file and renamed into place: a crash leaves the old list or the new one, never half of one
            with open(periodic_list + ".tmp", "w") as fh:
                json.dump(sorted(os.path.basename(f) for f in periodic), fh)
            os.replace(periodic_list + ".tmp", periodic_list)
    if graphed is not None:
        graphed.drop_weight_caches()
    _, val_bpd = evaluate(ddp, val_sets, device, rank, world, args.eval_batch_size)
    _, test_bpd = evaluate(ddp, test_sets, device, rank, world, args.eval_batch_size)
    say("final: validation %.4f bpd, test %.4f bpd (optimum %.4f)" % (val_bpd, test_bpd, optimum))
    if args.checkpoint_path and rank == 0:
        with open(os.path.join(args.checkpoint_path, "results.txt"), "w") as f:
            f.write("Best validation performance: %s\nFinal validation bpd: %.6f\nTest bpd: %.6f\nOptimum bpd: %.6f\n"
                    % (str(best["metric"]), val_bpd, test_bpd, optimum))
    if world > 1:
        torch.distributed.barrier()
    return {"val_bpd": val_bpd, "test_bpd": test_bpd, "optimum_bpd": optimum, "best_val_nll": best["metric"],
            "best_file": best["file"]}


if __name__ == "__main__":
    main()
