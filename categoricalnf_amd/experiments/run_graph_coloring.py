"""Train / evaluate the graph-colouring flow (node-based GraphCNF) on MI355X: the host side of the reference's
`experiments/graph_coloring/train.py` + `task.py` reduced to what this experiment needs.

    python -m categoricalnf_amd.experiments.run_graph_coloring --dataset tiny_3 --generate_data \\
        --data_root data/ --max_iterations 20000 --checkpoint_path checkpoints/tiny_3_CNF

Hyper-parameter names and defaults are the reference's (train.py:82-93 and its README: `tiny_3`: batch 384, encoding_dim
2, 8 mixtures; `large_3`: batch 128, encoding_dim 6, 16 mixtures; RAdam, lr 7.5e-4 decayed by 0.999975 per step, gradient
norm 0.25, beta from 1 to 2 on the exponential schedule of parameter_scheduler.py:109-121 with step size 5000), batches
come from the length-bucketed sampler, the loss is the per-node negative log-likelihood (task.py:84-101), evaluation
reports bits per node on the validation graphs and the share of VALID colourings among samples drawn for them
(task.py:170-215), checkpoints use the reference's file format (`run_set_modeling.save_checkpoint`).  The reference's
data files are not reachable from here: `--generate_data` writes a synthetic set in the same format (planted colourings,
`graph_coloring_data.generate_planted_dataset`) when the files are missing.

Multi-GPU: one process per GPU under `python -m torch.distributed.run --nproc-per-node N ... -m
categoricalnf_amd.experiments.run_graph_coloring ...`: every rank walks its own bucketed order of the training graphs with
`batch_size / N` graphs per step, gradients are all-reduced by DistributedDataParallel over RCCL, evaluation batches are dealt
out to the ranks round-robin and (sum of NLL, graphs, valid colourings) is all-reduced once."""
import argparse
import contextlib
import io
import os
import random
import time

import numpy as np
import torch

from ..distributed import init_process_group, wrap_ddp
from ..layers.flows.distributions import LogisticDistribution
from .graph_coloring import GraphNodeFlow, flow_nll, generation_validity
from .graph_coloring_data import GraphColoringDataset, generate_planted_dataset
from .run_set_modeling import checkpoint_file, load_checkpoint, save_args, save_checkpoint

LOG2E = float(np.log2(np.e))
SIZES = {"tiny": (10, 20), "large": (25, 50)}


def parse(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--dataset", default="tiny_3", help="size_numcolors: tiny_3 or large_3")
    p.add_argument("--data_root", default="data/")
    p.add_argument("--generate_data", action="store_true", help="write a synthetic data set if the files are missing")
    p.add_argument("--num_graphs", type=int, default=60000, help="size of the generated data set")
    p.add_argument("--max_iterations", type=int, default=200000)
    p.add_argument("--batch_size", type=int, default=384)
    p.add_argument("--eval_freq", type=int, default=2000)
    p.add_argument("--print_freq", type=int, default=250)
    p.add_argument("--eval_batch_size", type=int, default=1024)
    p.add_argument("--seed", type=int, default=42)
    p.add_argument("--checkpoint_path", default=None)
    p.add_argument("--only_eval", action="store_true")
    p.add_argument("--learning_rate", type=float, default=7.5e-4)
    p.add_argument("--lr_decay_factor", type=float, default=0.999975)
    p.add_argument("--max_gradient_norm", type=float, default=0.25)
    p.add_argument("--encoding_dim", type=int, default=2)
    p.add_argument("--coupling_hidden_size", type=int, default=384)
    p.add_argument("--coupling_hidden_layers", type=int, default=4)
    p.add_argument("--coupling_num_flows", type=int, default=8)
    p.add_argument("--coupling_mask_ratio", type=float, default=0.5)
    p.add_argument("--coupling_num_mixtures", type=int, default=8)
    p.add_argument("--coupling_dropout", type=float, default=0.0)
    p.add_argument("--beta_scheduler_start_val", type=float, default=1.0)
    p.add_argument("--beta_scheduler_end_val", type=float, default=2.0)
    p.add_argument("--beta_scheduler_step_size", type=int, default=5000)
    p.add_argument("--beta_scheduler_logit", type=float, default=2.0)
    p.add_argument("--graph_step", action="store_true",
                   help="capture the training step (forward, HIP backward kernels, clipping, RAdam) in a HIP graph and replay it; "
                        "single process; batches keep the data set's full width, beta and the learning rate live in device scalars")
    p.add_argument("--graph_step_unverified", action="store_true",
                   help="with --graph_step: accept a captured step whose hipGraph could not be inspected for memset nodes "
                        "(a torch without CUDAGraph(keep_graph=True)); without it such a step is refused")
    p.add_argument("--backend", default=None, help="torch.distributed backend when started with WORLD_SIZE > 1 (nccl = RCCL)")
    p.add_argument("--share_device", action="store_true", help="TEST ONLY: every rank on cuda:0 (1-GPU box, --backend gloo)")
    return p.parse_args(argv)


def model_params(args):
    return {"coupling_num_flows": args.coupling_num_flows, "coupling_hidden_size": args.coupling_hidden_size,
            "coupling_hidden_layers": args.coupling_hidden_layers, "coupling_num_mixtures": args.coupling_num_mixtures,
            "coupling_mask_ratio": args.coupling_mask_ratio, "coupling_dropout": args.coupling_dropout,
            "categ_encoding": {"use_dequantization": False, "use_variational": False, "use_decoder": False,
                               "num_dimensions": args.encoding_dim, "flow_config": {"num_flows": 0}, "decoder_config": {}}}


def beta_at(args, iteration):
    """ExponentialScheduler.get_val (general/parameter_scheduler.py:120-121)."""
    a, b = args.beta_scheduler_start_val, args.beta_scheduler_end_val
    return a + (b - a) * (1.0 - args.beta_scheduler_logit ** (-iteration * 1.0 / args.beta_scheduler_step_size))


def collate(dataset, indices, device, clip=True):
    """(nodes, adjacency, length) of a batch, clipped to its longest graph (task.py:133-141); `clip=False` keeps the data set's
    full width (one shape for every batch: a captured training step)."""
    items = [dataset[i] for i in indices]
    length = torch.from_numpy(np.array([it[2] for it in items], dtype=np.int64))
    n = int(length.max()) if clip else int(items[0][0].shape[0])
    nodes = torch.from_numpy(np.stack([it[0][:n] for it in items]))
    adjacency = torch.from_numpy(np.stack([it[1][:n, :n] for it in items]))
    return nodes.to(device), adjacency.to(device), length.to(device)


def batches(dataset, batch_size, device, drop_last, clip=True):
    for idx in dataset.get_sampler(batch_size, drop_last=drop_last):
        yield collate(dataset, idx, device, clip=clip)


def evaluation_share(dataset, batch_size, rank=0, world=1, max_graphs=None):
    """The evaluation batches (lists of dataset indices) `rank` of `world` is responsible for: the bucketed batch order is
    drawn under a fixed seed — every rank must see the SAME order to take every `world`-th batch of it — the first batches
    covering `max_graphs` graphs are kept, and the caller's numpy random state is left as it was."""
    state = np.random.get_state()
    np.random.seed(1234)
    order = list(dataset.get_sampler(batch_size, drop_last=False))
    np.random.set_state(state)
    mine, seen = [], 0
    for b, idx in enumerate(order):
        if max_graphs is not None and seen >= max_graphs:
            break
        seen += len(idx)
        if b % world == rank:
            mine.append(list(idx))
    return mine


@torch.no_grad()
def evaluate(model, prior, dataset, device, batch_size, max_graphs=None, rank=0, world=1):
    """(bits per node, validity of sampled colourings) on `dataset`; with several ranks the batches are dealt out
    round-robin and the three sums are all-reduced once."""
    inner = model.module if hasattr(model, "module") else model
    inner.eval()
    sums = torch.zeros(3, dtype=torch.float64, device=device)          # sum of nll, graphs, valid colourings
    for idx in evaluation_share(dataset, batch_size, rank, world, max_graphs):
        nodes, adjacency, length = collate(dataset, idx, device)
        nll, _ = flow_nll(inner, prior, nodes, adjacency, length)
        validity = generation_validity(inner, prior, [(nodes, adjacency, length)], type(dataset))
        sums[0] += nll.double().sum()
        sums[1] += nodes.shape[0]
        sums[2] += validity["valid_ratio"] * nodes.shape[0]
    if world > 1:
        torch.distributed.all_reduce(sums)
    inner.train()
    total, count, valid = (float(v) for v in sums)
    return total / max(count, 1.0) * LOG2E, valid / max(count, 1.0)


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
    rank, local_rank, world = init_process_group("gloo" if args.share_device else args.backend)
    device = torch.device("cuda", local_rank if (world > 1 and not args.share_device) else 0)
    torch.cuda.set_device(device)
    say = (lambda *a: print(*a, flush=True)) if rank == 0 else (lambda *a: None)
    torch.manual_seed(args.seed)
    np.random.seed(args.seed)
    size, colours = args.dataset.split("_")[0], int(args.dataset.split("_")[1])
    GraphColoringDataset.set_dataset(prefix="_" + size, num_colors=colours)
    GraphColoringDataset.DATASET_NODES = GraphColoringDataset.DATASET_VAL_IDX = None          # a new data set selection
    data_file = os.path.join(args.data_root, GraphColoringDataset.DATA_FILENAME)
    if args.generate_data and not os.path.isfile(data_file) and rank == 0:
        lo, hi = SIZES.get(size, (10, 20))
        say("generating %d synthetic graphs with %d..%d nodes (planted %d-colourings) under %s"
            % (args.num_graphs, lo, hi, colours, args.data_root))
        generate_planted_dataset(args.data_root, prefix="_" + size, num_colors=colours, num_graphs=args.num_graphs,
                                 n_min=lo, n_max=hi, seed=args.seed)
    if world > 1:
        torch.distributed.barrier()                  # the files are there before anyone reads them
    train_set = GraphColoringDataset(num_colors=colours, train=True, data_root=args.data_root)
    val_set = GraphColoringDataset(num_colors=colours, val=True, data_root=args.data_root)
    test_set = GraphColoringDataset(num_colors=colours, test=True, data_root=args.data_root)
    with contextlib.redirect_stdout(io.StringIO()):
        model = GraphNodeFlow(model_params(args), GraphColoringDataset).to(device)
    prior = LogisticDistribution(mu=0.0, sigma=1.0)
    optimizer = torch.optim.RAdam(model.parameters(), lr=args.learning_rate)
    scheduler = torch.optim.lr_scheduler.LambdaLR(optimizer, lambda step: args.lr_decay_factor ** step)
    state = {"iteration": 0, "best_save_dict": {"file": None, "metric": 1e6, "detailed_metrics": None, "test": None},
             "evaluation_dict": {}}
    if args.checkpoint_path and os.path.exists(args.checkpoint_path):
        state.update(load_checkpoint(args.checkpoint_path, model, optimizer, scheduler, device=device))

    per_rank = max(1, args.batch_size // world)

    graph_mode = args.graph_step and world == 1
    if args.graph_step and not graph_mode:
        say("[#] --graph_step ignored: it is a single-process mode")

    def stream():
        while True:
            yield from batches(train_set, per_rank, device, drop_last=True, clip=not graph_mode)
    feed = stream()
    if state["iteration"] == 0 and not args.only_eval:
        # data-dependent ActNorm initialisation on 16 batches (task.py:144-157), full-width graphs
        init = []
        for _ in range(16):
            idx = np.random.randint(0, len(train_set), size=args.batch_size)
            items = [train_set[i] for i in idx]
            x = torch.from_numpy(np.stack([it[0] for it in items])).to(device)
            adj = torch.from_numpy(np.stack([it[1] for it in items])).to(device)
            ln = torch.from_numpy(np.array([it[2] for it in items], dtype=np.int64)).to(device)
            init.append((x, {"length": ln, "adjacency": adj}))
        with contextlib.redirect_stdout(io.StringIO()):
            model.initialize_data_dependent(init)
        nodes, adjacency, length = next(feed)
        assert model.test_permutation(nodes, adjacency, length), "[!] ERROR: Permutation test failed."
        assert model.test_reversibility(nodes, adjacency, length), "[!] ERROR: Reversibility test failed."
    np.random.seed(args.seed + 1000 * rank + 1)          # from here on every rank draws its own order of training graphs
    random.seed(args.seed + 1000 * rank + 1)             # the dataset's colour permutations / random node orders (general/mutils.py:29-36 seeds it too)
    torch.manual_seed(args.seed + 1000 * rank + 1)       # dropout masks differ between the ranks; the replicas are initialised
    ddp = wrap_ddp(model, device)
    if not args.only_eval and args.checkpoint_path and rank == 0:
        save_args(args.checkpoint_path, args)
    if args.only_eval:
        val_bpd, val_valid = evaluate(ddp, prior, val_set, device, args.eval_batch_size, rank=rank, world=world)
        say("validation %.4f bits per node, %.2f %% valid colourings" % (val_bpd, 100 * val_valid))
        return {"val_bpd": val_bpd, "val_valid_ratio": val_valid}

    graphed = None
    if graph_mode:
        from ..graphs import GraphedTraining
        lr_of = lambda step: args.learning_rate * args.lr_decay_factor ** step
        model.train()
        s_nodes, s_adj, s_len = (t.clone() for t in next(feed))
        s_noise = torch.rand(s_nodes.numel(), 1, model.embed_dim, device=device)
        beta_t = torch.tensor(beta_at(args, state["iteration"]), dtype=torch.float32, device=device)      # the beta schedule lives in a device scalar
        graphed = GraphedTraining(model, lambda: model(s_nodes, s_adj, reverse=False, beta=beta_t, length=s_len, noise=s_noise,
                                                       _nll=model.nll_request(length=s_len, prior=prior))[2].mean(),
                                  device, args.max_gradient_norm, lr=lr_of(state["iteration"]), eager_optimizer=optimizer, allow_unverified=args.graph_step_unverified)
        optimizer = graphed.optimizer
        say("[#] --graph_step: captured training step, hipGraph nodes %s" % (graphed.nodes,))
    ddp.train()
    best = state["best_save_dict"]
    t0, run_loss, seen = time.time(), torch.zeros((), device=device), 0
    for it in range(state["iteration"], args.max_iterations):
        nodes, adjacency, length = next(feed)
        if graphed is not None:
            s_nodes.copy_(nodes, non_blocking=True); s_adj.copy_(adjacency, non_blocking=True); s_len.copy_(length, non_blocking=True)
            s_noise.uniform_()
            beta_t.fill_(beta_at(args, it))
            loss = graphed(lr_of(it))
            scheduler.last_epoch, scheduler._last_lr = it + 1, [lr_of(it + 1)]      # what the checkpoint stores of the schedule
        else:
            nll, _ = flow_nll(ddp, prior, nodes, adjacency, length, beta=beta_at(args, it))
            loss = nll.mean()
            optimizer.zero_grad(set_to_none=True)
            loss.backward()
            torch.nn.utils.clip_grad_norm_(ddp.parameters(), args.max_gradient_norm)
            optimizer.step()
            scheduler.step()
        run_loss += loss.detach()
        seen += 1
        step = it + 1
        if step % args.print_freq == 0:
            say("iteration %7d | train %.4f bits per node (beta %.2f) | %.1f it/s"
                % (step, float(run_loss) / seen * LOG2E, beta_at(args, it), seen / (time.time() - t0)))
            t0, seen = time.time(), 0
            run_loss.zero_()
        if step % args.eval_freq == 0 or step == args.max_iterations:
            if graphed is not None:
                graphed.drop_weight_caches()
            val_bpd, val_valid = evaluate(ddp, prior, val_set, device, args.eval_batch_size, max_graphs=8192, rank=rank,
                                          world=world)
            state["evaluation_dict"][step] = val_bpd
            say("iteration %7d | validation %.4f bits per node, %.2f %% valid colourings" % (step, val_bpd, 100 * val_valid))
            if val_bpd < best["metric"] and args.checkpoint_path and rank == 0:
                if best["file"] and os.path.isfile(best["file"]):
                    os.remove(best["file"])
                best.update(file=checkpoint_file(args.checkpoint_path, step), metric=val_bpd,
                            detailed_metrics={"val_bpd": val_bpd, "valid_ratio": val_valid})
                with (graphed.checkpoint_groups(lr_of(step)) if graphed is not None else contextlib.nullcontext()):
                    save_checkpoint(args.checkpoint_path, step, ddp, optimizer, scheduler, best_save_dict=best,
                                    evaluation_dict=state["evaluation_dict"])
    if graphed is not None:
        graphed.drop_weight_caches()
    val_bpd, val_valid = evaluate(ddp, prior, val_set, device, args.eval_batch_size, rank=rank, world=world)
    test_bpd, test_valid = evaluate(ddp, prior, test_set, device, args.eval_batch_size, rank=rank, world=world)
    say("final: validation %.4f bits per node / %.2f %% valid, test %.4f / %.2f %%"
        % (val_bpd, 100 * val_valid, test_bpd, 100 * test_valid))
    if world > 1:
        torch.distributed.barrier()
    return {"val_bpd": val_bpd, "val_valid_ratio": val_valid, "test_bpd": test_bpd, "test_valid_ratio": test_valid,
            "best_file": best["file"]}


if __name__ == "__main__":
    main()
