"""Train / evaluate the language-modelling flow (autoregressive mixture-CDF coupling with an LSTM sub-network) on MI355X:
the host side of the reference's `experiments/language_modeling/train.py` + `task.py` reduced to what the flow needs.

    python -m categoricalnf_amd.experiments.run_language_modeling --max_iterations 100000 --checkpoint_path checkpoints/lm

Hyper-parameter names and defaults are the reference's text8 recipe (its README: max_seq_len 256, batch 128, encoding_dim 3,
2 LSTM layers of 1024, 27 mixtures, input dropout 0.05, RAdam, lr 7.5e-4; train.py:71-85: one flow, beta from 1 to 2 on the
exponential schedule with step size 5000; general/train.py: lr decayed by 0.999975 per step, gradient norm 0.25); the loss
is the per-character negative log-likelihood `(-ldj - sum log p(z)) / length` (task.py:75-119), reported in bits per
character, checkpoints use the reference's file format (`run_set_modeling.save_checkpoint`).

The reference's corpora (Penn Treebank, text8, Wikitext) are downloaded by torchnlp / torchtext, neither of which — nor the
files — is reachable from here.  The data set is therefore a synthetic character source with the alphabet size of text8
(27 symbols) whose TRUE entropy rate is known: a second-order Markov chain with transition probabilities drawn once from a
Dirichlet distribution (`MarkovCorpus`).  That gives the run a yardstick no real corpus has: the bits per character of a
perfect model are `corpus.entropy_rate()` exactly, and a model that ignores context cannot do better than
`corpus.unigram_entropy()`.  `--variable_length` draws sentence lengths uniformly from [max_seq_len / 4, max_seq_len] and
pads, which exercises the padding masks the way Penn Treebank does.

Multi-GPU: one process per GPU under `python -m torch.distributed.run --nproc-per-node N ... -m
categoricalnf_amd.experiments.run_language_modeling ...` — every rank draws its own `batch_size / N` sentences, gradients
are all-reduced by DistributedDataParallel over RCCL, the held-out sentences are sharded over the ranks and their
(sum of NLL, count) pair is all-reduced once per evaluation (`categoricalnf_amd.distributed`)."""
import argparse
import contextlib
import io
import os
import time

import numpy as np
import torch

from ..distributed import allreduce_nll, init_process_group, shard_bounds, wrap_ddp
from ..host_utils import create_channel_mask
from ..layers.flows.distributions import LogisticDistribution
from .language_modeling import FlowLanguageModeling
from .run_set_modeling import checkpoint_file, load_checkpoint, save_args, save_checkpoint

LOG2E = float(np.log2(np.e))


class MarkovCorpus:
    """Second-order Markov source over `vocab_size` symbols: p(x_t | x_{t-2}, x_{t-1}) = T[x_{t-2}, x_{t-1}, :], rows drawn
    from Dirichlet(alpha) under `seed` (alpha < 1: peaked rows, a few likely continuations per context — text-like)."""

    def __init__(self, vocab_size=27, alpha=0.15, seed=0):
        self.vocab_size = vocab_size
        rng = np.random.RandomState(seed)
        self.T = rng.dirichlet(np.full(vocab_size, alpha), size=(vocab_size, vocab_size))     # [a, b, c]
        # stationary distribution of the pair chain (a, b) -> (b, c) by power iteration
        pi = np.full((vocab_size, vocab_size), 1.0 / vocab_size ** 2)
        for _ in range(2000):
            new = np.einsum("ab,abc->bc", pi, self.T)
            done = np.abs(new - pi).max() < 1e-15
            pi = new
            if done:
                break
        self.pair_stationary = pi / pi.sum()
        self.cum = np.cumsum(self.T, axis=-1)

    def entropy_rate(self):
        """Bits per symbol of the source: sum_ab pi(a, b) H(T[a, b, :])."""
        h = -(self.T * np.log2(np.clip(self.T, 1e-300, None))).sum(-1)
        return float((self.pair_stationary * h).sum())

    def unigram_entropy(self):
        """Bits per symbol of the best context-free model (entropy of the marginal symbol distribution)."""
        p = self.pair_stationary.sum(0)
        return float(-(p * np.log2(np.clip(p, 1e-300, None))).sum())

    def sample(self, num, length, rng):
        """int64 [num, length]: sequences started from the stationary pair distribution."""
        V = self.vocab_size
        start = rng.choice(V * V, size=num, p=self.pair_stationary.reshape(-1))
        out = np.empty((num, length), dtype=np.int64)
        out[:, 0], out[:, 1] = start // V, start % V
        u = rng.rand(num, length)
        for t in range(2, length):
            rows = self.cum[out[:, t - 2], out[:, t - 1]]                   # [num, V]
            out[:, t] = np.minimum((u[:, t, None] > rows).sum(-1), V - 1)
        return out


def parse(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--vocab_size", type=int, default=27)
    p.add_argument("--source_alpha", type=float, default=0.15, help="Dirichlet concentration of the synthetic source")
    p.add_argument("--source_seed", type=int, default=0)
    p.add_argument("--variable_length", action="store_true")
    p.add_argument("--num_val", type=int, default=2048)
    p.add_argument("--max_seq_len", type=int, default=256)
    p.add_argument("--max_iterations", type=int, default=100000)
    p.add_argument("--batch_size", type=int, default=128)
    p.add_argument("--eval_freq", type=int, default=2000)
    p.add_argument("--print_freq", type=int, default=250)
    p.add_argument("--seed", type=int, default=42)
    p.add_argument("--checkpoint_path", default=None)
    p.add_argument("--only_eval", action="store_true")
    p.add_argument("--learning_rate", type=float, default=7.5e-4)
    p.add_argument("--lr_decay_factor", type=float, default=0.999975)
    p.add_argument("--max_gradient_norm", type=float, default=0.25)
    p.add_argument("--encoding_dim", type=int, default=3)
    p.add_argument("--coupling_hidden_size", type=int, default=1024)
    p.add_argument("--coupling_hidden_layers", type=int, default=2)
    p.add_argument("--coupling_num_flows", type=int, default=1)
    p.add_argument("--coupling_num_mixtures", type=int, default=27)
    p.add_argument("--coupling_dropout", type=float, default=0.0)
    p.add_argument("--coupling_input_dropout", type=float, default=0.05)
    p.add_argument("--beta_scheduler_start_val", type=float, default=1.0)
    p.add_argument("--beta_scheduler_end_val", type=float, default=2.0)
    p.add_argument("--beta_scheduler_step_size", type=int, default=5000)
    p.add_argument("--beta_scheduler_logit", type=float, default=2.0)
    p.add_argument("--graph_step", action="store_true",
                   help="capture the training step (forward, HIP backward kernels, clipping, RAdam) in a HIP graph and replay it; "
                        "single process; beta and the learning rate live in device scalars; the LSTM runs on PyTorch's native "
                        "path (MIOpen's RNN calls are refused inside a stream capture)")
    p.add_argument("--graph_step_unverified", action="store_true",
                   help="with --graph_step: accept a captured step whose hipGraph could not be inspected for memset nodes "
                        "(a torch without CUDAGraph(keep_graph=True)); without it such a step is refused")
    p.add_argument("--backend", default=None, help="torch.distributed backend when started with WORLD_SIZE > 1 (nccl = RCCL)")
    p.add_argument("--share_device", action="store_true", help="TEST ONLY: every rank on cuda:0 (1-GPU box, --backend gloo)")
    return p.parse_args(argv)


def model_params(args):
    return {"max_seq_len": args.max_seq_len, "coupling_hidden_layers": args.coupling_hidden_layers,
            "coupling_hidden_size": args.coupling_hidden_size, "coupling_num_flows": args.coupling_num_flows,
            "coupling_num_mixtures": args.coupling_num_mixtures, "coupling_dropout": args.coupling_dropout,
            "coupling_input_dropout": args.coupling_input_dropout,
            "categ_encoding": {"use_dequantization": False, "use_variational": False, "use_decoder": False,
                               "num_dimensions": args.encoding_dim, "flow_config": {"num_flows": 0}, "decoder_config": {}}}


def beta_at(args, iteration):
    """ExponentialScheduler.get_val (general/parameter_scheduler.py:120-121)."""
    a, b = args.beta_scheduler_start_val, args.beta_scheduler_end_val
    return a + (b - a) * (1.0 - args.beta_scheduler_logit ** (-iteration * 1.0 / args.beta_scheduler_step_size))


def draw_batch(corpus, args, num, rng, device):
    """(tokens int64 [num, T], length int64 [num]); positions past a sentence's length hold symbol 0 (padding)."""
    x = corpus.sample(num, args.max_seq_len, rng)
    if args.variable_length:
        length = rng.randint(max(2, args.max_seq_len // 4), args.max_seq_len + 1, size=num)
        length[0] = args.max_seq_len                      # training batches keep one shape (no clipping, no reallocation)
        x[np.arange(args.max_seq_len)[None, :] >= length[:, None]] = 0
    else:
        length = np.full(num, args.max_seq_len)
    return torch.from_numpy(x).to(device), torch.from_numpy(length.astype(np.int64)).to(device)


def sentence_nll(model, prior, x, length, beta=1.0):
    """[B] negative log-likelihood per character, task.py:75-119 (`_train_batch_flow`, `_calc_loss`)."""
    x = x[:, :int(length.max())]                          # the batch is as wide as its longest sentence (task.py:123)
    inner = model.module if hasattr(model, "module") else model
    # (-sum_{n,d} log p(z) pad - ldj) / length, assembled by the flow pass itself: one kernel forward, one backward
    return model(x, reverse=False, beta=beta, length=length, _nll=inner.nll_request(length=length, prior=prior))[2]


@torch.no_grad()
def evaluate(model, prior, val, batch_size, rank=0, world=1):
    """Bits per character on the held-out sentences: every sentence weighs the same, like the reference's mean of
    per-sentence losses (task.py:107-113).  The sentences are sharded over the ranks; one all-reduce of (sum, count)."""
    inner = model.module if hasattr(model, "module") else model
    inner.eval()
    x_all, len_all = val
    lo, hi = shard_bounds(x_all.shape[0], rank, world)
    total = torch.zeros(2, dtype=torch.float64, device=x_all.device)
    for i in range(lo, hi, batch_size):
        j = min(i + batch_size, hi)
        total[0] += sentence_nll(inner, prior, x_all[i:j], len_all[i:j]).double().sum()
        total[1] += j - i
    mean_nll, bpc = allreduce_nll(total)
    inner.train()
    return bpc


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
    torch.manual_seed(args.seed)
    say = (lambda *a: print(*a, flush=True)) if rank == 0 else (lambda *a: None)
    corpus = MarkovCorpus(args.vocab_size, args.source_alpha, args.source_seed)
    say("synthetic source: %d symbols, entropy rate %.4f bits per character (context-free optimum %.4f)"
        % (args.vocab_size, corpus.entropy_rate(), corpus.unigram_entropy()))
    val = draw_batch(corpus, args, args.num_val, np.random.RandomState(123), device)      # fixed held-out sentences
    init_rng = np.random.RandomState(args.seed)                    # the same on every rank: replicas start identical
    rng = np.random.RandomState(args.seed + 1000 * rank + 1)       # every rank draws different training sentences
    per_rank = max(1, args.batch_size // world)

    class Vocab:
        vectors = None
    with contextlib.redirect_stdout(io.StringIO()):
        model = FlowLanguageModeling(model_params(args), None, vocab_size=args.vocab_size, vocab=Vocab()).to(device)
    prior = LogisticDistribution(mu=0.0, sigma=1.0)
    optimizer = torch.optim.RAdam(model.parameters(), lr=args.learning_rate)
    scheduler = torch.optim.lr_scheduler.LambdaLR(optimizer, lambda step: args.lr_decay_factor ** step)
    state = {"iteration": 0, "best_save_dict": {"file": None, "metric": 1e6, "detailed_metrics": None, "test": None},
             "evaluation_dict": {}}
    if args.checkpoint_path and os.path.exists(args.checkpoint_path):
        state.update(load_checkpoint(args.checkpoint_path, model, optimizer, scheduler, device=device))
    if state["iteration"] == 0 and not args.only_eval:
        init = []                                         # data-dependent ActNorm initialisation (general/task.py:112-128)
        for _ in range(8):
            x, length = draw_batch(corpus, args, args.batch_size, init_rng, device)
            init.append((x, {"length": length}))
        with contextlib.redirect_stdout(io.StringIO()):
            model.initialize_data_dependent(init)
    torch.manual_seed(args.seed + 1000 * rank + 1)        # dropout masks differ between the ranks (the replicas are initialised)
    ddp = wrap_ddp(model, device)
    if not args.only_eval and args.checkpoint_path and rank == 0:
        save_args(args.checkpoint_path, args)
    if args.only_eval:
        bpc = evaluate(ddp, prior, val, args.batch_size, rank, world)
        say("validation %.4f bits per character (source %.4f)" % (bpc, corpus.entropy_rate()))
        return {"val_bpc": bpc, "entropy_rate": corpus.entropy_rate()}

    graphed, eval_cudnn = None, torch.backends.cudnn.enabled
    if args.graph_step and world > 1:
        say("[#] --graph_step ignored: it is a single-process mode")
    elif args.graph_step:
        from ..graphs import GraphedTraining
        # MIOpen's RNN (hipBLASLt inside it) aborts under a stream capture; nn.LSTM's native per-time-step path is captured
        # instead — a few thousand small kernels, which is exactly what a graph replay is good at
        eval_cudnn = torch.backends.cudnn.enabled        # evaluation keeps the RNN path it would have without --graph_step
        torch.backends.cudnn.enabled = False
        lr_of = lambda step: args.learning_rate * args.lr_decay_factor ** step
        model.train()
        s_x, s_len = (t.clone() for t in draw_batch(corpus, args, per_rank, rng, device))      # training batches keep one shape
        s_noise = torch.rand(s_x.numel(), 1, args.encoding_dim, device=device)
        beta_t = torch.tensor(beta_at(args, state["iteration"]), dtype=torch.float32, device=device)      # the beta schedule lives in a device scalar
        graphed = GraphedTraining(model, lambda: model(s_x, reverse=False, beta=beta_t, length=s_len, noise=s_noise,
                                                       _nll=model.nll_request(length=s_len, prior=prior))[2].mean(),
                                  device, args.max_gradient_norm, lr=lr_of(state["iteration"]), eager_optimizer=optimizer, allow_unverified=args.graph_step_unverified)
        optimizer = graphed.optimizer
        say("[#] --graph_step: captured training step, hipGraph nodes %s" % (graphed.nodes,))
    ddp.train()
    best = state["best_save_dict"]
    t0, run_loss, seen = time.time(), torch.zeros((), device=device), 0
    for it in range(state["iteration"], args.max_iterations):
        x, length = draw_batch(corpus, args, per_rank, rng, device)
        if graphed is not None:
            s_x.copy_(x, non_blocking=True); s_len.copy_(length, non_blocking=True)
            s_noise.uniform_()
            beta_t.fill_(beta_at(args, it))
            loss = graphed(lr_of(it))
            scheduler.last_epoch, scheduler._last_lr = it + 1, [lr_of(it + 1)]      # what the checkpoint stores of the schedule
        else:
            loss = sentence_nll(ddp, prior, x, length, beta=beta_at(args, it)).mean()
            optimizer.zero_grad(set_to_none=True)
            loss.backward()
            torch.nn.utils.clip_grad_norm_(ddp.parameters(), args.max_gradient_norm)
            optimizer.step()
            scheduler.step()
        run_loss += loss.detach()
        seen += 1
        step = it + 1
        if step % args.print_freq == 0:
            say("iteration %7d | train %.4f bits per character (beta %.2f) | %.1f it/s"
                % (step, float(run_loss) / seen * LOG2E, beta_at(args, it), seen / (time.time() - t0)))
            t0, seen = time.time(), 0
            run_loss.zero_()
        if step % args.eval_freq == 0 or step == args.max_iterations:
            if graphed is not None:
                graphed.drop_weight_caches()
            with torch.backends.cudnn.flags(enabled=eval_cudnn):
                bpc = evaluate(ddp, prior, val, args.batch_size, rank, world)
            state["evaluation_dict"][step] = bpc
            say("iteration %7d | validation %.4f bits per character (source %.4f, context-free %.4f)"
                % (step, bpc, corpus.entropy_rate(), corpus.unigram_entropy()))
            if bpc < best["metric"] and args.checkpoint_path and rank == 0:
                if best["file"] and os.path.isfile(best["file"]):
                    os.remove(best["file"])
                best.update(file=checkpoint_file(args.checkpoint_path, step), metric=bpc, detailed_metrics={"val_bpc": bpc})
                with (graphed.checkpoint_groups(lr_of(step)) if graphed is not None else contextlib.nullcontext()):
                    save_checkpoint(args.checkpoint_path, step, ddp, optimizer, scheduler, best_save_dict=best,
                                    evaluation_dict=state["evaluation_dict"])
    if graphed is not None:
        graphed.drop_weight_caches()
    with torch.backends.cudnn.flags(enabled=eval_cudnn):
        bpc = evaluate(ddp, prior, val, args.batch_size, rank, world)
    say("final: validation %.4f bits per character; source entropy rate %.4f, context-free optimum %.4f"
        % (bpc, corpus.entropy_rate(), corpus.unigram_entropy()))
    if world > 1:
        torch.distributed.barrier()
    return {"val_bpc": bpc, "entropy_rate": corpus.entropy_rate(), "unigram_entropy": corpus.unigram_entropy(),
            "best_file": best["file"]}


if __name__ == "__main__":
    main()
