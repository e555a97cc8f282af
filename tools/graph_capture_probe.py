"""Can the training step of the language-modelling (LSTM sub-network, MIOpen) and graph-colouring (RGCN) flows be captured in a HIP
graph?  Builds each model as its driver does (small sizes), captures one step (encoder noise in a static buffer, beta fixed),
prints the census of the graph's nodes, and compares the losses of replays with an eager twin fed the same data.
    python tools/graph_capture_probe.py [lm] [gc]"""
import contextlib, copy, io, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from categoricalnf_amd.graphs import GraphedTrainStep, capture_safe_linear
from categoricalnf_amd.layers.flows.distributions import LogisticDistribution

dev = torch.device("cuda", 0)
which = sys.argv[1:] or ["lm", "gc"]


def run(name, model, loss_of, statics, refresh, steps=30):
    twin = copy.deepcopy(model)
    plist = [p for p in model.parameters() if p.requires_grad]
    tl = [p for p in twin.parameters() if p.requires_grad]
    opt = torch.optim.RAdam(plist, lr=torch.tensor(7.5e-4, device=dev), capturable=True)
    opt_t = torch.optim.RAdam(tl, lr=torch.tensor(7.5e-4, device=dev), capturable=True)

    def make_step(m, pl, o):
        def step():
            loss = loss_of(m)
            for p, g in zip(pl, torch.autograd.grad(loss, pl, allow_unused=True)):
                p.grad = g
            torch.nn.utils.clip_grad_norm_(pl, 0.25, foreach=True)
            o.step()
            return loss.detach()
        return step
    snap = [p.detach().clone() for p in plist]
    try:
        graphed = GraphedTrainStep(make_step(model, plist, opt), dev)
    except Exception as e:
        import traceback
        print("%s: capture FAILED: %s: %s" % (name, type(e).__name__, str(e)[:200]))
        print("".join(traceback.format_tb(e.__traceback__)[-8:]))
        return
    print("%s: captured, nodes %s" % (name, graphed.nodes))
    with torch.no_grad():
        for p, old in zip(plist, snap):
            p.copy_(old)
        for st in opt.state.values():
            for v in st.values():
                if torch.is_tensor(v):
                    v.zero_()
    eager = make_step(twin, tl, opt_t)
    worst = 0.0
    for i in range(steps):
        refresh(i)
        a = graphed().clone()
        with capture_safe_linear():
            b = eager()
        worst = max(worst, abs(float(a) - float(b)))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        refresh(i)
        graphed()
    torch.cuda.synchronize()
    tg = (time.perf_counter() - t0) / steps
    t0 = time.perf_counter()
    for i in range(steps):
        refresh(i)
        with capture_safe_linear():
            eager()
    torch.cuda.synchronize()
    te = (time.perf_counter() - t0) / steps
    print("%s: %d replays against the eager twin: largest loss difference %.3g (loss %.4f); %.2f ms per replay, %.2f ms per eager step"
          % (name, steps, worst, float(a), tg * 1e3, te * 1e3))


if "lm" in which:
    from categoricalnf_amd.experiments import run_language_modeling as L
    args = L.parse(["--max_seq_len", "64", "--batch_size", "32", "--coupling_hidden_size", "256"])
    torch.manual_seed(0)

    class Vocab:
        vectors = None
    with contextlib.redirect_stdout(io.StringIO()):
        model = L.FlowLanguageModeling(L.model_params(args), None, vocab_size=args.vocab_size, vocab=Vocab()).to(dev)
    corpus = L.MarkovCorpus(args.vocab_size, args.source_alpha, args.source_seed)
    rng = np.random.RandomState(0)
    with contextlib.redirect_stdout(io.StringIO()):
        model.initialize_data_dependent([(lambda b: (b[0], {"length": b[1]}))(L.draw_batch(corpus, args, 32, rng, dev)) for _ in range(2)])
    model.train()
    x, ln = L.draw_batch(corpus, args, 32, rng, dev)
    static_x, static_noise = x.clone(), torch.rand(x.numel(), 1, args.encoding_dim, device=dev)
    pool = [L.draw_batch(corpus, args, 32, rng, dev)[0] for _ in range(4)]

    def refresh(i):
        static_x.copy_(pool[i % 4]); static_noise.uniform_()
    if os.environ.get("CNF_NATIVE_LSTM", "1") == "1":
        torch.backends.cudnn.enabled = False          # nn.LSTM's native path: MIOpen's RNN calls are refused by a stream capture
    run("language modelling (LSTM)", model,
        lambda m: m(static_x, reverse=False, beta=1.5, length=ln, noise=static_noise, _nll=m.nll_request(length=ln))[2].mean(), None, refresh)

if "gc" in which:
    from categoricalnf_amd.experiments import run_graph_coloring as G
    from categoricalnf_amd.experiments.graph_coloring import GraphNodeFlow
    from categoricalnf_amd.experiments.graph_coloring_data import GraphColoringDataset, generate_planted_dataset
    import tempfile
    root = tempfile.mkdtemp()
    GraphColoringDataset.set_dataset(prefix="_tiny", num_colors=3)
    GraphColoringDataset.DATASET_NODES = GraphColoringDataset.DATASET_VAL_IDX = None
    generate_planted_dataset(root, prefix="_tiny", num_colors=3, num_graphs=2000, n_min=10, n_max=20, seed=0)
    train = GraphColoringDataset(num_colors=3, train=True, data_root=root)
    args = G.parse(["--dataset", "tiny_3", "--batch_size", "128"])
    torch.manual_seed(0)
    with contextlib.redirect_stdout(io.StringIO()):
        model = GraphNodeFlow(G.model_params(args), GraphColoringDataset).to(dev)

    def full(idx):
        items = [train[i] for i in idx]
        return (torch.from_numpy(np.stack([it[0] for it in items])).to(dev), torch.from_numpy(np.stack([it[1] for it in items])).to(dev),
                torch.from_numpy(np.array([it[2] for it in items], dtype=np.int64)).to(dev))
    rng = np.random.RandomState(0)
    init = []
    for _ in range(4):
        x, adj, ln = full(rng.randint(0, len(train), size=128))
        init.append((x, {"length": ln, "adjacency": adj}))
    with contextlib.redirect_stdout(io.StringIO()):
        model.initialize_data_dependent(init)
    model.train()
    pool = [full(rng.randint(0, len(train), size=128)) for _ in range(4)]
    sx, sadj, sln = (t.clone() for t in pool[0])
    static_noise = torch.rand(sx.numel(), 1, model.embed_dim, device=dev)

    def refresh(i):
        x, adj, ln = pool[i % 4]
        sx.copy_(x); sadj.copy_(adj); sln.copy_(ln); static_noise.uniform_()
    run("graph colouring (RGCN)", model,
        lambda m: m(sx, sadj, reverse=False, beta=1.5, length=sln, noise=static_noise, _nll=m.nll_request(length=sln))[2].mean(), None, refresh)
