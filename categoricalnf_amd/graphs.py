"""hipGraph capture of a flow pass.

A forward / inverse pass of a flow at training-sized batches is a chain of 10-40 short kernels (tens of
microseconds each) driven from Python; the host, not the GPU, sets the pace.  Capturing the chain ONCE in a
HIP graph (through torch.cuda.CUDAGraph — our launches go to PyTorch's current stream, so they are
recorded like any other kernel) and replaying it removes the per-launch host cost.  This is plumbing only: the
graph replays exactly the kernels the eager pass would launch.

    runner = GraphedFlow(model, example_categories, reverse=False, length=length)   # warm-up + capture
    z, ldj = runner(categories)                                                     # replay

Constraints (as for any HIP graph): static shapes, inputs are copied into the captured buffers, no host
synchronisation inside the pass — the device-side NaN / range flag word is therefore read AFTER the replay.
"""
import contextlib

import torch
import torch.nn.functional as F

from . import ops


class GraphedFlow:
    """Capture `model(x, reverse=..., **kwargs)` (no autograd) and replay it on new inputs of the same shape."""

    def __init__(self, model, example_input, warmup=3, **kwargs):
        if not example_input.is_cuda:
            raise ops.HipOnlyError("GraphedFlow needs CUDA(HIP) tensors")
        self.model = model
        self.device = example_input.device
        self.static_in = example_input.clone()
        self.static_kwargs = {k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in kwargs.items()}
        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(warmup):                       # lazy initialisation, allocator warm-up, weight caches
                self._run()
        torch.cuda.current_stream(self.device).wait_stream(side)
        ops.check_flags(self.device, "GraphedFlow warm-up")
        self.graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(self.graph):
            self.static_out = self._run()

    def _run(self):
        ops.CAPTURING = True                              # FlowModel.forward must not sync on the flag word
        try:
            return self.model(self.static_in, **self.static_kwargs)
        finally:
            ops.CAPTURING = False

    def __call__(self, x, check=True, **kwargs):
        self.static_in.copy_(x)
        for k, v in kwargs.items():
            if isinstance(v, torch.Tensor):
                self.static_kwargs[k].copy_(v)
        self.graph.replay()
        if check:
            ops.check_flags(self.device, "GraphedFlow replay")
        return self.static_out


# ---- a captured TRAINING step --------------------------------------------------------------------------------------
# Round 2 removed its GraphedTrainStep: a 6000-step run stopped improving after ~2000 replays and the cause was not known.
# Round 3 found it (tools/graph_grad_diag.py, tools/graph_linear_probe.py; profiles/r03_graph_train_root_cause.txt): on this
# stack — torch 2.10 + rocm7.0, whose wheel bundles the HIP runtime 7.0.51831 that the whole process runs on (the image's ROCm 7.2
# runtime does not show the fault: tools/repro/hip_graph_memset_node.cpp) — a hipGraph MEMSET NODE of 16 B ... 4 KiB writes garbage from the second
# launch of the graph on (a captured hipMemsetAsync followed by `buf += 1` leaves 1 after the first replay and inf after
# every later one; 4-byte and 1-MiB memsets are fine).  PyTorch's own `sum` reduction zeroes the block semaphores of its
# two-pass ("global reduce") configuration with cudaMemsetAsync — that is how the BIAS gradient of every nn.Linear is
# taken — so from the second replay on every Linear bias gradient of a captured step is wrong (forward, weight gradients and
# every kernel of this library bit-equal to eager; one torch.nn.Linear alone reproduces it, with either BLAS back end).
# Nothing in this library's default path issues a memset (its zero fills are kernels).  Until the runtime is fixed a
# captured step must not contain such a reduction: `capture_safe_linear` takes the bias gradient in single-pass stages.

_orig_linear = F.linear


def _column_sums(m):
    """Column sums of a [M, N] matrix by reductions over at most 64 rows at a time.  PyTorch's reduce kernel goes to its
    two-pass configuration — the one with the memset — only when a thread has >= 256 values to add (ATen/native/cuda/
    Reduce.cuh, setReduceConfig); over <= 64 rows it is a single pass, in a fixed order: deterministic, and no memset node."""
    while m.shape[0] > 64:
        rows = m.shape[0]
        c = next((c for c in (64, 32, 48, 16, 24, 8, 12, 4, 6, 2, 3) if rows % c == 0), 0)
        if c == 0:                                   # no small divisor: zero rows up to the next multiple of 32
            pad = (-rows) % 32
            m = torch.cat([m, m.new_zeros(pad, m.shape[1])], 0)
            c = 32
        m = m.reshape(m.shape[0] // c, c, m.shape[1]).sum(1)
    return m.sum(0)


class _LinearNoReduce(torch.autograd.Function):
    """F.linear whose backward takes the bias gradient by `_column_sums` instead of dY.sum(0) (PyTorch's two-pass reduce
    kernel with a memset node in front of it).  Same values up to the order of the additions; deterministic."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return _orig_linear(x, weight, bias)

    @staticmethod
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        gy2 = gy.reshape(-1, gy.shape[-1])
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = (gy2 @ weight).reshape(x.shape)
        if ctx.needs_input_grad[1]:
            gw = gy2.t() @ x.reshape(-1, x.shape[-1])
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gb = _column_sums(gy2)
        return gx, gw, gb


def _safe_linear(input, weight, bias=None):
    if torch.is_grad_enabled() and (input.requires_grad or weight.requires_grad or (bias is not None and bias.requires_grad)):
        return _LinearNoReduce.apply(input, weight, bias)
    return _orig_linear(input, weight, bias)


@contextlib.contextmanager
def capture_safe_linear():
    """Inside: torch.nn.functional.linear (nn.Linear, nn.MultiheadAttention's projections) differentiates without
    PyTorch's two-pass sum.  Use it around BOTH the capture and any eager run that is to be compared with the replays."""
    prev = F.linear
    F.linear = _safe_linear
    try:
        yield
    finally:
        F.linear = prev


def graph_node_census(graph):
    """{node type name: count} of a torch.cuda.CUDAGraph created with keep_graph=True (its raw hipGraph_t is walked with
    hipGraphGetNodes / hipGraphNodeGetType through the HIP runtime the process already runs on).  What it is for: on the
    runtime the torch wheel bundles a MEMSET node is replayed wrongly (see above), so a captured step should hold none —
    GraphedTrainStep refuses one that does.  Returns None when the handle or the runtime entry points are not there."""
    import ctypes
    try:
        raw = int(graph.raw_cuda_graph())
        # the runtime the process ALREADY runs on (the wheel's bundled libamdhip64, not whatever the loader path finds):
        # the graph handle belongs to it
        loaded = [line.split()[-1] for line in open("/proc/self/maps") if "libamdhip64" in line]
        if not loaded:
            return None
        hip = ctypes.CDLL(loaded[0])
        count = ctypes.c_size_t(0)
        if hip.hipGraphGetNodes(ctypes.c_void_p(raw), None, ctypes.byref(count)) != 0:
            return None
        nodes = (ctypes.c_void_p * max(count.value, 1))()
        if hip.hipGraphGetNodes(ctypes.c_void_p(raw), nodes, ctypes.byref(count)) != 0:
            return None
        names = {0: "kernel", 1: "memcpy", 2: "memset", 3: "host", 4: "graph", 5: "empty"}
        census = {}
        for i in range(count.value):
            kind = ctypes.c_int(-1)
            if hip.hipGraphNodeGetType(ctypes.c_void_p(nodes[i]), ctypes.byref(kind)) != 0:
                return None
            name = names.get(kind.value, "type%d" % kind.value)
            census[name] = census.get(name, 0) + 1
        return census
    except Exception:
        return None


class GraphedTrainStep:
    """One training step — forward, backward, clipping, optimiser — captured in a HIP graph and replayed.

        step = GraphedTrainStep(lambda: train_step(model, optimizer, static_x, static_noise), device)
        for batch in data:
            static_x.copy_(batch); static_noise.uniform_()      # only copies into the static inputs between replays
            loss = step()                                        # replay; `loss` is the captured step's return value

    `step_fn` must follow the usual rules of a captured step: static input tensors, random numbers drawn OUTSIDE into
    static buffers, gradients through torch.autograd.grad (or .backward() with grads set to None before the capture), a
    `capturable` optimiser with its learning rate in a device tensor, no host synchronisation.  The warm-up runs on a
    side stream; capture and replays run with `capture_safe_linear` (see above for why), and the captured graph is refused if
    it still holds a memset node (`.nodes` is the census of its node types).  Accepted by the soak
    (tools/graph_train_soak.py: 3000 replays against an eager twin fed the same data) — the GPU suite runs a short one."""

    def __init__(self, step_fn, device, warmup=3, allow_memset_nodes=False):
        self.step_fn, self.device = step_fn, torch.device(device)
        main = torch.cuda.current_stream(self.device)
        side = torch.cuda.Stream(device=self.device)
        for _ in range(warmup):
            side.wait_stream(main)
            with torch.cuda.stream(side):
                self._run()
            main.wait_stream(side)
        torch.cuda.synchronize(self.device)
        ops.check_flags(self.device, "GraphedTrainStep warm-up")
        try:
            self.graph = torch.cuda.CUDAGraph(keep_graph=True)        # keeps the hipGraph_t so that its nodes can be counted
        except TypeError:
            self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.static_out = self._run()
        # a step that still contains a memset node would replay wrongly from its second launch on, silently: refuse it
        self.nodes = graph_node_census(self.graph)
        if self.nodes is None:
            # the one guard against the silent wrong-gradient fault could not run (a torch without keep_graph, libamdhip64 not
            # found in the process map, a HIP error): an UNVERIFIED step is not accepted silently
            self.nodes = "unverified"
            if not allow_memset_nodes:
                raise RuntimeError(
                    "the captured step's hipGraph could not be inspected (torch.cuda.CUDAGraph(keep_graph=True) / hipGraphGetNodes "
                    "unavailable), so it cannot be shown to be free of memset nodes, which this HIP runtime replays wrongly from the "
                    "second launch on (profiles/r03_graph_train_root_cause.txt).  Pass allow_memset_nodes=True (GraphedTraining: "
                    "allow_unverified=True; the drivers: --graph_step_unverified) to accept the step unverified (nodes == 'unverified').")
            import warnings
            warnings.warn("GraphedTrainStep: captured graph accepted WITHOUT a node census (allow_memset_nodes=True)")
        elif self.nodes.get("memset", 0) and not allow_memset_nodes:
            raise RuntimeError(
                "the captured step contains %d memset node(s) (%s): on the HIP runtime bundled with this torch wheel a hipGraph "
                "memset node is replayed wrongly from the second launch on (profiles/r03_graph_train_root_cause.txt).  They come "
                "from cudaMemsetAsync inside an op — PyTorch's two-pass sum over a long dimension does it; nn.Linear's bias gradient is "
                "covered by capture_safe_linear.  Reformulate the reduction in stages of <= 64 rows (graphs._column_sums) or pass "
                "allow_memset_nodes=True if the runtime at hand is known to be sound." % (self.nodes["memset"], self.nodes))

    def _run(self):
        ops.CAPTURING = True
        try:
            with capture_safe_linear():
                return self.step_fn()
        finally:
            ops.CAPTURING = False

    def __call__(self):
        self.graph.replay()
        return self.static_out


class GraphedTraining:
    """What a training driver needs around a captured step (`--graph_step` of the three experiment drivers): the model's
    training step — `loss_fn()` on static inputs, gradients, clipping, RAdam — captured once by GraphedTrainStep and replayed,
    with the learning rate in a device scalar (`.lr`, written before every replay), the moments of the driver's eager optimiser
    carried over (a resumed run), and the warm-up steps the capture needs undone afterwards (parameters and moments).

        run = GraphedTraining(model, lambda: model(static_x, ..., _nll=model.nll_request(length=static_ln))[2].mean(),
                              device, max_grad_norm, eager_optimizer=optimizer)
        for it in ...:
            static_x.copy_(x); static_noise.uniform_()
            loss = run(lr_of(it))

    `.optimizer` is the capturable optimiser (for checkpoints: `checkpoint_groups(lr)` switches its param_groups to what an
    eager run stores and back); `drop_weight_caches()` must be called before the model runs in eval mode (replays do not run
    the modules' Python, so the eval-mode caches of the 1x1 convolutions are not dropped by a training forward any more)."""

    def __init__(self, model, loss_fn, device, max_grad_norm, lr=7.5e-4, eager_optimizer=None, optimizer_cls=torch.optim.RAdam,
                 allow_unverified=False):
        # allow_unverified: accept a captured step whose hipGraph could not be inspected for memset nodes (GraphedTrainStep's
        # allow_memset_nodes; the drivers' --graph_step_unverified) — without it such a step is refused
        self.model, self.device = model, torch.device(device)
        self.lr = torch.tensor(float(lr), dtype=torch.float32, device=self.device)       # the schedule lives in a device scalar
        eager_state = eager_optimizer.state_dict()["state"] if eager_optimizer is not None else {}
        # one multi-tensor kernel per optimiser phase instead of five elementwise kernels per parameter tensor (a GraphCNF has
        # 1 740 of them: the per-tensor form was a fifth of the captured step's kernel time): the fused implementation where the
        # class has one (Adam / AdamW), the foreach one otherwise (RAdam)
        self.optimizer = None
        for extra in ({"fused": True}, {"foreach": True}, {}):
            try:
                self.optimizer = optimizer_cls(model.parameters(), lr=self.lr, capturable=True, **extra)
                break
            except (TypeError, RuntimeError, ValueError):
                continue
        if eager_state:                                   # resumed: carry the moments over (step counters move to the device)
            sd = self.optimizer.state_dict()
            sd["state"] = {k: {n: (v.to(device=self.device, dtype=torch.float32) if n == "step" else v) for n, v in st.items()}
                           for k, st in eager_state.items()}
            self.optimizer.load_state_dict(sd)
            for group in self.optimizer.param_groups:
                group["lr"], group["capturable"] = self.lr, True
        model.train()
        plist = [p for p in model.parameters() if p.requires_grad]
        optimizer = self.optimizer

        def train_step():
            loss = loss_fn()
            for p, g in zip(plist, torch.autograd.grad(loss, plist, allow_unused=True)):
                # what AccumulateGrad does for `backward()`: a gradient laid out like its parameter (the multi-tensor optimiser
                # kernels refuse lists whose members differ in strides)
                if g is not None and (g.stride() != p.stride() or g.dtype != p.dtype):
                    g = torch.empty_like(p).copy_(g)
                p.grad = g
            torch.nn.utils.clip_grad_norm_(plist, max_grad_norm, foreach=True)
            optimizer.step()
            return loss.detach()
        for p in plist:
            p.grad = None
        snapshot = [p.detach().clone() for p in plist], {k: {n: v.clone() for n, v in st.items() if torch.is_tensor(v)}
                                                         for k, st in optimizer.state.items()}
        self.step = GraphedTrainStep(train_step, self.device, allow_memset_nodes=allow_unverified)
        self.nodes = self.step.nodes
        # the warm-up steps before the capture trained on one batch: undo them (parameters and moments)
        with torch.no_grad():
            for p, old in zip(plist, snapshot[0]):
                p.copy_(old)
            for k, st in optimizer.state.items():
                for n, v in st.items():
                    if torch.is_tensor(v):
                        if k in snapshot[1] and n in snapshot[1][k]:
                            v.copy_(snapshot[1][k][n])
                        else:
                            v.zero_()

    def __call__(self, lr=None):
        if lr is not None:
            self.lr.fill_(float(lr))
        return self.step()

    def drop_weight_caches(self):
        for m in self.model.modules():
            if hasattr(m, "_empty_eval_dict"):
                m._empty_eval_dict()

    @contextlib.contextmanager
    def checkpoint_groups(self, lr):
        """Inside: the optimiser's param_groups read as an eager run stores them — the learning rate as a number, no capturable
        flag (an eager resume must not inherit the device-side step counters' mode)."""
        for group in self.optimizer.param_groups:
            group["lr"], group["capturable"] = float(lr), False
        # ... and the step counters as an eager (non-capturable) optimiser keeps them: CPU scalars.  Left on the device, an eager
        # resume would call step.item() — a host sync — for every parameter on every step.
        device_steps = {}
        for p_, st in self.optimizer.state.items():
            if torch.is_tensor(st.get("step")) and st["step"].is_cuda:
                device_steps[p_] = st["step"]
                st["step"] = st["step"].detach().to("cpu")
        try:
            yield self.optimizer
        finally:
            for p_, t in device_steps.items():
                self.optimizer.state[p_]["step"] = t
            for group in self.optimizer.param_groups:
                group["lr"], group["capturable"] = self.lr, True
