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
import torch

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


class GraphedTrainStep:
    """EXPERIMENTAL.  One whole training step — forward, NLL, backward (HIP backward kernels), gradient clipping,
    optimiser update, learning-rate decay — captured in ONE HIP graph and replayed per batch.

    At the reference's batch sizes (64-512 sets) a step is ~10^3 short kernels and the Python host paces it; a
    replay takes the host out of the loop: 2.0x at batch 64, 1.5x at 256 for the default set-modelling flow, nothing
    from batch 1024 up (tools/bench_train_step.py).  Over short horizons the replayed optimisation trajectory equals
    the eager one (tests: 6-10 steps to 1e-3; 4000 steps of RAdam with nothing else touching the graph's tensors).

    Status on this stack (ROCm 7.2, PyTorch 2.10): NOT reliable for long runs, which is why the training driver does
    not use it.  What was found while chasing wrong replays, and how the class works around it:
      * in-graph `torch.rand` after eager use of the generator returned unusable draws -> the encoder noise is drawn
        OUTSIDE the graph into a static buffer (`noise_shape`) from a generator of its own;
      * `loss.backward()` inside the capture leaves gradient accumulation on AccumulateGrad nodes bound to another
        stream, an unordered side branch of the graph (NaNs whose onset depended on host timing) -> gradients are
        taken with `torch.autograd.grad` and assigned;
      * an eager `backward()` of ANY module between two replays corrupts the captured gradients (a plain
        nn.Sequential twin suffices); evaluation under `torch.no_grad()` in between is fine;
      * capture before the module's parameters take part in an eager backward, or on a fresh module with the same
        state_dict;
      * eager kernels that touch graph-referenced tensors between replays (an lr `fill_`, a `loss_sum +=`) made
        2000+ step runs blow up -> the schedule (`lr_decay`, `lr_minimum`) and the running loss (`pop_loss_sum`) live
        inside the graph;
      * even so, in a 6000-step run of the default flow the loss stopped improving after ~2000 replays where the
        eager run kept falling.  The cause was not found (suspected: ordering between successive launches of one
        executable graph); treat results of long replay runs as unverified.
    Requirements: static batch shape, a `capturable` optimiser, single GPU.  The warm-up iterations are real
    optimisation steps on the example batch.

        step = GraphedTrainStep(model, lambda params: torch.optim.RAdam(params, lr=torch.tensor(7.5e-4), capturable=True),
                                example_x, example_length, max_grad_norm=0.25, noise_shape=(B * N, 1, D), beta=1)
        loss = step(x, length)             # device scalar (mean NLL per element)
    """

    def __init__(self, model, make_optimizer, example_x, example_length, max_grad_norm=0.25, warmup=3, noise_shape=None,
                 math_attention=False, lr_decay=1.0, lr_minimum=0.0, **kwargs):
        from . import functional as Fn
        if not example_x.is_cuda:
            raise ops.HipOnlyError("GraphedTrainStep needs CUDA(HIP) tensors")
        self.model = model
        self.device = example_x.device
        self.max_grad_norm = max_grad_norm
        self.static_x = example_x.clone()
        self.static_len = example_length.clone()
        self.kwargs = dict(kwargs)
        self.math_attention = math_attention
        # the learning-rate schedule (lr <- max(lr * lr_decay, lr_minimum) per step) and the running loss sum live INSIDE
        # the graph: eager kernels that touch graph-referenced tensors between two replays (an lr fill_, a loss +=)
        # made long runs go wrong on this stack, so nothing but the input copies happens between replays
        self.lr_decay, self.lr_minimum = float(lr_decay), float(lr_minimum)
        self.loss_sum = torch.zeros((), dtype=torch.float32, device=self.device)
        self.max_in_flight = 4
        self._in_flight = []
        # encoder noise: drawn OUTSIDE the graph into a static buffer before every replay and handed to the model as
        # `noise=` (one small eager kernel per step).  Drawing it inside the capture relies on the generator's
        # graph-safe offset bookkeeping, which on this stack returned unusable draws when the process had used the
        # generator eagerly before the capture (losses of 10-60 instead of 3).
        self.static_noise = None
        if noise_shape is not None:
            # a generator of its own: the default one is registered with the capture machinery
            self.noise_generator = torch.Generator(device=self.device)
            self.noise_generator.manual_seed(torch.initial_seed() % (2 ** 63 - 1) + 1)
            self.static_noise = torch.rand(tuple(noise_shape), dtype=torch.float32, device=self.device,
                                           generator=self.noise_generator)
            self.kwargs["noise"] = self.static_noise
        for p in model.parameters():
            p.grad = None
        self.optimizer = make_optimizer(model.parameters())
        for grp in self.optimizer.param_groups:
            if not isinstance(grp["lr"], torch.Tensor):
                grp["lr"] = torch.tensor(float(grp["lr"]), device=self.device)
            elif grp["lr"].device != self.device:
                grp["lr"] = grp["lr"].to(self.device)
        self._Fn = Fn
        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(side):
            for _ in range(warmup):                        # optimiser state, allocator pools, lazy caches
                if self.static_noise is not None:
                    self.static_noise.uniform_(generator=self.noise_generator)
                self._step()
        torch.cuda.current_stream(self.device).wait_stream(side)
        ops.check_flags(self.device, "GraphedTrainStep warm-up")
        self.graph = torch.cuda.CUDAGraph()
        self.optimizer.zero_grad(set_to_none=True)
        # capture on the SAME stream the warm-up ran on: gradient-accumulation nodes created during the warm-up are
        # bound to that stream, and on a different capture stream they run as a forked branch whose buffers are
        # not part of the graph's private pool (replays then read gradients that later eager allocations overwrite —
        # training went wrong after several hundred replays)
        with torch.cuda.graph(self.graph, stream=side):
            self.static_loss = self._step()

    def _step(self):
        ops.CAPTURING = True
        try:
            if self.math_attention:
                # optional: capture the unfused attention kernels instead of the flash / memory-efficient ones
                from torch.nn.attention import SDPBackend, sdpa_kernel
                with sdpa_kernel(SDPBackend.MATH):
                    return self._step_body()
            return self._step_body()
        finally:
            ops.CAPTURING = False

    def _step_body(self):
        try:
            z, ldj = self.model(self.static_x, reverse=False, length=self.static_len, **self.kwargs)
            loss = self._Fn.PriorNllFn.apply(z, ldj, self.static_len, None).mean()
            # gradients through torch.autograd.grad, not loss.backward(): no AccumulateGrad nodes (they are bound to the
            # stream on which they were first created and ran as an unordered side branch of the captured graph —
            # replays then raced on the .grad buffers: wrong steps whose onset depended on host timing)
            params = [p for p in self.model.parameters() if p.requires_grad]
            grads = torch.autograd.grad(loss, params, allow_unused=True)
            for p, g in zip(params, grads):
                p.grad = g
            if self.max_grad_norm is not None:
                torch.nn.utils.clip_grad_norm_(self.model.parameters(), self.max_grad_norm, foreach=True)
            self.optimizer.step()
            if self.lr_decay != 1.0:
                for grp in self.optimizer.param_groups:
                    grp["lr"].mul_(self.lr_decay).clamp_(min=self.lr_minimum)
            self.loss_sum += loss.detach()
            return loss.detach()
        finally:
            pass

    def pop_loss_sum(self):
        """Sum of the losses of the replays since the last call (one host sync)."""
        total = float(self.loss_sum)
        self.loss_sum.zero_()
        return total

    def set_lr(self, value):
        for grp in self.optimizer.param_groups:
            grp["lr"].fill_(float(value))

    def __call__(self, x, length=None):
        if self.static_noise is not None:
            self.static_noise.uniform_(generator=self.noise_generator)
        self.static_x.copy_(x)
        if length is not None:
            self.static_len.copy_(length)
        self.graph.replay()
        # bound the number of replays in flight: the host enqueues a replay in ~0.1 ms while the GPU needs 10+ ms for
        # it, and several hundred queued launches of the same executable graph ended in corrupted steps on this stack
        # (training blew up after 850-950 unsynchronised replays, independent of seed, data and optimiser)
        ev = torch.cuda.Event()
        ev.record()
        self._in_flight.append(ev)
        if len(self._in_flight) > self.max_in_flight:
            self._in_flight.pop(0).synchronize()
        return self.static_loss
