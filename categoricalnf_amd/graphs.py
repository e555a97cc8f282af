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
    """One whole training step — forward, NLL, backward (HIP backward kernels), gradient clipping, optimiser update —
    captured in ONE HIP graph and replayed per batch.

    At the reference's batch sizes (64-512 sets) a step is ~10^3 short kernels and the Python host paces it; the
    replay removes the host from the loop.  Requirements: static batch shape, a `capturable` optimiser (the learning
    rate lives in a device tensor, `set_lr` changes it between replays), single GPU (DDP's bucket hooks are not
    captured here); capture BEFORE the model's parameters take part in any eager backward, or on a fresh module
    with the same state_dict (parameters that were differentiated eagerly keep AccumulateGrad nodes bound to the
    default stream; PyTorch warns "AccumulateGrad node's stream does not match" and the gradient accumulation then
    happens outside the graph), and NO other autograd backward may run
    in the process between replays — on ROCm 7.2 / PyTorch 2.10 an eager `backward()` of any module (a plain
    nn.Sequential suffices) between two replays corrupts the captured step's gradients; evaluation under
    `torch.no_grad()` between replays is fine.  The warm-up iterations are real optimisation steps on the example
    batch.  With these rules the replayed trajectory is bit-identical to the eager one (tests).  The encoder's noise comes from PyTorch's device generator, which CUDA/HIP graphs advance
    correctly between replays.

        step = GraphedTrainStep(model, lambda params: torch.optim.RAdam(params, lr=torch.tensor(7.5e-4), capturable=True),
                                example_x, example_length, max_grad_norm=0.25)
        loss = step(x, length)             # device scalar (mean NLL per element)
    """

    def __init__(self, model, make_optimizer, example_x, example_length, max_grad_norm=0.25, warmup=3, noise_shape=None,
                 math_attention=False, **kwargs):
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
        # encoder noise: drawn OUTSIDE the graph into a static buffer before every replay and handed to the model as
        # `noise=` (one small eager kernel per step).  Drawing it inside the capture relies on the generator's
        # graph-safe offset bookkeeping, which on this stack returned unusable draws when the process had used the
        # generator eagerly before the capture (losses of 10-60 instead of 3).
        self.static_noise = None
        if noise_shape is not None:
            self.static_noise = torch.rand(tuple(noise_shape), dtype=torch.float32, device=self.device)
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
                    self.static_noise.uniform_()
                self._step()
        torch.cuda.current_stream(self.device).wait_stream(side)
        ops.check_flags(self.device, "GraphedTrainStep warm-up")
        self.graph = torch.cuda.CUDAGraph()
        self.optimizer.zero_grad(set_to_none=True)
        with torch.cuda.graph(self.graph):
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
            self.optimizer.zero_grad(set_to_none=True)
            loss.backward()
            if self.max_grad_norm is not None:
                torch.nn.utils.clip_grad_norm_(self.model.parameters(), self.max_grad_norm, foreach=True)
            self.optimizer.step()
            return loss.detach()
        finally:
            pass

    def set_lr(self, value):
        for grp in self.optimizer.param_groups:
            grp["lr"].fill_(float(value))

    def __call__(self, x, length=None):
        if self.static_noise is not None:
            self.static_noise.uniform_()
        self.static_x.copy_(x)
        if length is not None:
            self.static_len.copy_(length)
        self.graph.replay()
        return self.static_loss
