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
