"""Why does a HIP-graph replay of a training step produce wrong nn.Linear BIAS gradients on this stack (tools/
graph_grad_diag.py: forward bit-equal, weight gradients equal, 44 bias gradients off by 100x from the second replay on)?
The smallest reproducer: one torch.nn.Linear (no kernel of this library), forward + backward captured, replayed on fresh
inputs, against eager; 2-D and 3-D inputs, both BLAS back ends, and the kernels autograd launches for the bias gradient."""
import os, sys
import torch
import torch.nn.functional as F
dev = torch.device("cuda", 0)
print("torch", torch.__version__, "hip", torch.version.hip, "TORCH_BLAS_PREFER_HIPBLASLT =", os.environ.get("TORCH_BLAS_PREFER_HIPBLASLT"), flush=True)


def probe(shape, out_f, lib, bias_via_sum=False):
    try:
        torch.backends.cuda.preferred_blas_library(lib)
    except Exception as e:
        return "preferred_blas_library(%s): %s" % (lib, e)
    torch.manual_seed(0)
    lin = torch.nn.Linear(shape[-1], out_f).to(dev)
    x = torch.randn(*shape, device=dev)
    w = torch.randn(*shape[:-1], out_f, device=dev)

    def step(inp):
        if bias_via_sum:
            y = F.linear(inp, lin.weight) + lin.bias
        else:
            y = lin(inp)
        return torch.autograd.grad((torch.tanh(y) * w).sum(), [lin.weight, lin.bias])
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        for _ in range(3):
            step(x)
    torch.cuda.current_stream(dev).wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        gw, gb = step(x)
    out = []
    for r in range(4):
        x.normal_()
        ew, eb = step(x)
        g.replay()
        torch.cuda.synchronize()
        out.append("r%d dW %.1e db %.1e" % (r, float((ew - gw).abs().max() / ew.abs().max()), float((eb - gb).abs().max() / eb.abs().max())))
    return " | ".join(out)


for lib in ("hipblaslt", "cublas"):
    for shape in ((1024, 256), (64, 16, 256), (64, 16, 32)):
        for out_f in (256, 1024):
            print("%-10s x%-16s out %4d  module: %s" % (lib, shape, out_f, probe(shape, out_f, lib)), flush=True)
    print("%-10s x%-16s out %4d  x @ W^T + b: %s" % (lib, (64, 16, 256), 256, probe((64, 16, 256), 256, lib, bias_via_sum=True)), flush=True)

# which kernels compute the bias gradient
from torch.profiler import profile, ProfilerActivity
torch.backends.cuda.preferred_blas_library("hipblaslt")
lin = torch.nn.Linear(256, 256).to(dev)
x = torch.randn(64, 16, 256, device=dev)
for _ in range(2):
    torch.autograd.grad(lin(x).sum(), [lin.weight, lin.bias])
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    torch.autograd.grad(lin(x).sum(), [lin.weight, lin.bias])
    torch.cuda.synchronize()
for e in prof.events():
    if e.device_type is not None and "cuda" in str(e.device_type).lower():
        print("   kernel:", e.name[:150])

# ---- is it the memset node?  hipMemsetAsync captured in front of `buf += 1`: every replay must leave 1 in the buffer
import ctypes
hip = ctypes.CDLL("libamdhip64.so")
hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
for nbytes in (4, 16, 64, 256, 4096, 1 << 20):
    buf = torch.full((max(nbytes // 4, 1),), 7.0, device=dev)
    one = torch.ones_like(buf)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        st = torch.cuda.current_stream(dev).cuda_stream
        rc = hip.hipMemsetAsync(buf.data_ptr(), 0, nbytes, st)
        buf.add_(one)
    vals = []
    for r in range(4):
        g.replay()
        torch.cuda.synchronize()
        vals.append("%.0f/%.0f" % (float(buf.min()), float(buf.max())))
    print("memset node of %8d bytes + (buf += 1), rc %d: buffer min/max after replays %s   (expected 1/1 every time)" % (nbytes, rc, " ".join(vals)), flush=True)
# the same with the memset in the middle of a chain whose earlier node wrote the buffer (ordering inside the graph)
buf = torch.zeros(64, device=dev); one = torch.ones_like(buf)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    st = torch.cuda.current_stream(dev).cuda_stream
    buf.add_(one); buf.add_(one)
    hip.hipMemsetAsync(buf.data_ptr(), 0, 256, st)
    buf.add_(one)
vals = []
for r in range(4):
    g.replay(); torch.cuda.synchronize(); vals.append("%.0f" % float(buf.max()))
print("add, add, memset, add: buffer after replays %s (expected 1 every time)" % " ".join(vals), flush=True)
