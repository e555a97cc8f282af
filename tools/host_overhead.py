"""How long the host needs to enqueue one bench step (no sync inside the loop)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from categoricalnf_amd import ops
B, N, D = 16384, 64, 6
dev = torch.device("cuda:0")
z = torch.randn(B, N, D, device=dev); nn_out = torch.randn(B, N, 2 * D, device=dev)
sf = torch.zeros(D, device=dev); mask = torch.tensor([[1., 1., 1., 0., 0., 0.]], device=dev)
zo, lo = torch.empty_like(z), torch.empty(B, device=dev)
f = ops.affine_coupling_launch(z, nn_out, sf, mask, zo, lo)
for _ in range(10): f()
torch.cuda.synchronize()
for n in (1, 100, 1000):
    t0 = time.perf_counter()
    for _ in range(n): f()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("n=%d enqueue %.1f us/launch, total %.1f us/launch" % (n, (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6))
e = torch.cuda.Event(enable_timing=True)
t0 = time.perf_counter()
for _ in range(1000): e.record()
print("event.record %.1f us" % ((time.perf_counter() - t0) / 1000 * 1e6))
t0 = time.perf_counter()
for _ in range(1000): torch.cuda.current_stream(dev).cuda_stream
print("current_stream %.1f us" % ((time.perf_counter() - t0) / 1000 * 1e6))
s = torch.zeros(2, dtype=torch.float64, device=dev)
t0 = time.perf_counter()
for _ in range(1000): s.zero_()
print("zero_ %.1f us" % ((time.perf_counter() - t0) / 1000 * 1e6))
torch.cuda.synchronize()
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for p in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    if os.path.exists(p): print(p, open(p).read().strip())
os.system("lscpu | head -20")
