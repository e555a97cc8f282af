"""What does a launch of the bench's dominant kernel cost, and what do per-launch timestamps say?  The affine forward + NLL
+ batch-sum kernel at B=16384,N=64,D=6 on 4 rotating buffer sets, the inverse, and the stream-probe kernel (same traffic,
no arithmetic), each in three streams:
  (a) bursts: 100 launches, host sync, repeat (what tools/sweep_nll.py times),
  (b) sustained: thousands of launches back to back,
  (c) the bench step: forward (or the probe in its place) alternating with the inverse,
measured twice: per-launch dispatch timestamps (cnf_prof_arm; every 8th launch, or the last 8 of a burst) and
start-to-start times from one event pair around a block of launches.  Findings (profiles/HISTORY.md section 4): consecutive
launches overlap, so timestamps in a continuous stream exceed the stream's advance per launch; there is no burst-versus-
sustained clock effect; in the bench step the inverse runs faster and the forward slower than in streams of their own."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from categoricalnf_amd import _lib, ops
dev = torch.device("cuda:0")
lib = _lib.load()
B, N, D, R = int(os.environ.get("CNF_PROBE_B", 16384)), 64, 6, int(os.environ.get("CNF_PROBE_R", 4))
g = torch.Generator(device=dev).manual_seed(0)
zs = [torch.randn(B, N, D, generator=g, device=dev) for _ in range(R)]
nns = [0.5 * torch.randn(B, N, 2 * D, generator=g, device=dev) for _ in range(R)]
sf = torch.zeros(D, device=dev); mask = torch.tensor([[1., 1., 1., 0., 0., 0.]], device=dev)
zf = [torch.empty_like(zs[0]) for _ in range(R)]; zr = [torch.empty_like(zs[0]) for _ in range(R)]
lf = [torch.empty(B, device=dev) for _ in range(R)]; lr = [torch.empty(B, device=dev) for _ in range(R)]
ln = torch.full((B,), float(N), device=dev)
neglog, nll = torch.empty(B, device=dev), torch.empty(B, device=dev)
acc = torch.zeros(ops.NLL_ACC_SLOTS, dtype=torch.int64, device=dev)
fwd = [ops.affine_coupling_nll_acc_launch(zs[r], nns[r], sf, mask, zf[r], lf[r], ln, neglog, nll, acc) for r in range(R)]
inv = [ops.affine_coupling_launch(zf[r], nns[r], sf, mask, zr[r], lr[r], reverse=True) for r in range(R)]
elems = B * N * D


def probe(r):
    ops._launch(dev, "cnf_stream_probe", zs[r].data_ptr(), nns[r].data_ptr(), zr[r].data_ptr(), elems, 2, ops._stream(dev))


def collect(n):
    buf = (ctypes.c_float * n)()
    got = lib.cnf_prof_collect(buf, n)
    return np.array([buf[i] for i in range(got)]) * 1e3


def bursts(launch, rounds=20):
    out = []
    for _ in range(rounds):
        for i in range(92):
            launch(i % R)
        lib.cnf_prof_arm(8)
        for i in range(8):
            launch(i % R)
        torch.cuda.synchronize()
        out.append(collect(8))
    return np.concatenate(out[2:])


def sustained(launch, between=None, n=6000, every=8):
    out = []
    for i in range(n):
        if i % every == 0:
            lib.cnf_prof_arm(1)
        launch(i % R)
        if between is not None:
            between(i % R)
    torch.cuda.synchronize()
    out.append(collect(4096))
    return np.concatenate(out)


def show(name, v):
    q = len(v) // 4
    print("%-46s n=%4d  mean %6.2f us | first quarter %6.2f | last quarter %6.2f | min %6.2f" %
          (name, len(v), v.mean(), v[:q].mean(), v[-q:].mean(), v.min()), flush=True)


def s2s_bursts(launch, rounds=20, n=100):
    """start-to-start time from one event pair around each burst (no per-launch timestamps involved)"""
    out = []
    for _ in range(rounds):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for i in range(n):
            launch(i % R)
        b.record()
        torch.cuda.synchronize()
        out.append(a.elapsed_time(b) / n * 1e3)
    return np.array(out[2:])


def s2s_sustained(launch, between=None, blocks=30, n=200):
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(blocks + 1)]
    marks[0].record()
    for k in range(blocks):
        for i in range(n):
            launch(i % R)
            if between is not None:
                between(i % R)
        marks[k + 1].record()
    torch.cuda.synchronize()
    return np.array([marks[k].elapsed_time(marks[k + 1]) / n * 1e3 for k in range(blocks)])


for i in range(300):
    fwd[i % R]()
torch.cuda.synchronize()
print("dispatch-timestamp durations (cnf_prof_arm); a launch's timestamps overlap the drain of the launch before it")
show("(a) forward+NLL, bursts of 100", bursts(lambda r: fwd[r]()))
show("(b) forward+NLL, sustained forward-only", sustained(lambda r: fwd[r]()))
show("(c) forward+NLL, sustained fwd/inv alternating", sustained(lambda r: fwd[r](), between=lambda r: inv[r]()))
show("(a) stream probe, bursts of 100", bursts(probe))
show("(b) stream probe, sustained", sustained(probe))
show("(c) stream probe alternating with the inverse", sustained(probe, between=lambda r: inv[r]()))

print("start-to-start times (one event pair around a block of launches: throughput, no per-launch timestamps)")
show("(a) forward+NLL, bursts of 100", s2s_bursts(lambda r: fwd[r]()))
show("(b) forward+NLL, sustained 30 x 200", s2s_sustained(lambda r: fwd[r]()))
show("(c) forward+NLL + inverse (one bench step)", s2s_sustained(lambda r: fwd[r](), between=lambda r: inv[r]()))
show("(a) inverse, bursts of 100", s2s_bursts(lambda r: inv[r]()))
show("(b) inverse, sustained 30 x 200", s2s_sustained(lambda r: inv[r]()))
show("(a) stream probe, bursts of 100", s2s_bursts(probe))
show("(b) stream probe, sustained 30 x 200", s2s_sustained(probe))
show("(c) stream probe + inverse", s2s_sustained(probe, between=lambda r: inv[r]()))
