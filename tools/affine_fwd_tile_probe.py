"""Experiment: the affine coupling forward in token-owner wave-tile form (cnf_probe_affine_fwd_tile, csrc/cnf_probe.hip) against the
shipped flat row-tile kernel at S* = (16384, 64, 6): same results, start-to-start microseconds on rotating buffers."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from categoricalnf_amd import _lib, ops
from categoricalnf_amd.ops import _ptr, _stream
dev = torch.device("cuda:0"); lib = _lib.load()
B, N, D, R = 16384, 64, 6, 4
g = torch.Generator(device=dev).manual_seed(0)
zs = [torch.randn(B, N, D, generator=g, device=dev) for _ in range(R)]
nns = [0.5 * torch.randn(B, N, 2 * D, generator=g, device=dev) for _ in range(R)]
sf = 0.1 * torch.randn(D, generator=g, device=dev)
mask = torch.tensor([[1., 1., 1., 0., 0., 0.]], device=dev)
ldj0 = torch.randn(B, generator=g, device=dev)
zo, lo = torch.empty(B, N, D, device=dev), torch.empty(B, device=dev)
fn = lib.cnf_probe_affine_fwd_tile
ref_z, ref_l = ops.affine_coupling(zs[0], nns[0], sf, mask, ldj=ldj0)
assert fn(_ptr(zs[0]), _ptr(nns[0]), _ptr(sf), _ptr(mask), _ptr(ldj0), _ptr(zo), _ptr(lo), B, N, 2, 1, _stream(dev)) == 0
torch.cuda.synchronize()
print("z: equal bits %s, max |dev| %.2e; ldj max |dev| %.2e (scale %.1f)" % (torch.equal(zo, ref_z), float((zo - ref_z).abs().max()), float((lo - ref_l).abs().max()), float(ref_l.abs().max())))


def timeit(f, reps=200, blocks=5):
    for i in range(8):
        f(i % R)
    torch.cuda.synchronize()
    ts = []
    for _ in range(blocks + 1):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for i in range(reps):
            f(i % R)
        b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / reps * 1e3)
    ts = sorted(ts[1:])
    return ts[len(ts) // 2]


launches = [ops.affine_coupling_launch(zs[r], nns[r], sf, mask, zo, lo, ldj=ldj0) for r in range(R)]
print("shipped flat row-tile kernel:            %.2f us" % timeit(lambda i: launches[i]()))
for G in (1, 2, 4, 8):
    for nt in (1, 0):
        t = timeit(lambda i: fn(_ptr(zs[i]), _ptr(nns[i]), _ptr(sf), _ptr(mask), _ptr(ldj0), _ptr(zo), _ptr(lo), B, N, G, nt, _stream(dev)))
        print("token-owner tile form, %d tiles per wave, nn loads %s: %.2f us" % (G, "nontemporal" if nt else "plain", t))
print("shipped flat row-tile kernel (again):    %.2f us" % timeit(lambda i: launches[i]()))
