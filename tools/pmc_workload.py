"""Workload for the rocprofv3 --pmc passes: N launches each of the affine coupling fwd/inv kernels at the
benchmark shape (rotating buffer sets), the mixture fwd kernel at configs[1], and a torch float4 d2d
copy of KNOWN size used to calibrate FETCH_SIZE / WRITE_SIZE (MI355X_MICROARCH.md §HBM)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from categoricalnf_amd import ops
dev = torch.device("cuda:0")
B, N, D, R, REP = 16384, 64, 6, 4, 12
g = torch.Generator(device=dev).manual_seed(0)
zs = [torch.randn(B, N, D, generator=g, device=dev) for _ in range(R)]
nns = [0.5 * torch.randn(B, N, 2 * D, generator=g, device=dev) for _ in range(R)]
sf = torch.zeros(D, device=dev)
mask = torch.tensor([[1., 1., 1., 0., 0., 0.]], device=dev)
zo = [torch.empty_like(zs[0]) for _ in range(R)]
lo = [torch.empty(B, device=dev) for _ in range(R)]
ln = torch.full((B,), float(N), device=dev)
neglog, nll = torch.empty(B, device=dev), torch.empty(B, device=dev)
# the bench's dominant kernel: forward coupling with the NLL epilogue
fwd = [ops.affine_coupling_nll_launch(zs[r], nns[r], sf, mask, zo[r], lo[r], ln, neglog, nll, None) for r in range(R)]
plain = [ops.affine_coupling_launch(zs[r], nns[r], sf, mask, zo[r], lo[r]) for r in range(R)]
inv = [ops.affine_coupling_launch(zs[r], nns[r], sf, mask, zo[r], lo[r], reverse=True) for r in range(R)]
for i in range(REP):
    fwd[i % R]()
for i in range(REP):
    plain[i % R]()
for i in range(REP):
    inv[i % R]()
# calibration copy: 25,165,824 floats = 100.66 MB read + 100.66 MB written per launch
srcs = [torch.randn(B * N * D * 4, generator=g, device=dev) for _ in range(R)]
dsts = [torch.empty_like(srcs[0]) for _ in range(R)]
for i in range(REP):
    dsts[i % R].copy_(srcs[i % R])
# the fp32 mixture forward: configs[1] and S*, reference layout then compact layout — FOUR groups of REP launches of the same
# kernel, in this order (tools/pmc_summarize.py tells them apart by dispatch order)
K = 8
del zs, nns, zo, srcs, dsts
for (Bm, Nm, Dm) in ((16384, 16, 4), (16384, 64, 6)):
    DAm = Dm - Dm // 2
    mm = torch.tensor([[1.] * (Dm // 2) + [0.] * DAm], device=dev)
    zm = [torch.randn(Bm, Nm, Dm, generator=g, device=dev) for _ in range(R)]
    zmo, lmo = torch.empty_like(zm[0]), torch.empty(Bm, device=dev)
    for width in (Dm * (2 + 3 * K), DAm * (2 + 3 * K)):
        nm = [0.5 * torch.randn(Bm, Nm, width, generator=g, device=dev) for _ in range(R)]
        mf = [ops.mixture_coupling_launch(zm[r], nm[r], mm, K, zmo, lmo) for r in range(R)]
        for i in range(REP):
            mf[i % R]()
        torch.cuda.synchronize()
        del nm, mf
torch.cuda.synchronize()
print("done")
