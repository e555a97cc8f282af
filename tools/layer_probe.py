"""Every single-layer kernel of the path at the benchmark shape on 4 rotating input sets; run under
`rocprofv3 --kernel-trace --stats` for kernel-only times (tools/refresh_profiles.sh does)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from categoricalnf_amd import ops
dev = torch.device("cuda:0")
B, N, D, R, REP = 16384, 64, 6, 4, 40
g = torch.Generator(device=dev).manual_seed(0)
zs = [torch.randn(B, N, D, generator=g, device=dev) for _ in range(R)]
ext = [0.3 * torch.randn(B, N, 2 * D, generator=g, device=dev) for _ in range(R)]
us = [torch.rand(B, N, D, generator=g, device=dev) for _ in range(R)]
bias, scales = torch.randn(1, 1, D, device=dev), 0.1 * torch.randn(1, 1, D, device=dev)
w = torch.linalg.qr(torch.randn(D, D, device=dev))[0].contiguous()
sldj = torch.slogdet(w)[1]
ln = torch.full((B,), float(N), device=dev)
ldj = torch.zeros(B, device=dev)
for i in range(REP):
    z = zs[i % R]
    ops.actnorm(z, bias, scales)
    ops.invconv(z, w, sldj)
    ops.actnorm_invconv(z, bias, scales, w, sldj)
    ops.ext_actnorm(z, ext[i % R])
    ops.prior_nll(z, ldj, ln)
    ops.logistic_log_prob(z)
    ops.logistic_from_uniform(us[i % R])
    ops.sigmoid_flow(z)
torch.cuda.synchronize()
print("done")
