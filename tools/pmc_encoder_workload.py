"""Workload for rocprofv3 counter passes on the categorical-encoder kernels at the benchmark shape
(B=16384, N=64, D=6; C from argv, default 16): REP forward and REP decode launches."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from categoricalnf_amd import ops
dev = torch.device("cuda:0")
B, N, D = 16384, 64, 6
C = int(sys.argv[1]) if len(sys.argv) > 1 else 16
REP = 10
g = torch.Generator(device=dev).manual_seed(0)
categ = torch.randint(0, C, (B, N), generator=g, device=dev)
table = 0.5 * torch.randn(C, 2 * D, generator=g, device=dev)
prior = torch.log_softmax(torch.randn(C, generator=g, device=dev), 0)
eps = ops.logistic_from_uniform(torch.rand(B * N, D, generator=g, device=dev))
z, ldj, _ = ops.encoder_forward(categ, eps, table, prior)
torch.cuda.synchronize()
for _ in range(REP):
    ops.encoder_forward(categ, eps, table, prior)
torch.cuda.synchronize()
for _ in range(REP):
    ops.encoder_decode(z, table, prior)
torch.cuda.synchronize()
