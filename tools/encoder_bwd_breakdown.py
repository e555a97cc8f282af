"""rocprofv3 --kernel-trace --stats target: the encoder's forward + backward at the benchmark shape (10^6 tokens, 16 classes)
and at a word-level vocabulary (36 864 tokens, 10^4 classes), 20 steps each."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from categoricalnf_amd import ops, functional as Fn
dev = torch.device("cuda:0")
D = 6
SHAPES = ((16384, 64, 16), (128, 288, 10000))
if len(sys.argv) > 1:          # e.g. 16384,64,16  16384,64,51
    SHAPES = tuple(tuple(int(v) for v in a.split(",")) for a in sys.argv[1:])
for B, N, C in SHAPES:
    g = torch.Generator(device=dev).manual_seed(1)
    categ = torch.randint(0, C, (B, N), generator=g, device=dev)
    table = (0.5 * torch.randn(C, 2 * D, generator=g, device=dev)).requires_grad_()
    prior = torch.log_softmax(torch.randn(C, generator=g, device=dev), 0)
    eps = ops.logistic_from_uniform(torch.rand(B * N, D, generator=g, device=dev))
    gz, gl = torch.randn(B, N, D, device=dev), torch.randn(B, device=dev)
    for _ in range(20):
        z, ldj, _ = Fn.EncoderForwardFn.apply(table, categ, eps, prior, None, 1.0, False, None)
        torch.autograd.backward([z, ldj], [gz, gl])
        table.grad = None
    torch.cuda.synchronize()
