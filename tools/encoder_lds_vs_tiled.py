"""LDS-resident vs class-tiled encoder forward over vocabulary sizes and launch shapes: the data behind
ops.encoder_prefers_tiled_forward (profiles/r03_encoder_lds_vs_tiled.txt)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from categoricalnf_amd import _lib, ops
lib = _lib.load()
dev = torch.device("cuda:0")
def steady(fn, reps=30, blocks=5):
    m = [torch.cuda.Event(enable_timing=True) for _ in range(blocks + 1)]
    fn(); torch.cuda.synchronize(); m[0].record()
    for b in range(blocks):
        for i in range(reps): fn()
        m[b + 1].record()
    torch.cuda.synchronize()
    return float(np.median([m[b].elapsed_time(m[b + 1]) / reps * 1e3 for b in range(1, blocks)]))
SHAPES = (((2048, 64, 6), (512, 64, 6), (16384, 16, 4)), (160, 200, 256, 300, 350, 500)), \
         (((128, 288, 3), (128, 256, 3), (384, 64, 6), (1024, 64, 6), (64, 703, 2)), (3, 9, 27, 51, 100, 160, 500))
for (B, N, D), C in ((s, c) for shapes, cs in SHAPES for s in shapes for c in cs):
    if True:
        if not ops.encoder_fused_supported(C, D): continue
        g = torch.Generator(device=dev).manual_seed(0)
        categ = torch.randint(0, C, (B, N), generator=g, device=dev)
        table = 0.5 * torch.randn(C, 2 * D, generator=g, device=dev)
        prior = torch.log_softmax(torch.randn(C, generator=g, device=dev), 0)
        eps = ops.logistic_from_uniform(torch.rand(B * N, D, generator=g, device=dev))
        f = min(steady(lambda: ops.encoder_forward(categ, eps, table, prior, tiled=False)) for _ in range(2))
        t = min(steady(lambda: ops.encoder_forward(categ, eps, table, prior, tiled=True)) for _ in range(2))
        print("B=%5d N=%3d D=%d C=%4d table %6d B | LDS-resident %7.2f us | tiled %7.2f us" % (B, N, D, C, C * (6 * D + 3) * 4, f, t), flush=True)
