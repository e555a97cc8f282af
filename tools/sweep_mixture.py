"""Interleaved timing of the mixture-CDF coupling kernels (fwd / inverse, both inverse modes) on the
config-shaped workloads of SURVEY.md §8d.  GPU only."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from categoricalnf_amd import _lib, ops
dev = torch.device("cuda:0")
lib = _lib.load()
shapes = [("set_summation configs[1]", 16384, 16, 4, 8, "channel"), ("north-star S* mixture", 16384, 64, 6, 8, "channel"),
          ("graph colouring large", 128, 50, 6, 16, "channel"), ("PTB AR (mask=None)", 128, 288, 3, 51, "none"),
          ("zinc nodes", 512, 38, 6, 16, "channel"), ("zinc edges", 512, 703, 2, 8, "channel"),
          ("graph colouring tiny", 384, 20, 2, 8, "channel"), ("text8 AR (mask=None)", 128, 256, 3, 27, "none"),
          ("default K = 10", 1024, 64, 4, 10, "channel")]
for name, B, N, D, K, kind in shapes:
    g = torch.Generator(device=dev).manual_seed(1)
    z = torch.randn(B, N, D, generator=g, device=dev)
    nn_out = 0.5 * torch.randn(B, N, D * (2 + 3 * K), generator=g, device=dev)
    mask = None if kind == "none" else torch.cat([torch.ones(1, D // 2), torch.zeros(1, D - D // 2)], 1).to(dev)
    zf, zr = torch.empty_like(z), torch.empty_like(z)
    lf, lr = torch.empty(B, device=dev), torch.empty(B, device=dev)
    fwd = ops.mixture_coupling_launch(z, nn_out, mask, K, zf, lf)
    inv = ops.mixture_coupling_launch(zf, nn_out, mask, K, zr, lr, reverse=True)

    def timeit(fn, reps=10):
        fn(); torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(reps):
                fn()
            b.record(); torch.cuda.synchronize()
            ts.append(a.elapsed_time(b) / reps * 1e3)
        return min(ts)
    tf = timeit(fwd)
    lib.cnf_set_mixture_kernel(1)
    tf_r1 = timeit(fwd)
    ti_r1 = timeit(inv, reps=5)
    lib.cnf_set_mixture_kernel(0)
    splits = []
    if B < 2048:
        for w in (1024, 2048, 4096, 8192, 16384):
            lib.cnf_set_mixture_split(w)
            splits.append("%d:%.1f/%.1f" % (w, timeit(fwd), timeit(inv, reps=5)))
        lib.cnf_set_mixture_split(4096)
    whole = ""
    if kind == "channel" and (D // 2 * (2 + 3 * K) * 4 + 128) * 20 >= D * (2 + 3 * K) * 4 * 19:
        lib.cnf_set_mixture_whole_tokens(0)
        whole = "; spans only (whole-token staging off) %.1f/%.1f" % (timeit(fwd), timeit(inv, reps=5))
        lib.cnf_set_mixture_whole_tokens(1)
    lanes = []
    if K not in (4, 8, 16):
        for l in (1, 2, 4):
            lib.cnf_set_mixture_lanes(l)
            lanes.append("%d:%.1f/%.1f" % (l, timeit(fwd), timeit(inv, reps=5)))
        lib.cnf_set_mixture_lanes(0)
    lib.cnf_set_math_mode(0)
    tf64 = timeit(fwd)
    lib.cnf_set_math_mode(1)
    tiles = []
    for tile in (64, 128, 256):
        lib.cnf_set_mixture_tile(tile)
        tiles.append("%d:%.1f" % (tile, timeit(fwd)))
    lib.cnf_set_mixture_tile(128)
    elems = B * N * D
    line = "%-26s B=%5d N=%3d D=%d K=%2d | fwd %8.1f us (%6.2f Gelem/s, %5.0f GB/s alg = %.2f of 8 TB/s; round-1 kernel %.1f us fwd / %.1f us inv; fp64 kernel %7.1f us; tile %s; split-waves fwd/inv %s; rolled loop with lanes fwd/inv %s%s)" % (
        name, B, N, D, K, tf, elems / tf / 1e3, elems * (16 + 12 * K) / tf / 1e3, elems * (16 + 12 * K) / tf / 1e3 / 8000.0,
        tf_r1, ti_r1, tf64, " ".join(tiles), " ".join(splits), " ".join(lanes), whole)
    for tag, math, mode in (("fp64 bisect", 0, 0), ("fp64 newton", 0, 1), ("fp32 newton", 1, 1)):
        lib.cnf_set_math_mode(math); lib.cnf_set_inverse_mode(mode)
        ti = timeit(inv, reps=5)
        err = (zr - z).abs().max().item()
        line += " | inv[%s] %7.1f us err %.1e" % (tag, ti, err)
    lib.cnf_set_math_mode(1); lib.cnf_set_inverse_mode(1)
    print(line, flush=True)
