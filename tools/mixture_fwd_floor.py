"""Forward / inverse of the fp32 mixture coupling at the benchmark shapes, reference and compact parameter layout, for A/B builds of the
library (tools/build_variant.sh; -DCNF_MIXFWD_ABLATE compiles the arithmetic out: the data-movement floor of the launch geometry):
    CNF_LIB_OVERRIDE=categoricalnf_amd/lib/var_<name>.so python tools/mixture_fwd_floor.py [tile_items ...]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from categoricalnf_amd import _lib, ops
from categoricalnf_amd.layers.flows.coupling_layer import CouplingLayer
dev = torch.device("cuda:0")
lib = _lib.load()
tiles = [int(t) for t in sys.argv[1:]] or [0]
lib.cnf_set_math_mode(int(os.environ.get("MATH_MODE", "1")))
for tile in tiles:
    if tile:
        lib.cnf_set_mixture_tile(tile)
    out = []
    for name, B, N, D, K in [("cfg1", 16384, 16, 4, 8), ("S*", 16384, 64, 6, 8)]:
        g = torch.Generator(device=dev).manual_seed(1)
        R = 3
        DA, P = D - D // 2, 2 + 3 * K
        zs = [torch.randn(B, N, D, generator=g, device=dev) for _ in range(R)]
        mask = CouplingLayer.create_channel_mask(D).to(dev)
        zf, zr = torch.empty_like(zs[0]), torch.empty_like(zs[0])
        lf, lr = torch.empty(B, device=dev), torch.empty(B, device=dev)
        for layout, width in (("ref", D * P), ("compact", DA * P)):
            nns = [0.5 * torch.randn(B, N, width, generator=g, device=dev) for _ in range(R)]
            fwd = [ops.mixture_coupling_launch(zs[r], nns[r], mask, K, zf, lf) for r in range(R)]
            inv = [ops.mixture_coupling_launch(zf, nns[r], mask, K, zr, lr, reverse=True) for r in range(R)]

            def timeit(ls, reps=12):
                for l in ls:
                    l()
                torch.cuda.synchronize()
                ts = []
                for _ in range(7):
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record()
                    for i in range(reps):
                        ls[i % R]()
                    b.record(); torch.cuda.synchronize()
                    ts.append(a.elapsed_time(b) / reps * 1e3)
                return sorted(ts)[len(ts) // 2]
            needed = B * N * (DA * P * 4 + 8 * D) + 4 * B
            tf, ti = timeit(fwd), timeit(inv)
            out.append("%s/%s %.1f / %.1f us (%.2f / %.2f TB/s on %.0f MB)" % (name, layout, tf, ti, needed / tf / 1e6, needed / ti / 1e6, needed / 1e6))
            del nns, fwd, inv
    print(os.environ.get("CNF_LIB_OVERRIDE", "default"), "tile", tile or "default", "| fwd / inv:", "   ".join(out), flush=True)
    ops.flag_word(dev).zero_()
