"""fp32 mixture backward at the benchmark shapes, reference and compact parameter layout, by lanes per item, for A/B builds of the
library (tools/build_variant.sh; -DCNF_MIXBWD_ABLATE[=2|3]: arithmetic / + g_nn write-back / + g_z stores compiled out):
    CNF_LIB_OVERRIDE=categoricalnf_amd/lib/var_<name>.so python tools/mixture_bwd_floor.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from categoricalnf_amd import _lib, ops
from categoricalnf_amd.layers.flows.coupling_layer import CouplingLayer
dev = torch.device("cuda:0")
lib = _lib.load()
P_ = ops._ptr
modes = [int(m) for m in sys.argv[1:]] or [-1, 2, 3, 4, 0]
shapes = [("cfg1", 16384, 16, 4, 8), ("S*", 16384, 64, 6, 8)]
if os.environ.get("SHAPES"):           # SHAPES="name:B:N:D:K,..."
    shapes = [(f[0],) + tuple(int(v) for v in f[1:]) for f in (t.split(":") for t in os.environ["SHAPES"].split(","))]
for name, B, N, D, K in shapes:
    g = torch.Generator(device=dev).manual_seed(1)
    R = 2
    DA, P = D - D // 2, 2 + 3 * K
    zs = [torch.randn(B, N, D, generator=g, device=dev) for _ in range(R)]
    gzs = [torch.randn(B, N, D, generator=g, device=dev) for _ in range(R)]
    gl = torch.randn(B, generator=g, device=dev)
    mask = CouplingLayer.create_channel_mask(D).to(dev)
    m, mr, mc = ops._mask_desc(mask, D, dev)
    act, n_act = ops._act_list(mask, m, mr, mc, D)
    sf, msf = torch.zeros(D, device=dev), torch.zeros(D, K, device=dev)
    g_sf, g_msf = torch.empty_like(sf), torch.empty_like(msf)
    g_z = torch.empty_like(zs[0])
    ws = torch.empty(int(lib.cnf_bwd_workspace_floats(D + D * K)), device=dev)
    layouts = (("ref", D * P, "cnf_mixture_coupling_bwd_f32"), ("compact", DA * P, "cnf_mixture_coupling_compact_bwd_f32"))
    if os.environ.get("LAYOUT"):            # LAYOUT=ref|compact: one layout only (rocprofv3 runs: the kernel name is the same for both)
        layouts = tuple(l for l in layouts if l[0] == os.environ["LAYOUT"])
    for layout, width, entry in layouts:
        nns = [0.5 * torch.randn(B, N, width, generator=g, device=dev) for _ in range(R)]
        g_nn = torch.empty_like(nns[0])
        fn = getattr(lib, entry)

        def call(r):
            rc = fn(P_(zs[r]), P_(nns[r]), P_(sf), P_(msf), P_(m), mr, mc, act, n_act, None, 0, 0, P_(gzs[r]), P_(gl), P_(g_z), P_(g_nn),
                    P_(g_sf), P_(g_msf), P_(ws), B, N, D, K, -1.0, 1.0, 1, ops._stream(dev))
            assert rc == 0, lib.cnf_last_error()

        def timeit(reps=8):
            for r in range(R):
                call(r)
            torch.cuda.synchronize()
            ts = []
            for _ in range(5):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for i in range(reps):
                    call(i % R)
                b.record(); torch.cuda.synchronize()
                ts.append(a.elapsed_time(b) / reps * 1e3)
            return sorted(ts)[len(ts) // 2]
        moved = B * N * (2 * DA * P * 4 + 12 * D) + 4 * B
        out = []
        for mode in modes:
            lib.cnf_set_mixture_bwd_waves(mode)
            t = timeit()
            out.append("mode %2d: %6.1f us (%.2f TB/s)" % (mode, t, moved / t / 1e6))
        lib.cnf_set_mixture_bwd_waves(-1)
        print(os.environ.get("CNF_LIB_OVERRIDE", "default"), "%s/%s (%.0f MB to move)" % (name, layout, moved / 1e6), " | ".join(out), flush=True)
        del nns, g_nn
