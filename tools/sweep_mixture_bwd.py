"""Timing of the mixture-coupling backward: fp32 token-pass kernel (cnf_mixture_coupling_bwd_f32) vs the fp64 kernel
(cnf_mixture_coupling_bwd, which also needs a memset of g_nn), config-shaped workloads.  GPU only."""
import ctypes, os, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from categoricalnf_amd import _lib, ops
from categoricalnf_amd.functional import _ws
from categoricalnf_amd.layers.flows.coupling_layer import CouplingLayer
dev = torch.device("cuda:0")
lib = _lib.load()
P_ = ops._ptr
shapes = [("set_summation configs[1]", 16384, 16, 4, 8, True), ("north-star S* mixture", 16384, 64, 6, 8, True),
          ("graph colouring large", 128, 50, 6, 16, True), ("PTB AR (mask=None)", 128, 288, 3, 51, False),
          ("zinc nodes", 512, 38, 6, 16, True), ("zinc edges", 512, 703, 2, 8, True), ("training batch 1024 sets", 1024, 16, 4, 8, True)]
for name, B, N, D, K, masked in shapes:
    g = torch.Generator(device=dev).manual_seed(1)
    z = torch.randn(B, N, D, generator=g, device=dev)
    nn_out = 0.5 * torch.randn(B, N, D * (2 + 3 * K), generator=g, device=dev)
    sf, msf = torch.zeros(D, device=dev), torch.zeros(D, K, device=dev)
    mask = CouplingLayer.create_channel_mask(D).to(dev) if masked else None
    m, mr, mc = ops._mask_desc(mask, D, dev)
    act, n_act = ops._act_list(mask, m, mr, mc, D)
    gz, gl = torch.randn(B, N, D, generator=g, device=dev), torch.randn(B, generator=g, device=dev)
    g_z, g_nn = torch.empty_like(z), torch.empty_like(nn_out)
    g_sf, g_msf = torch.empty_like(sf), torch.empty_like(msf)
    ws = _ws(D + D * K, dev)
    st = lambda: ops._stream(dev)

    def f32():
        lib.cnf_mixture_coupling_bwd_f32(P_(z), P_(nn_out), P_(sf), P_(msf), P_(m), mr, mc, act, n_act, None, 0, 0, P_(gz), P_(gl),
                                         P_(g_z), P_(g_nn), P_(g_sf), P_(g_msf), P_(ws), B, N, D, K, -1.0, 1.0, 1, st())

    def f64():
        lib.cnf_mixture_coupling_bwd(P_(z), P_(nn_out), P_(sf), P_(msf), P_(m), mr, mc, None, 0, 0, P_(gz), P_(gl),
                                     P_(g_z), P_(g_nn), P_(g_sf), P_(g_msf), P_(ws), B, N, D, K, -1.0, 1.0, 1, st())

    def timeit(fn, reps=10):
        """(start-to-start us per call incl. the host's ctypes marshalling, us of the call's kernels from dispatch-bound events)"""
        fn(); torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(reps):
                fn()
            b.record(); torch.cuda.synchronize()
            ts.append(a.elapsed_time(b) / reps * 1e3)
        lib.cnf_prof_arm(3 * 5)
        for _ in range(5):
            fn()
        buf = (ctypes.c_float * 16)()
        n = lib.cnf_prof_collect(buf, 16)
        kern = sum(buf[i] for i in range(n)) / 5 * 1e3
        return min(ts), kern
    per_mode = []
    for mode in (0, 1):             # natural register allocation / held to 4 waves per SIMD
        lib.cnf_set_mixture_bwd_waves(mode)
        per_mode.append(timeit(f32))
    if True:
        # A/B: the rolled run-time-K kernel with 1 / 2 / 4 lanes per item, natural and 4-wave builds (the default is the rolled kernel
        # with G by the amount of work; modes 0 / 1 are the unrolled register-slot kernels)
        extra = []
        for mode in range(2, 8):
            lib.cnf_set_mixture_bwd_waves(mode)
            try:
                extra.append("%.1f" % timeit(f32)[0])
            except Exception as e:
                extra.append("n/a")
        print("   rolled kernel, lanes per item 1 / 2 / 4: natural %s / %s / %s us, 4 waves per SIMD %s / %s / %s us" % tuple(extra), flush=True)
    lib.cnf_set_mixture_bwd_waves(-1)
    (t32, k32), (t64, k64) = timeit(f32), timeit(f64, reps=3)
    e = B * N * D
    byts = e * (16 + 24 * K) + 8 * e          # read z, g_zout, nn_out; write g_z, g_nn (all blocks)
    print("%-26s B=%5d N=%3d D=%d K=%2d | fp32 token-pass %8.1f us per call, kernels %7.1f us (%.0f GB/s incl. the g_nn zeros) | fp64 kernel %8.1f us per call "
          "(with its memset), kernels %7.1f us | %.1fx per call, %.1fx kernels | unrolled kernels: natural regs %.1f / %.1f us, 4 waves per SIMD %.1f / %.1f us"
          % (name, B, N, D, K, t32, k32, byts / k32 / 1e3, t64, k64, t64 / t32, k64 / k32, per_mode[0][0], per_mode[0][1], per_mode[1][0], per_mode[1][1]), flush=True)
