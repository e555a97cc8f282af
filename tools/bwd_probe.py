"""Every streaming backward kernel of csrc/cnf_backward.hip at the north-star shape (B=16384, N=64, D=6) through the C ABI
on rotating buffer sets: start-to-start microseconds over blocks of back-to-back launches, algorithmic bytes (the tensors a
kernel must read and write once), fraction of the 8 TB/s HBM peak; a sweep of the flat-tile knobs (cnf_set_bwd_tile) for
the affine kernel; and a run-to-run bit-equality check of the parameter gradients.  GPU only.

  python tools/bwd_probe.py [--sweep] [--B 16384 --N 64 --D 6]
"""
import argparse, ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from categoricalnf_amd import _lib, ops
from categoricalnf_amd.ops import _ptr, _stream

ap = argparse.ArgumentParser()
ap.add_argument("--B", type=int, default=16384); ap.add_argument("--N", type=int, default=64); ap.add_argument("--D", type=int, default=6)
ap.add_argument("--sweep", action="store_true"); ap.add_argument("--reps", type=int, default=100)
ap.add_argument("--only", default=""); ap.add_argument("--U", type=int, default=0); ap.add_argument("--G", type=int, default=0)
ap.add_argument("--pmc", action="store_true", help="counter workload: 12 launches of each kernel and the manifest $CNF_MANIFEST for tools/ceilings.py")
args = ap.parse_args()
dev = torch.device("cuda:0")
lib = _lib.load()
B, N, D, R = args.B, args.N, args.D, 4
lib.cnf_set_bwd_tile(args.U, args.G)
elems = B * N * D
g = torch.Generator(device=dev).manual_seed(0)
rn = lambda *s, k=1.0: [k * torch.randn(*s, generator=g, device=dev) for _ in range(R)]
zo, nn2, gzo = rn(B, N, D), rn(B, N, 2 * D, k=0.5), rn(B, N, D)
s_, t_ = rn(B, N, D, k=0.3), rn(B, N, D)
gl = torch.randn(B, generator=g, device=dev)
gz = [torch.empty(B, N, D, device=dev) for _ in range(R)]
gz2 = [torch.empty(B, N, D, device=dev) for _ in range(R)]
gz3 = [torch.empty(B, N, D, device=dev) for _ in range(R)]
gnn = [torch.empty(B, N, 2 * D, device=dev) for _ in range(R)]
mask = torch.cat([torch.ones(1, D // 2), torch.zeros(1, D - D // 2)], 1).to(dev)
sf = 0.1 * torch.randn(D, generator=g, device=dev)
bias, scales = torch.randn(D, generator=g, device=dev), 0.1 * torch.randn(D, generator=g, device=dev)
w = torch.linalg.qr(torch.randn(D, D))[0].to(dev).contiguous()
ln = torch.full((B,), float(N), device=dev)
pad = (torch.rand(B, N, generator=g, device=dev) > 0.1).float()
ws = torch.empty(int(lib.cnf_bwd_workspace_floats(D * D + 2 + 2 * D)), device=dev)
g_sf, g_b, g_s, g_w, g_sl = (torch.empty(n, device=dev) for n in (D, D, D, D * D, 1))
gldj = torch.empty(B, device=dev)
st = lambda: _stream(dev)


def call(name, *a):
    rc = getattr(lib, name)(*a)
    if rc != 0:
        raise RuntimeError("%s -> %d: %s" % (name, rc, lib.cnf_last_error().decode()))


def affine(i, rev=0, with_sf=True):
    call("cnf_affine_coupling_bwd", _ptr(zo[i]), _ptr(nn2[i]), _ptr(sf) if with_sf else None, _ptr(mask), 1, D, _ptr(gzo[i]), _ptr(gl),
         _ptr(gz[i]), _ptr(gnn[i]), _ptr(g_sf) if with_sf else None, _ptr(ws), B, N, D, rev, st())


def actnorm(i, rev=0, padded=False):
    call("cnf_actnorm_bwd", _ptr(zo[i]), _ptr(bias), _ptr(scales), _ptr(pad) if padded else None, _ptr(ln), _ptr(gzo[i]), _ptr(gl),
         _ptr(gz[i]), _ptr(g_b), _ptr(g_s), _ptr(ws), B, N, D, rev, st())


def invconv(i, rev=0, padded=False):
    call("cnf_invconv_bwd", _ptr(zo[i]), _ptr(w), _ptr(pad) if padded else None, _ptr(ln), _ptr(gzo[i]), _ptr(gl),
         _ptr(gz[i]), _ptr(g_w), _ptr(g_sl), _ptr(ws), B, N, D, rev, st())


g_par = torch.empty(D * D + 1 + 2 * D, device=dev)
sldj_t = torch.slogdet(w)[1].reshape(1).contiguous()
outs = None


def actconv(i, from_out=0):
    global outs
    if from_out and outs is None:
        outs = [ops.actnorm_invconv(zo[r], bias, scales, w, sldj_t)[0] for r in range(R)]
    call("cnf_actnorm_invconv_bwd", _ptr(outs[i] if from_out else zo[i]), from_out, _ptr(bias), _ptr(scales), _ptr(w), None, None, _ptr(ln),
         _ptr(gzo[i]), _ptr(gl), _ptr(gz[i]), _ptr(g_par), _ptr(ws), B, N, D, st())


def ext(i, rev=0):
    call("cnf_ext_actnorm_bwd", _ptr(zo[i]), _ptr(nn2[i]), None, _ptr(gzo[i]), _ptr(gl), _ptr(gz[i]), _ptr(gnn[i]), B, N, D, rev, st())


def nll(i):
    call("cnf_prior_nll_bwd", _ptr(zo[i]), None, _ptr(ln), _ptr(gl), _ptr(gz[i]), _ptr(gldj), B, N, D, ctypes.c_float(ops.LOGISTIC_SIGMA), st())


def logp(i):
    call("cnf_logistic_log_prob_bwd", _ptr(zo[i]), _ptr(gzo[i]), _ptr(gz[i]), elems, ctypes.c_float(0.0), ctypes.c_float(ops.LOGISTIC_SIGMA), st())


def sigmoid(i, rev=0):
    call("cnf_sigmoid_flow_bwd", _ptr(zo[i]), _ptr(gzo[i]), _ptr(gl), _ptr(gz[i]), B, N * D, rev, ctypes.c_float(1e-5), st())


def aff_params(i):
    call("cnf_affine_params_bwd", _ptr(nn2[i]), _ptr(sf), _ptr(mask), 1, D, _ptr(s_[i]), _ptr(t_[i]), _ptr(gnn[i]), _ptr(g_sf), _ptr(ws), B, N, D, st())


def aff_transform(i, rev=0):
    call("cnf_affine_transform_bwd", _ptr(zo[i]), _ptr(s_[i]), _ptr(t_[i]), _ptr(gzo[i]), _ptr(gl), _ptr(gz[i]), _ptr(gz2[i]), _ptr(gz3[i]), B, N, D, rev, st())


def forced(mode, fn, *a):
    lib.cnf_set_affine_bwd_tiles(mode)
    fn(*a)
    lib.cnf_set_affine_bwd_tiles(1)


def timeit(fn, reps=args.reps, blocks=5):
    for i in range(2 * R):
        fn(i % R)
    torch.cuda.synchronize()
    ts = []
    for _ in range(blocks + 1):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for i in range(reps):
            fn(i % R)
        b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / reps * 1e3)
    ts = sorted(ts[1:])            # the first block ramps the clocks
    return ts[len(ts) // 2], ts[0]


rows = [
    ("affine_bwd fwd-dir (sf), token-owner kernel (default)", 28, lambda i: forced(2, affine, i, 0)),
    ("affine_bwd fwd-dir (sf), flat-tile kernel", 28, lambda i: forced(0, affine, i, 0)),
    ("affine_bwd inv-dir (sf), flat-tile kernel (default)", 28, lambda i: forced(0, affine, i, 1)),
    ("affine_bwd inv-dir (sf), token-owner kernel", 28, lambda i: forced(2, affine, i, 1)),
    ("affine_bwd fwd-dir (no sf), token-owner kernel (default)", 28, lambda i: forced(2, affine, i, 0, False)),
    ("affine_bwd fwd-dir (no sf), flat-tile kernel", 28, lambda i: forced(0, affine, i, 0, False)),
    ("actnorm_bwd", 12, lambda i: actnorm(i, 0)),
    ("actnorm_bwd padded", 12, lambda i: actnorm(i, 0, True)),
    ("actnorm_bwd (flat-tile kernel)", 12, lambda i: (lib.cnf_set_actnorm_bwd_tiles(0), actnorm(i, 0), lib.cnf_set_actnorm_bwd_tiles(1))),
    ("invconv_bwd", 12, lambda i: invconv(i, 0)),
    ("invconv_bwd padded", 12, lambda i: invconv(i, 0, True)),
    ("actnorm+invconv_bwd fused (from input)", 12, lambda i: actconv(i, 0)),
    ("actnorm+invconv_bwd fused (from output)", 12, lambda i: actconv(i, 1)),
    ("ext_actnorm_bwd", 28, lambda i: ext(i, 0)),
    ("prior_nll_bwd", 8, nll),
    ("logistic_log_prob_bwd", 12, logp),
    ("sigmoid_flow_bwd", 12, lambda i: sigmoid(i, 0)),
    ("affine_params_bwd", 24, aff_params),
    ("affine_transform_bwd", 28, lambda i: aff_transform(i, 0)),
]
if args.only:
    rows = [r for r in rows if any(k in r[0] for k in args.only.split(","))]
if args.pmc:
    import json
    frag = {"affine_bwd fwd-dir (sf), token-owner kernel (default)": "affine_bwd_tile_kernel<6, true, false", "affine_bwd fwd-dir (sf), flat-tile kernel": "affine_bwd_kernel<4, 2, true, false",
            "affine_bwd inv-dir (sf), flat-tile kernel (default)": "affine_bwd_kernel<4, 2, true, true", "affine_bwd inv-dir (sf), token-owner kernel": "affine_bwd_tile_kernel<6, true, true",
            "affine_bwd fwd-dir (no sf), token-owner kernel (default)": "affine_bwd_tile_kernel<6, false, false", "affine_bwd fwd-dir (no sf), flat-tile kernel": "affine_bwd_kernel<4, 2, false, false",
            "actnorm_bwd": "::actnorm_bwd_tile_kernel<", "actnorm_bwd (flat-tile kernel)": "::actnorm_bwd_kernel<",
            "invconv_bwd": "::invconv_bwd_kernel<6>", "actnorm+invconv_bwd fused (from input)": "actconv_bwd_kernel<6, false>",
            "actnorm+invconv_bwd fused (from output)": "actconv_bwd_kernel<6, true>", "ext_actnorm_bwd": "ext_actnorm_bwd_tile_kernel<6", "prior_nll_bwd": "prior_nll_bwd_kernel",
            "logistic_log_prob_bwd": "logistic_log_prob_bwd_kernel", "sigmoid_flow_bwd": "sigmoid_flow_bwd_kernel",
            "affine_params_bwd": "affine_params_bwd_kernel<4, 2, true", "affine_transform_bwd": "affine_transform_bwd_kernel<4, 2"}
    manifest = {}
    for name, bpe, fn in rows:
        if name not in frag:
            continue
        for i in range(12):
            fn(i % R)
        torch.cuda.synchronize()
        manifest[frag[name]] = {"what": name + ", S*", "alg_bytes": bpe * elems, "elems": elems}
    if os.environ.get("CNF_MANIFEST"):
        json.dump(manifest, open(os.environ["CNF_MANIFEST"], "w"), indent=1)
    print("done")
    sys.exit(0)
print("shape B=%d N=%d D=%d (%.2f M elems); start-to-start over blocks of %d launches incl. the partials reduction launch where there is one"
      % (B, N, D, elems / 1e6, args.reps))
print("%-58s %6s %9s %9s %10s %8s" % ("kernel", "B/elem", "us (med)", "us (min)", "alg GB/s", "of 8TB/s"))
for name, bpe, fn in rows:
    med, mn = timeit(fn)
    print("%-58s %6d %9.2f %9.2f %10.0f %8.3f" % (name, bpe, med, mn, bpe * elems / med / 1e3, bpe * elems / med / 1e3 / 8000), flush=True)

if hasattr(lib, "cnf_stream_probe_bwd") and not args.only:
    print("\nstream ceiling for the affine backward's mix (16 B read + 12 B written per element, no arithmetic), us / TB/s")
    for hint in range(8):
        line = []
        for u in (1, 2):
            f = lambda i: call("cnf_stream_probe_bwd", _ptr(zo[i]), _ptr(nn2[i]), _ptr(gzo[i]), _ptr(gz[i]), _ptr(gnn[i]), elems, u, hint, st())
            t = timeit(f)[0]
            line.append("U=%d %6.2f us %.2f TB/s" % (u, t, 28 * elems / t / 1e6))
        print("hint %d (nt saved loads %d, nt upstream loads %d, nt stores %d): %s" % (hint, hint & 1, (hint >> 1) & 1, (hint >> 2) & 1, "   ".join(line)), flush=True)

if args.sweep:
    print("\nflat-tile knobs (cnf_set_bwd_tile): chunks in flight U x groups per wave G, us (median, start-to-start)")
    lib.cnf_set_affine_bwd_tiles(0)       # the U knob belongs to the flat-tile kernels
    kernels = (("affine sf", lambda i: affine(i, 0)), ("affine inv sf", lambda i: affine(i, 1)), ("affine no sf", lambda i: affine(i, 0, False)),
               ("actnorm", lambda i: actnorm(i, 0)), ("invconv", lambda i: invconv(i, 0)), ("prior_nll", nll), ("sigmoid", lambda i: sigmoid(i, 0)))
    print("%-6s %s" % ("U G", " ".join("%13s" % k for k, _ in kernels)))
    for u in (1, 2, 3):
        for grp in (1, 2, 4, 8):
            lib.cnf_set_bwd_tile(u, grp)
            print("%d %-4d %s" % (u, grp, " ".join("%13.2f" % timeit(f, reps=60, blocks=3)[0] for _, f in kernels)), flush=True)
    lib.cnf_set_bwd_tile(0, 0)
    lib.cnf_set_affine_bwd_tiles(1)

# run-to-run reproducibility of the parameter gradients
ok = True
for name, fn, outs in (("affine", lambda: affine(0, 0), (g_sf,)), ("actnorm", lambda: actnorm(0, 0, True), (g_b, g_s)),
                       ("invconv", lambda: invconv(0, 0, True), (g_w, g_sl)), ("affine_params", lambda: aff_params(0), (g_sf,))):
    ref = None
    for k in range(4):
        fn(); torch.cuda.synchronize()
        cur = [o.clone() for o in outs]
        if ref is None:
            ref = cur
        elif not all(torch.equal(a, b) for a, b in zip(ref, cur)):
            ok = False
            print("NOT reproducible:", name)
print("parameter gradients bit-identical over 4 runs:", ok)
