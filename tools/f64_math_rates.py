"""Accuracy (ulp against mpmath on a sample, against numpy on everything) and cost (ns per call per wave and SIMD) of the fp64
log / log1p / reciprocal of csrc/cnf_f64_math.h beside the library's.  GPU only."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from categoricalnf_amd import _lib
lib = _lib.load(); dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
n = 1 << 20

def run(which, x, reps=1):
    xi = torch.from_numpy(x).to(dev); out = torch.empty_like(xi)
    rc = lib.cnf_probe_f64_math(which, xi.data_ptr(), out.data_ptr(), xi.numel(), reps, None)
    assert rc == 0
    torch.cuda.synchronize()
    return out.cpu().numpy()

def ulp_err(got, ref):
    return np.abs(got - ref) / np.spacing(np.abs(ref))

def timed(which, x, reps=64):
    xi = torch.from_numpy(x).to(dev); out = torch.empty_like(xi)
    best = 1e9
    for _ in range(4):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); lib.cnf_probe_f64_math(which, xi.data_ptr(), out.data_ptr(), xi.numel(), reps, None); b.record()
        torch.cuda.synchronize(); best = min(best, a.elapsed_time(b))
    waves = xi.numel() / 64
    return best * 1e6 / (waves * reps / 1024)        # ns per call per SIMD

cases = {
    "log": (0, 4, np.concatenate([rng.uniform(0.5, 2.0, n // 4), np.exp(rng.uniform(-50, 50, n // 4)), 1.0 - 10.0 ** rng.uniform(-12, -1, n // 4),
                                  np.exp(rng.uniform(-660, 700, n // 4))]), np.log),
    "1/x [1,2]": (1, 5, rng.uniform(1.0, 2.0, n), lambda v: 1.0 / v),
    "log1p [0,1]": (2, 6, np.concatenate([rng.uniform(0.0, 1.0, n // 2), 10.0 ** rng.uniform(-300, 0, n // 2)]), np.log1p),
    "exp (library only)": (3, 3, -np.abs(np.concatenate([rng.uniform(0, 40, n // 2), rng.exponential(1.0, n // 2)])), np.exp),
}
print("%-10s %14s %14s %12s %12s" % ("function", "max ulp (ours)", "max ulp (lib)", "ns ours", "ns library"))
import mpmath as mp
mp.mp.prec = 120
for name, (w, wl, x, ref) in cases.items():
    with np.errstate(all="ignore"):
        r = ref(x)
    ours, libv = run(w, x), run(wl, x)
    ok = np.isfinite(r) & (r != 0) & (np.abs(r) > 1e-300)
    # numpy's own error is <= 1 ulp: the exact check on a sample with mpmath
    idx = rng.choice(np.nonzero(ok)[0], 4000, replace=False)
    f = {"exp": mp.exp, "log": mp.log, "log1p": mp.log1p}.get(name.split()[0], lambda v: 1 / v)
    exact = np.array([float(f(mp.mpf(float(v)))) for v in x[idx]])
    e_ours = ulp_err(ours[idx], exact).max(); e_lib = ulp_err(libv[idx], exact).max()
    e_all = ulp_err(ours[ok], r[ok]).max()
    edge = (~ok) & ~((ours == r) | (np.isnan(ours) & np.isnan(r)))
    print("%-11s %9.2f (all vs numpy %.2f) %9.2f %12.2f %12.2f   edge mismatches %d" % (name, e_ours, e_all, e_lib, timed(w, x), timed(wl, x), int(edge.sum())))
spec = np.array([np.nan, 1.0, 2.2250738585072014e-308, 1e-22, 1e-290, 1.7976931348623157e308])
print("log specials", spec, "->", run(0, spec), "numpy", np.log(spec))
