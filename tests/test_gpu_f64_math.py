"""csrc/cnf_f64_math.h (the log / log1p / reciprocal of the reference-precision mixture kernels) against mpmath and numpy.

mixture_cdf_layer.py:62,173-178 computes in fp64 with torch's CPU / CUDA log: <= 1 ulp functions.  The replacements must be as
good on the arguments the kernels feed them: positive normals for log, [0, 1] for log1p, [1, 2.5] for the reciprocal."""
import numpy as np
import pytest
import torch

from categoricalnf_amd import _lib

pytestmark = pytest.mark.gpu


def _run(which, x):
    lib = _lib.load()
    dev = torch.device("cuda:0")
    xi = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float64)).to(dev)
    out = torch.empty_like(xi)
    assert lib.cnf_probe_f64_math(which, xi.data_ptr(), out.data_ptr(), xi.numel(), 1, None) == 0
    torch.cuda.synchronize()
    return out.cpu().numpy()


def _ulp(got, ref):
    return np.abs(got - ref) / np.spacing(np.abs(ref))


def _exact(fn, xs):
    import mpmath as mp
    mp.mp.prec = 120
    return np.array([float(fn(mp.mpf(float(v)))) for v in xs])


def test_log_of_positive_normals():
    import mpmath as mp
    rng = np.random.default_rng(1)
    n = 1 << 18
    x = np.concatenate([rng.uniform(0.5, 2.0, n), np.exp(rng.uniform(-50, 50, n)), 1.0 - 10.0 ** rng.uniform(-14, -1, n),
                        1.0 + 10.0 ** rng.uniform(-14, -1, n), np.exp(rng.uniform(-700, 700, n)),
                        [1.0, 2.2250738585072014e-308, 1e-22, 1e-290, 1.7976931348623157e308, 0.7071067811865475, 0.7071067811865476]])
    got = _run(0, x)
    ref = np.log(x)
    nz = ref != 0
    assert _ulp(got[nz], ref[nz]).max() <= 2.0                 # numpy itself is within 1 ulp of the truth
    assert np.all(got[~nz] == 0.0)
    idx = rng.choice(x.size, 3000, replace=False)
    idx = idx[ref[idx] != 0]
    assert _ulp(got[idx], _exact(mp.log, x[idx])).max() <= 1.0
    assert np.isnan(_run(0, np.array([np.nan])))[0]


def test_log1p_on_the_unit_interval():
    import mpmath as mp
    rng = np.random.default_rng(2)
    n = 1 << 18
    x = np.concatenate([rng.uniform(0.0, 1.0, n), 10.0 ** rng.uniform(-300, 0, n), [0.0, 1.0, 0.41421356237309503, 0.4142135623730951]])
    got = _run(2, x)
    ref = np.log1p(x)
    nz = ref != 0
    assert _ulp(got[nz], ref[nz]).max() <= 2.0
    assert np.all(got[~nz] == 0.0)
    idx = rng.choice(np.nonzero(nz)[0], 3000, replace=False)
    assert _ulp(got[idx], _exact(mp.log1p, x[idx])).max() <= 1.0


def test_reciprocal_is_correctly_rounded_on_its_range():
    rng = np.random.default_rng(3)
    x = np.concatenate([rng.uniform(1.0, 2.0, 1 << 19), rng.uniform(1.7, 2.5, 1 << 17), [1.0, 2.0, 1.0 + 2.0 ** -52, 2.0 - 2.0 ** -52]])
    got = _run(1, x)
    assert np.array_equal(got, 1.0 / x)

