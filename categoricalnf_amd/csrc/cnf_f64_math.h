// fp64 log / log1p / reciprocal for the reference-precision mixture kernels (math mode 0), written for the argument
// ranges those kernels have.  The library (ocml) versions are general: log is double-double (43 v_add_f64 + 21 fma / mul
// + a full division, ~90 instructions), a division is 16 instructions with its scaling.  Every fp64 instruction costs 4
// issue cycles per wave (profiles/r05_op_rates.txt) and the kernels run at 0.6-0.8 of that issue ceiling, so the
// instruction count IS the run time.  Measured per call and SIMD (tools/f64_math_rates.py, profiles/r05_f64_math.txt):
// log 85 ns against the library's 185, reciprocal 34 against 42; the library's exp (58 ns) stays: two replacements were built and
// measured — the library's structure (degree-11 polynomial) with one clamp instead of its two range checks: 55 ns; a 64-entry
// 2^(j/64) table in LDS with a degree-5 polynomial: 49 ns by itself, 2 % on the forward kernel and a loss on the inverse (the
// table reads and index arithmetic raise its register pressure past three waves per SIMD).
// Polynomial: Chebyshev interpolant of (log((1+s)/(1-s)) - 2s) / s^3 in z = s^2, computed with mpmath at 200 bits
// (tools/f64_math_coeffs.py), truncation error 5e-18 relative.  Accuracy on the GPU against mpmath / numpy
// (tests/test_gpu_f64_math.py): reciprocal on [1, 2] correctly rounded on every sample, log and log1p <= 1 ulp.
#pragma once
#include <hip/hip_runtime.h>

namespace cnf {

// 1 / d for a normal d well inside the exponent range (no scaling): v_rcp_f64 and three Newton steps
__device__ __forceinline__ double rcp64(double d) {
    double y = __builtin_amdgcn_rcp(d);
    double e = fma(-d, y, 1.0);
    y = fma(y, e, y);
    e = fma(-d, y, 1.0);
    y = fma(y, e, y);
    e = fma(-d, y, 1.0);
    return fma(y, e, y);
}

// a / d under the same condition on d, |a| far from the overflow / underflow thresholds: quotient with one residual step
__device__ __forceinline__ double div64(double a, double d) {
    const double y = rcp64(d);
    const double q = a * y;
    const double rem = fma(-q, d, a);
    return fma(rem, y, q);
}

// log(1 + f) + e ln 2 for f in [sqrt(1/2) - 1, sqrt(2) - 1]: s = f / (2 + f), log(1 + f) = 2 s + s z R(z), z = s^2
__device__ __forceinline__ double log64_core(double f, double ed) {
    const double d = 2.0 + f;
    const double y = rcp64(d);
    double s = f * y;
    s = fma(fma(-s, d, f), y, s);                        // s = f / (2 + f)
    const double z = s * s;
    double R = 0x1.2ba1c63d07ce0p-3;
    R = fma(z, R, 0x1.39f866a142829p-3);
    R = fma(z, R, 0x1.7462e2edad3a5p-3);
    R = fma(z, R, 0x1.c71c62421e79cp-3);
    R = fma(z, R, 0x1.2492492e720f1p-2);
    R = fma(z, R, 0x1.9999999994e51p-2);
    R = fma(z, R, 0x1.5555555555558p-1);
    // 2 s = f - s f, so log(1 + f) = f - s (f - z R); e ln 2 with a 42-bit head (exact product) and its tail
    const double t = fma(-z, R, f);
    const double tail = fma(-s, t, ed * 0x1.ef35793c76730p-45);
    return fma(ed, 0x1.62e42fefa3800p-1, f + tail);
}

// log x for a positive normal x (NaN kept; x <= 0, subnormals and inf are the caller's business: the mixture kernels
// clamp at 1e-22 or branch at 1e-290 before they take a logarithm)
__device__ __forceinline__ double log64_pos(double x) {
    double m = __builtin_amdgcn_frexp_mant(x);           // [0.5, 1)
    int e = __builtin_amdgcn_frexp_exp(x);
    const bool low = m < 0x1.6a09e667f3bcdp-1;           // sqrt(1/2)
    m = low ? m + m : m;
    e = low ? e - 1 : e;
    return log64_core(m - 1.0, (double)e);               // m - 1 exact; m in [0.7071, 1.4142)
}

// log(1 + E) for 0 <= E <= 1 without forming 1 + E: f = E below sqrt(2) - 1, else (E - 1) / 2 with one factor of two
__device__ __forceinline__ double log1p64_unit(double E) {
    const bool hi = E > 0x1.a827999fcef32p-2;
    const double f = hi ? fma(E, 0.5, -0.5) : E;
    return log64_core(f, hi ? 1.0 : 0.0);
}

}  // namespace cnf
