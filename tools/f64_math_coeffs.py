"""Coefficients of csrc/cnf_f64_math.h: Chebyshev interpolant (degree 6 in z = s^2) of R(z) = (log((1+s)/(1-s)) - 2 s) / s^3 on
|s| <= 0.1716 * 1.005 (m in [sqrt(1/2), sqrt(2)), s = (m-1)/(m+1)), in 200-bit arithmetic; prints them as C hex doubles with the
largest relative error of 2 s + s^3 R(s^2) against log((1+s)/(1-s)), and the 42-bit head / tail of ln 2."""
import mpmath as mp
mp.mp.prec = 200

def cheb_fit(f, a, b, deg):
    n = deg + 1
    ts = [(b - a) / 2 * mp.cos(mp.pi * (2 * k + 1) / (2 * n)) + (a + b) / 2 for k in range(n)]
    A = mp.matrix(n, n)
    for i, t in enumerate(ts):
        for j in range(n):
            A[i, j] = t ** j
    c = mp.lu_solve(A, mp.matrix([f(t) for t in ts]))
    return [c[i] for i in range(n)]

zmax = mp.mpf("0.1716") ** 2 * mp.mpf("1.01")
def R(z):
    s = mp.sqrt(z)
    return (mp.log((1 + s) / (1 - s)) - 2 * s) / (s * z)
cl = cheb_fit(R, 0, zmax, 6)
err = 0
for i in range(1, 2001):
    z = zmax * i / 2000
    s = mp.sqrt(z)
    err = max(err, abs((2 * s + s * z * sum(c * z ** k for k, c in enumerate(cl))) / mp.log((1 + s) / (1 - s)) - 1))
print("log polynomial: relative error %s" % mp.nstr(err, 4))
for k, c in enumerate(cl):
    print("   z^%d  %s" % (k, float(c).hex()))
l2 = mp.log(2)
hi = mp.floor(l2 * 2 ** 42) / 2 ** 42
print("ln 2 = %s + %s" % (float(hi).hex(), float(l2 - hi).hex()))
print("sqrt(1/2) %s   sqrt(2) - 1 %s" % (float(mp.sqrt(mp.mpf(1) / 2)).hex(), float(mp.sqrt(2) - 1).hex()))
