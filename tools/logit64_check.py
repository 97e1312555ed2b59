"""accuracy of the fp64 logit of csrc/closures.hpp (frexp + nine-term atanh series), restated in numpy, against
40-digit arithmetic (mpmath): max absolute / relative error over c in (1e-6, 1 - 1e-6)"""
import numpy as np, mpmath

def logit64_one_division(c):
    """the form of csrc/closures.hpp since round 2: frexp of c and of 1 - c, one division"""
    b = 1.0 - c
    ma, ea = np.frexp(c)
    mb, eb = np.frexp(b)
    e = ea - eb
    lo = ma < 0.70710678118654752440 * mb
    hi = ma > 1.41421356237309504880 * mb
    k = np.where(lo, 1, np.where(hi, -1, 0))
    ma = np.ldexp(ma, k)
    e = e - k
    s = (ma - mb) / (ma + mb); z = s * s
    p = np.full_like(z, 1.0 / 19.0)
    for d in (17, 15, 13, 11, 9, 7, 5, 3):
        p = p * z + 1.0 / d
    lm = (s + s) + (s + s) * (p * z)
    return e * 6.93147180369123816490e-01 + (lm + e * 1.90821492927058770002e-10)


def logit64(c):
    r = c / (1.0 - c)
    m, e = np.frexp(r)
    lo = m < 0.70710678118654752440
    m = np.where(lo, 2.0 * m, m); e = np.where(lo, e - 1, e)
    s = (m - 1.0) / (m + 1.0); z = s * s
    p = np.full_like(z, 1.0 / 19.0)
    for d in (17, 15, 13, 11, 9, 7, 5, 3):
        p = p * z + 1.0 / d
    lm = (s + s) + (s + s) * (p * z)
    return e * 6.93147180369123816490e-01 + (lm + e * 1.90821492927058770002e-10)

rng = np.random.default_rng(0)
c = np.concatenate([rng.uniform(1e-6, 1 - 1e-6, 20000), np.linspace(0.05, 0.95, 2001), 0.5 + 1e-3 * rng.standard_normal(2000)])
mpmath.mp.dps = 40
exact = []
for ci in c:
    x = mpmath.mpf(float(ci)); exact.append(mpmath.log(x / (1 - x)))
for name, fn in (("two divisions (round 1)", logit64), ("one division", logit64_one_division)):
    g = fn(c)
    ea = er = 0.0
    for gi, ex in zip(g, exact):
        d = abs(mpmath.mpf(float(gi)) - ex)
        ea = max(ea, float(d)); er = max(er, float(d / abs(ex)) if abs(ex) > 1e-3 else 0.0)
    print("%-24s max abs err %.3e  max rel err (|logit| > 1e-3) %.3e" % (name, ea, er))
