"""Near-minimax polynomial coefficients (mpmath chebyfit, 60 digits) for the device math:
   2^f on [-1/2, 1/2];  sin(pi/2 f)/f and cos(pi/2 f) as polynomials in g = f^2 on [0, 1/4].
Prints C initialisers and the max relative / absolute error of a fp64 Horner evaluation."""
import mpmath as mp
import numpy as np
mp.mp.dps = 60

def fit(fun, lo, hi, deg):
    c = mp.chebyfit(fun, [lo, hi], deg + 1)     # highest degree first
    return [mp.mpf(x) for x in c]

def horner64(coefs_hi_first, x):
    acc = np.full_like(x, np.float64(coefs_hi_first[0]))
    for c in coefs_hi_first[1:]:
        acc = acc * x + np.float64(c)           # numpy has no fma; close enough for a bound
    return acc

x = np.linspace(-0.5, 0.5, 200001)
for deg in (10, 11, 12):
    c = fit(lambda t: mp.mpf(2) ** t, -0.5, 0.5, deg)
    approx = horner64(c, x)
    exact = np.array([float(mp.mpf(2) ** mp.mpf(float(t))) for t in x[::200]])
    err = np.max(np.abs(approx[::200] / exact - 1))
    print("exp2 degree", deg, "max rel err", err)
    if deg == 10:
        print("EXP2_C (low->high):", ", ".join(mp.nstr(v, 20) for v in c[::-1]))

g = np.linspace(0, 0.25, 100001)
for deg in (6, 7, 8):
    cs = fit(lambda t: mp.sin(mp.pi / 2 * mp.sqrt(t)) / mp.sqrt(t) if t > 0 else mp.pi / 2, 0, 0.25, deg)
    cc = fit(lambda t: mp.cos(mp.pi / 2 * mp.sqrt(t)), 0, 0.25, deg)
    f = np.sqrt(g[::100])
    s_ex = np.array([float(mp.sin(mp.pi / 2 * mp.mpf(float(t)))) for t in f])
    c_ex = np.array([float(mp.cos(mp.pi / 2 * mp.mpf(float(t)))) for t in f])
    s_ap = horner64(cs, g[::100]) * f
    c_ap = horner64(cc, g[::100])
    print("sincos degree", deg, "max abs err sin", np.max(np.abs(s_ap - s_ex)), "cos", np.max(np.abs(c_ap - c_ex)))
    if deg == 7:
        print("SIN_C (low->high):", ", ".join(mp.nstr(v, 20) for v in cs[::-1]))
        print("COS_C (low->high):", ", ".join(mp.nstr(v, 20) for v in cc[::-1]))
