"""One-off high-precision referee (mpmath, 40 digits): dELBO/dZ for trial 0, latent 0 of the
MATLAB-geometry fixture by central differences, to decide which fp64 result (reference autograd
vs HIP engine) carries the error on this ill-conditioned case (cond(Kzz) ~ 1.6e5 .. 1e6)."""
import sys, os
import numpy as np
import mpmath as mp
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from helpers import load_golden

mp.mp.dps = 40
prob, ref = load_golden("matlab_estep")
r = 0
K, N, Q = prob["K"], prob["N"], prob["Q"]
off = prob["spike_offsets"]
st = [mp.mpf(float(x)) for x in prob["spike_times"][off[r]:off[r + 1]]]
sn = prob["spike_neuron"][off[r]:off[r + 1]]
tq = [mp.mpf(float(x)) for x in prob["quad_points"][r]]
wq = [mp.mpf(float(x)) for x in prob["quad_weights"][r]]
C = [[mp.mpf(float(x)) for x in row] for row in prob["C"]]
d = [mp.mpf(float(x)) for x in prob["d"]]
eps = mp.mpf(prob["reg"])


def kappa(k, x1, x2):
    th = prob["theta"][k]
    if prob["kernel_types"][k] == "periodic":
        l, p = mp.mpf(float(th[0])), mp.mpf(float(th[1]))
        return mp.e ** (-2 * mp.sin(mp.pi * (x1 - x2) / p) ** 2 / l ** 2)
    l = mp.mpf(float(th[0]))
    return mp.e ** (-(x1 - x2) ** 2 / (2 * l ** 2))


def elbo_trial(Z):
    muq = [[0] * K for _ in range(Q)]
    varq = [[0] * K for _ in range(Q)]
    mus = [[0] * K for _ in range(len(st))]
    kl = mp.mpf(0)
    for k in range(K):
        M = prob["Ms"][k]
        z = Z[k]
        Kzz = mp.matrix(M, M)
        for i in range(M):
            for j in range(M):
                Kzz[i, j] = kappa(k, z[i], z[j]) + (eps if i == j else 0)
        Kinv = Kzz ** -1
        m = mp.matrix([mp.mpf(float(x)) for x in prob["mean"][k][r]])
        L = mp.matrix(M, M)
        cv = prob["cholVecs"][k][r]
        idx = 0
        for i in range(M):
            for j in range(i + 1):
                L[i, j] = mp.mpf(float(cv[idx])); idx += 1
        S = L * L.T
        a = Kinv * m
        W = Kinv * S * Kinv - Kinv
        for q in range(Q):
            kv = mp.matrix([kappa(k, tq[q], z[j]) for j in range(M)])
            muq[q][k] = (kv.T * a)[0]
            varq[q][k] = 1 + (kv.T * W * kv)[0]
        for s in range(len(st)):
            kv = mp.matrix([kappa(k, st[s], z[j]) for j in range(M)])
            mus[s][k] = (kv.T * a)[0]
        kl += (mp.mpf(1) / 2) * ((Kinv * S).T.T[0, 0] * 0 + sum((Kinv * S)[i, i] for i in range(M)) + (m.T * a)[0]
                                 + mp.log(mp.det(Kzz)) - mp.log(mp.det(S)) - M)
    ell = mp.mpf(0)
    for q in range(Q):
        acc = mp.mpf(0)
        for n in range(N):
            h = d[n]
            for k in range(K):
                h += C[n][k] * muq[q][k] + C[n][k] ** 2 * varq[q][k] / 2
            acc += mp.e ** h
        ell -= wq[q] * acc
    for s in range(len(st)):
        n = int(sn[s])
        ell += d[n] + sum(C[n][k] * mus[s][k] for k in range(K))
    return ell - kl


Z0 = [[mp.mpf(float(x)) for x in prob["Z"][k][r]] for k in range(K)]
h = mp.mpf(10) ** -15
comps = [0, 1, 2, 4]
print("component  highprec           reference(fp64 autograd)")
for j in comps:
    Zp = [list(z) for z in Z0]; Zm = [list(z) for z in Z0]
    Zp[0][j] += h; Zm[0][j] -= h
    g = (elbo_trial(Zp) - elbo_trial(Zm)) / (2 * h)
    print(j, mp.nstr(g, 15), "%.10f" % ref["g_Z_0"][r, j])
