"""Parity against the oracle as Kzz + eps I becomes ill-conditioned (M = 32 equidistant inducing points on [0, 1],
RBF lengthscale 0.3 / periodic, eps from 1e-3 down to 1e-6): errors of the ELBO and of each gradient group."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
from helpers import rel_err
from oracle import svgpfa_oracle as orc
from svgpfa_amd.engine import engine_from_problem
from svgpfa_amd.utils import synthetic
for reg in (1e-3, 1e-4, 1e-5, 1e-6):
    prob = synthetic.make_problem(R=2, N=12, K=2, M=32, kernels="mix", seed=3, n_quad=100, rate=20.0, jitter=0.5, reg=reg)
    p = orc.from_numpy_problem(prob)
    Kzz = orc.build_kzz(p)[0] if hasattr(orc, "build_kzz") else None
    eng, P = engine_from_problem(prob)
    shared, gm, gc, gz = eng.elbo(P["mean"], P["cholVecs"], P["C"], P["d"], 15)
    eng.sync()
    ref = orc.elbo_and_grads(p, 15)
    conds = [float(np.linalg.cond(eng.get_kzz(k).cpu().numpy()[0])) for k in range(2)]
    sh = eng.split_shared(shared)
    line = f"eps {reg:g} cond {conds[0]:.1e}/{conds[1]:.1e} elbo {abs(float(shared[0]) - float(ref['elbo'])) / abs(float(ref['elbo'])):.1e}"
    for k in range(2):
        line += f" | k{k} g_m {rel_err(gm[k].cpu().numpy(), ref['g_mean'][k].numpy()[:, :, 0]):.1e} g_S {rel_err(gc[k].cpu().numpy(), ref['g_cholVecs'][k].numpy()[:, :, 0]):.1e} g_Z {rel_err(gz[k].cpu().numpy(), ref['g_Z'][k].numpy()[:, :, 0]):.1e}"
    print(line)
    eng.close()
