"""Errors of precision mode "mixed" (fp32 spike pass + quadrature ELL, fp64 Cholesky / whitening / KL) against the fp64
engine and the oracle: the numbers behind the tolerances stated in DESIGN.md and asserted in tests/test_gpu_mixed.py."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import torch
from svgpfa_amd.engine import engine_from_problem
from svgpfa_amd.utils import synthetic
from helpers import rel_err

def run(prob, precision):
    eng, P = engine_from_problem(prob, precision=precision)
    shared, gm, gc, gz = eng.elbo(P["mean"], P["cholVecs"], P["C"], P["d"], 15)
    eng.sync()
    out = [shared.cpu().numpy()] + [torch.cat([t.reshape(-1) for t in grp]).cpu().numpy() for grp in (gm, gc, gz)]
    eng.close()
    return out

cases = {
    "cfg2": synthetic.make_config("cfg2"),
    "cfg3 x 20 trials": synthetic.make_config("cfg3", n_trials=20),
    "cfg5 x 4 trials": synthetic.make_config("cfg5", n_trials=4),
    "jitter mix": synthetic.make_problem(R=3, N=40, K=7, M=[48, 20, 33, 9, 16, 40, 5], kernels="mix", n_quad=50, rate=10.0, seed=7, jitter=1.0),
    "long trial (6 s, 15 periods)": synthetic.make_problem(R=2, N=6, K=2, M=[20, 24], kernels="periodic", n_quad=120, rate=6.0, seed=8, trial_len=6.0, jitter=1.0),
    "long trial 50 s rbf": synthetic.make_problem(R=2, N=6, K=2, M=[32, 32], kernels="rbf", n_quad=200, rate=6.0, seed=9, trial_len=50.0, jitter=1.0),
}
for name, prob in cases.items():
    a, b = run(prob, "fp64"), run(prob, "mixed")
    N, K = prob["N"], prob["K"]
    s64, s32 = a[0], b[0]
    print(f"{name:30s} ELBO rel {abs(s32[0]-s64[0])/abs(s64[0]):.2e}  ELL {abs(s32[1]-s64[1])/abs(s64[1]):.2e}  KL {abs(s32[2]-s64[2])/abs(s64[2]):.2e} | "
          f"dC {rel_err(s32[3:3+N*K], s64[3:3+N*K]):.2e} dd {rel_err(s32[3+N*K:3+N*K+N], s64[3+N*K:3+N*K+N]):.2e} "
          f"dtheta {rel_err(s32[3+N*K+N:], s64[3+N*K+N:]):.2e} dmean {rel_err(b[1], a[1]):.2e} dchol {rel_err(b[2], a[2]):.2e} dZ {rel_err(b[3], a[3]):.2e}", flush=True)
