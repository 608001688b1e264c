"""Trial additivity of cfg5 in both precision modes: the measured error behind test_cfg5_properties' tolerances."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np, torch
from svgpfa_amd.engine import engine_from_problem
from svgpfa_amd.utils import synthetic
prob = synthetic.make_config("cfg5")
for precision in ("fp64", "mixed"):
    eng, P = engine_from_problem(prob, precision=precision)
    full = eng.elbo(P["mean"], P["cholVecs"], P["C"], P["d"], 15)[0]
    eng.sync(); full = full.cpu().numpy(); eng.close(); del eng, P; torch.cuda.empty_cache()
    acc = np.zeros_like(full)
    for rank in range(2):
        e2, P2 = engine_from_problem(synthetic.shard_trials(prob, rank, 2), precision=precision)
        s2 = e2.elbo(P2["mean"], P2["cholVecs"], P2["C"], P2["d"], 15)[0]
        e2.sync(); acc += s2.cpu().numpy(); e2.close(); del e2, P2; torch.cuda.empty_cache()
    print(precision, "ELBO rel", abs(acc[0] - full[0]) / abs(full[0]), "gradients rel (max-norm)",
          np.abs(acc[3:] - full[3:]).max() / np.abs(full[3:]).max())
