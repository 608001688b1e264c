"""Device memory across repeated engine life cycles (create / set data / build / evaluate / destroy)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np, torch
from svgpfa_amd.engine import Engine, engine_from_problem
from svgpfa_amd.utils import synthetic
prob = synthetic.make_problem(R=8, N=20, K=4, M=16, kernels="mix", seed=1, n_quad=64, rate=20.0)
def free():
    torch.cuda.synchronize(); torch.cuda.empty_cache(); return torch.cuda.mem_get_info()[0]
def cycle(stage):
    if stage == "create":
        e = Engine(prob["R"], prob["N"], prob["Q"], prob["Ms"], [0 if t == synthetic.RBF else 1 for t in prob["kernel_types"]])
        e.close(); return
    eng, P = engine_from_problem(prob)
    if stage == "eval":
        eng.elbo(P["mean"], P["cholVecs"], P["C"], P["d"], 15); eng.sync()
    eng.close()
for stage in ("create", "setup", "eval"):
    cycle(stage)
    f0 = free()
    for _ in range(100): cycle(stage)
    print(f"{stage:7s}: {(f0 - free()) / 100 / 1e3:.1f} KB per cycle")
