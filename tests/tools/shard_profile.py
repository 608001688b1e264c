import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
from svgpfa_amd.engine import engine_from_problem
from svgpfa_amd.utils import synthetic
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
full = synthetic.make_config("cfg3")
prob = synthetic.shard_trials(full, 0, world)
eng, P = engine_from_problem(prob)
shared = torch.empty(eng.shared_len, dtype=torch.float64, device=eng.device)
grads = eng.alloc_grads(15)
for _ in range(30):
    eng.build_kernels(P["Z"], P["theta"])
    eng.elbo(P["mean"], P["cholVecs"], P["C"], P["d"], 15, out=shared, grads=grads)
eng.sync()
