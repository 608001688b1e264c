"""One rank's shard of cfg3 at WORLD-way sharding, STEPS evaluations (for rocprofv3 --kernel-trace --stats):
python tests/tools/shard_one.py WORLD [STEPS]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
from svgpfa_amd.engine import engine_from_problem
from svgpfa_amd.utils import synthetic
world = int(sys.argv[1]); steps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
prob = synthetic.shard_trials(synthetic.make_config("cfg3"), 0, world)
eng, P = engine_from_problem(prob)
shared = torch.empty(eng.shared_len, dtype=torch.float64, device=eng.device)
grads = eng.alloc_grads(15)
def step():
    eng.build_kernels(P["Z"], P["theta"])
    eng.elbo(P["mean"], P["cholVecs"], P["C"], P["d"], 15, out=shared, grads=grads)
for _ in range(5): step()
eng.sync()
t0 = time.perf_counter()
for _ in range(steps): step()
torch.cuda.synchronize()
print(f"world {world}: trials {prob['R']}  step {1e3*(time.perf_counter()-t0)/steps:.3f} ms")
