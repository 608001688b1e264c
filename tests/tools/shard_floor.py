"""Per-step time of one rank's shard at 1/2/4/8-way sharding of cfg3 (emulated on one GPU, no all-reduce):
exposes the host-side floor (Python marshalling + kernel launches) that limits strong scaling.
    python tests/tools/shard_floor.py [--config cfg3|cfg5] [--dtype f64|mixed] [world ...]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import gc
import torch
from svgpfa_amd.engine import engine_from_problem
from svgpfa_amd.utils import synthetic
args = sys.argv[1:]
config, precision = "cfg3", "fp64"
while args and args[0].startswith("--"):
    if args[0] == "--config":
        config = args[1]
    elif args[0] == "--dtype":
        precision = "mixed" if args[1] == "mixed" else "fp64"
    args = args[2:]
full = synthetic.make_config(config)
nsteps = 100 if config != "cfg5" else 10
gc.collect()
gc.freeze()     # (a full pass of the cyclic collector over torch's heap is a 40-ms stall: tests/tools/step_jitter.py)
worlds = [int(a) for a in args] or [1, 2, 4, 8]
for world in worlds:
    prob = synthetic.shard_trials(full, 0, world)
    eng, P = engine_from_problem(prob, precision=precision)
    shared = torch.empty(eng.shared_len, dtype=torch.float64, device=eng.device)
    grads = eng.alloc_grads(15)
    def step():
        eng.build_kernels(P["Z"], P["theta"])
        eng.elbo(P["mean"], P["cholVecs"], P["C"], P["d"], 15, out=shared, grads=grads)
    for _ in range(30 if config != "cfg5" else 5): step()
    eng.sync()
    best, best_enq = 1e9, 1e9
    for rep in range(3):                       # best of three runs of nsteps steps (a busy host core shows up as an outlier)
        t0 = time.perf_counter()
        for _ in range(nsteps): step()
        t_enq = time.perf_counter() - t0
        torch.cuda.synchronize()
        t = time.perf_counter() - t0
        if t < best:
            best, best_enq = t, t_enq
    print(f"{config} {precision} world {world}: trials {prob['R']:4d}  step {1e3*best/nsteps:.3f} ms  (host enqueue only {1e3*best_enq/nsteps:.3f} ms)  elbo {float(shared[0]):.6f}")
    eng.close()
