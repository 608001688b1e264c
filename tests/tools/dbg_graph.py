import os, sys, torch
os.environ["SVGPFA_SERIES_MIN_PAIRS"] = "0"
os.environ["SVGPFA_GRAPH"] = sys.argv[1] if len(sys.argv) > 1 else "1"
sys.path.insert(0, os.getcwd())
from svgpfa_amd.utils import synthetic
from svgpfa_amd.engine import engine_from_problem
prob = synthetic.make_problem(R=6, N=12, K=3, M=[10, 14, 8], kernels="mix", seed=31, n_quad=40, rate=25.0, jitter=1.0)
eng, P = engine_from_problem(prob)
shared = torch.empty(eng.shared_len, dtype=torch.float64, device=P["C"].device)
grads = eng.alloc_grads(15)
for it in range(5):
    print("it", it, flush=True)
    eng.elbo(P["mean"], P["cholVecs"], P["C"], P["d"], 15, out=shared, grads=grads)
    eng.sync()
    print(float(shared[0]), eng.graph_replays(), flush=True)
