"""cProfile of one ECM iteration at the north-star size with the default optimiser (host-side view)."""
import cProfile, io, os, pstats, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from helpers import build_model
from svgpfa_amd.stats.svEM import SVEM_PyTorch
from svgpfa_amd.utils import synthetic
prob = synthetic.make_config("cfg3")
step = dict(max_iter=20, lr=1.0, tolerance_grad=1e-7, tolerance_change=1e-9, line_search_fn="strong_wolfe")
optim = {"em_max_iter": 1}
for name in ("estep", "mstep_embedding", "mstep_kernels", "mstep_indpointslocs"):
    optim[f"{name}_estimate"] = True
    optim[f"{name}_optim_params"] = dict(step)
model = build_model(prob)
SVEM_PyTorch().maximize(model=model, optim_params=optim, method="ECM", out=io.StringIO(), verbose=False)
model = build_model(prob)
pr = cProfile.Profile()
pr.enable()
SVEM_PyTorch().maximize(model=model, optim_params=optim, method="ECM", out=io.StringIO(), verbose=False)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
