"""Per-step view of ECM at the north-star size: wall-clock, closures and iterations of every conditional maximisation,
next to the time of the same number of bare closures (evalFlat in a loop, no optimiser): the optimiser's own cost."""
import io, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from helpers import build_model
from svgpfa_amd.stats import svEM
from svgpfa_amd.utils import synthetic

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 2
prob = synthetic.make_config("cfg3")
step = dict(max_iter=20, lr=1.0, tolerance_grad=1e-7, tolerance_change=1e-9, line_search_fn="strong_wolfe")
optim = {"em_max_iter": iters}
for name in ("estep", "mstep_embedding", "mstep_kernels", "mstep_indpointslocs"):
    optim[f"{name}_estimate"] = True
    optim[f"{name}_optim_params"] = dict(step)
rows = []
em = svEM.SVEM_PyTorch()
orig = em._lbfgs_step
def timed(x, evalFunc, optim_params, per_trial, model=None, group=None):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    res = orig(x, evalFunc, optim_params, per_trial, model=model, group=group)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    rows.append((group, dt, res["nfeval"], res["niter"]))
    return res
em._lbfgs_step = timed
for rep in range(2):          # second pass is the measured one (first touches every code path)
    rows.clear()
    model = build_model(prob)
    t0 = time.perf_counter()
    hist, _, info, _ = em.maximize(model=model, optim_params=optim, method="ECM", out=io.StringIO(), verbose=False)
    torch.cuda.synchronize()
    total = time.perf_counter() - t0
bare = {}
for group in ("var", "emb", "theta", "Z"):
    if group == "emb":
        stats = model.computeSVPosteriorOnLatentsStats()
    def closure():
        if group in ("theta", "Z"):
            model.buildKernelsMatrices()
        return model.evalFlat(group)
    try:
        for _ in range(5): closure()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(40): closure()
        torch.cuda.synchronize(); bare[group] = (time.perf_counter() - t0) / 40
    except Exception as e:
        bare[group] = float("nan"); print("bare", group, "failed:", e)
print(f"{iters} ECM iterations {total:.3f} s, lowerBound {hist[0]:.2f} -> {hist[-1]:.2f}")
for g, dt, nf, ni in rows:
    print(f"  {g:6s} {1e3*dt:8.2f} ms  closures {nf:3d}  iterations {ni:3d}  per closure {1e3*dt/max(nf,1):.3f} ms   bare closure {1e3*bare[g]:.3f} ms  ratio {dt/max(nf,1)/bare[g]:.2f}")
