import sys, time, gc
sys.path.insert(0, ".")
import torch
from svgpfa_amd.engine import engine_from_problem
from svgpfa_amd.utils import synthetic
prob = synthetic.make_config("cfg3")
eng, P = engine_from_problem(prob)
shared = torch.empty(eng.shared_len, dtype=torch.float64, device=eng.device)
grads = eng.alloc_grads(15)
def step():
    eng.build_kernels(P["Z"], P["theta"])
    eng.elbo(P["mean"], P["cholVecs"], P["C"], P["d"], 15, out=shared, grads=grads)
gc.collect(); gc.freeze()
for _ in range(50): step()
eng.sync()
for mode in (0, 1, 0, 1, 8, 0):
    eng.enable_timing(mode)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(300): step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 300
    ms, n = eng.kernel_ms(1)
    print("timing", mode, "step %.4f ms" % (1e3 * dt), "k_item_bwd launches timed", n, "avg %.4f" % (ms / max(n, 1)))
