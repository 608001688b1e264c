"""Host time of every step of the bench loop (build + evaluation enqueue), to find stalls: prints the steps that took more
than 5x the median, with and without the timing hook.   usage: python tests/tools/step_jitter.py [cfg2] [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from svgpfa_amd.utils import synthetic
from svgpfa_amd.engine import engine_from_problem
cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
eng, P = engine_from_problem(synthetic.make_config(cfg))
shared = torch.empty(eng.shared_len, dtype=torch.float64, device=P["C"].device)
grads = eng.alloc_grads(15)
import gc
_t = {}
def _cb(phase, info):
    if phase == "start":
        _t["s"] = time.perf_counter()
    elif info["generation"] == 2:
        print(f"  gc generation 2: {1e3 * (time.perf_counter() - _t['s']):.1f} ms, collected {info['collected']}", flush=True)
gc.callbacks.append(_cb)
if os.environ.get("JITTER_FREEZE"):
    gc.collect()
    gc.freeze()
def step():
    eng.build_kernels(P["Z"], P["theta"])
    eng.elbo(P["mean"], P["cholVecs"], P["C"], P["d"], 15, out=shared, grads=grads)
for timing in (False, True, False):
    for _ in range(30):
        step()
    eng.sync()
    eng.enable_timing(timing)
    torch.cuda.synchronize()
    ts = [time.perf_counter()]
    for _ in range(n):
        step()
        ts.append(time.perf_counter())
    torch.cuda.synchronize()
    tot = time.perf_counter() - ts[0]
    d = [1e6 * (b - a) for a, b in zip(ts, ts[1:])]
    med = sorted(d)[len(d) // 2]
    print(f"timing {timing}: {1e3 * tot / n:.4f} ms/step, median host {med:.1f} us; slow steps:",
          [(i, round(x)) for i, x in enumerate(d) if x > 5 * med])
    eng.enable_timing(False)
