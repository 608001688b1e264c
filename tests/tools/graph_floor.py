"""Does a captured graph (hipGraph through torch.cuda.CUDAGraph) shorten one rank's step on small shards?
Modes: plain launches / graph replay of the same single-stream sequence / graph replay with the spike pass forked onto
the side stream (SVGPFA_OVERLAP=1), where the cross-stream edges become graph dependencies instead of event waits."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
from svgpfa_amd.engine import engine_from_problem
from svgpfa_amd.utils import synthetic
full = synthetic.make_config("cfg3")
for world in (1, 4, 8):
    prob = synthetic.shard_trials(full, 0, world)
    for overlap in ("0", "1"):
        os.environ["SVGPFA_OVERLAP"] = overlap
        eng, P = engine_from_problem(prob)
        shared = torch.empty(eng.shared_len, dtype=torch.float64, device=eng.device)
        grads = eng.alloc_grads(15)
        def step():
            eng.build_kernels(P["Z"], P["theta"])
            eng.elbo(P["mean"], P["cholVecs"], P["C"], P["d"], 15, out=shared, grads=grads)
        for _ in range(5): step()
        eng.sync()
        ref = float(shared[0])
        def timeit(fn, n=100):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(n): fn()
            torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - t0) / n
        t_plain = timeit(step)
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            for _ in range(3): step()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                step()
            for _ in range(3): g.replay()
            torch.cuda.synchronize()
            t_graph = timeit(g.replay)
        ok = abs(float(shared[0]) - ref) <= 1e-12 * abs(ref)
        print(f"world {world} overlap {overlap}: plain {t_plain:.3f} ms  graph {t_graph:.3f} ms  (same ELBO: {ok})", flush=True)
        del g
        eng.close()
