"""Cost of the per-evaluation all-reduce of the shared vector through torch.distributed (RCCL), one rank:
the fixed per-call overhead (stream hand-off + collective launch) that bounds strong scaling of a ~0.3 ms step."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
import torch.distributed as dist
from svgpfa_amd.engine import engine_from_problem
from svgpfa_amd.utils import synthetic
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
prob = synthetic.shard_trials(synthetic.make_config("cfg3"), 0, world)
eng, P = engine_from_problem(prob)
shared = torch.empty(eng.shared_len, dtype=torch.float64, device=eng.device)
grads = eng.alloc_grads(15)
def step(mode):
    eng.build_kernels(P["Z"], P["theta"])
    eng.elbo(P["mean"], P["cholVecs"], P["C"], P["d"], 15, out=shared, grads=grads)
    if mode == "sync":
        dist.all_reduce(shared)
    elif mode == "async":
        return dist.all_reduce(shared, async_op=True)
def run(mode, n=200):
    for _ in range(10): step(mode)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    w = None
    for _ in range(n):
        w = step(mode)
    t_host = time.perf_counter() - t0
    if w is not None: w.wait()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n, 1e3 * t_host / n
for mode in ("none", "sync", "async"):
    t, th = run(mode)
    print(f"shard 1/{world}: step with all_reduce={mode:5s}: {t:.3f} ms (host enqueue {th:.3f} ms)")
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200): dist.all_reduce(shared)
torch.cuda.synchronize()
print(f"all_reduce alone: {1e3 * (time.perf_counter() - t0) / 200:.3f} ms per call")
dist.destroy_process_group()
