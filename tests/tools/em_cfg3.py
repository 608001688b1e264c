"""End-to-end ECM at the north-star size (500 trials x 200 neurons x 10 latents x 32 inducing points, 2.0 M spikes):
wall-clock of SVEM_PyTorch.maximize with the reference's optimiser settings (max_iter = 20 per step), both optimisers.
(The reference needs 20 s per E-step closure and 99 s per kernel-step closure on 8 CPU cores, BASELINE.md.)"""
import io, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from helpers import build_model
from svgpfa_amd.stats.svEM import SVEM_PyTorch
from svgpfa_amd.utils import synthetic

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 2
prob = synthetic.make_config("cfg3")
step = dict(max_iter=20, lr=1.0, tolerance_grad=1e-7, tolerance_change=1e-9, line_search_fn="strong_wolfe")
optim = {"em_max_iter": iters}
for name in ("estep", "mstep_embedding", "mstep_kernels", "mstep_indpointslocs"):
    optim[f"{name}_estimate"] = True
    optim[f"{name}_optim_params"] = dict(step)
for which in ("warm-up", "torch", "sharded"):
    if which == "warm-up":      # first use of every code path in this process (module loads, allocator growth): not timed
        for w in ("torch", "sharded"):
            SVEM_PyTorch(optimizer=w).maximize(model=build_model(prob), optim_params=dict(optim, em_max_iter=1), method="ECM",
                                               out=io.StringIO(), verbose=False)
        torch.cuda.synchronize()
        continue
    t0 = time.time()
    model = build_model(prob)
    t_setup = time.time() - t0
    log = io.StringIO()
    t0 = time.time()
    hist, _, info, _ = SVEM_PyTorch(optimizer=which).maximize(model=model, optim_params=optim, method="ECM", out=log,
                                                              verbose=True)
    torch.cuda.synchronize()
    dt = time.time() - t0
    nfe = sum(int(line.split("nfeval:")[1]) for line in log.getvalue().splitlines() if "nfeval:" in line)
    print(f"optimizer={which}: setParamsAndData {t_setup:.1f} s, {iters} ECM iterations {dt:.2f} s, {nfe} closure evals "
          f"({1e3 * dt / max(nfe, 1):.2f} ms each), lowerBound {hist[0]:.2f} -> {hist[-1]:.2f}  [{info.message}]", flush=True)
    del model
    torch.cuda.empty_cache()
