"""HBM rate of the optimiser's fused vector kernels (svgpfa_lbfgs_*) on the E-step group of the north-star size
(2.8 M doubles = 22.4 MB per vector), against the algorithmic bytes of each call."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np
import torch
from svgpfa_amd.stats.shardedLBFGS import _HipVec, _TorchVec

n = 500 * 10 * (32 + 32 * 33 // 2)
gen = torch.Generator().manual_seed(0)
x = torch.randn(n, dtype=torch.float64, generator=gen).cuda()
rnd = lambda *s: torch.randn(*s, dtype=torch.float64, generator=gen).cuda()
g, pg, d, d2, x0 = rnd(n), rnd(n), rnd(n), torch.empty(n, dtype=torch.float64, device="cuda"), torch.empty(n, dtype=torch.float64, device="cuda")
val = torch.tensor(1.0, dtype=torch.float64, device="cuda")
for cls in (_HipVec, _TorchVec):
    vec = cls(x, 0, 1.0, 100)
    print(cls.__name__)
    for m in (0, 5, 10, 20):
        S, Y = rnd(m + 1, n), rnd(m + 1, n)
        coef = np.linspace(-1.0, 1.0, 2 * m + 1)
        calls = {
            "grad_stats": (lambda: vec.grad_stats(g, -1.0, d, val, None), 3 * n * 8),
            "pair_dots": (lambda: vec.pair_dots(g, pg, d, 0.5, S, Y, m), (2 * m + 3 + 2) * n * 8),
            "direction": (lambda: vec.direction(g, S, Y, m, coef, d2, x, x0), (2 * m + 2 + 2) * n * 8),
        }
        for name, (fn, nbytes) in calls.items():
            if name == "grad_stats" and m:
                continue
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                fn()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 20
            print(f"  m = {m:2d}  {name:10s} {1e6 * dt:8.1f} us   {nbytes / 1e6:8.1f} MB  {nbytes / dt / 1e12:5.2f} TB/s")
