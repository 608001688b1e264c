"""Shared test helpers: golden-fixture loading and comparison utilities."""
import os

import numpy as np

from svgpfa_amd.utils import synthetic

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    """Return (problem, ref) where problem is the numpy layout of svgpfa_amd.utils.synthetic
    and ref the dict of reference outputs stored next to it."""
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    K = int(z["K"])
    prob = dict(R=int(z["R"]), N=int(z["N"]), K=K, Q=int(z["Q"]), Ms=[int(m) for m in z["Ms"]],
                reg=float(z["reg"]),
                kernel_types=[synthetic.RBF if t == 0 else synthetic.PERIODIC for t in z["kernel_types"]],
                scales=[float(x) for x in z["scales"]],
                lengthscale_scales=[float(x) for x in z["lengthscale_scales"]],
                period_scales=[float(x) for x in z["period_scales"]],
                spike_offsets=z["spike_offsets"].astype(np.int64), spike_times=z["spike_times"],
                spike_neuron=z["spike_neuron"].astype(np.int32),
                quad_points=z["quad_points"], quad_weights=z["quad_weights"],
                C=z["C"], d=z["d"],
                Z=[z[f"Z_{k}"] for k in range(K)], mean=[z[f"mean_{k}"] for k in range(K)],
                cholVecs=[z[f"cholVecs_{k}"] for k in range(K)],
                theta=[z[f"theta_{k}"] for k in range(K)])
    ref = {k: z[k] for k in z.files}
    return prob, ref


def rel_err(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    den = max(np.max(np.abs(b)) if b.size else 0.0, 1e-300)
    return float(np.max(np.abs(a - b)) / den) if b.size else 0.0
