"""Shared test helpers: golden-fixture loading and comparison utilities."""
import os

import numpy as np
import torch

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


def build_kernels(prob):
    from svgpfa_amd.stats import kernels
    out = []
    for k in range(prob["K"]):
        sc = prob.get("scales", [1.0] * prob["K"])[k]
        ls = prob.get("lengthscale_scales", [1.0] * prob["K"])[k]
        ps = prob.get("period_scales", [1.0] * prob["K"])[k]
        if prob["kernel_types"][k] == synthetic.RBF:
            out.append(kernels.ExponentialQuadraticKernel(scale=sc, lengthscaleScale=ls))
        else:
            out.append(kernels.PeriodicKernel(scale=sc, lengthscaleScale=ls, periodScale=ps))
    return out


def build_model(prob, **factory_kw):
    """Build the model exactly as a reference caller would (CPU tensors in, nested spike lists)."""
    from svgpfa_amd.stats.svGPFAModelFactory import SVGPFAModelFactory
    t = lambda a: torch.from_numpy(np.array(a, dtype=np.float64, copy=True, order="C"))
    off, times, neuron = prob["spike_offsets"], prob["spike_times"], prob["spike_neuron"]
    measurements = []
    for r in range(prob["R"]):
        tt, nn = times[off[r]:off[r + 1]], neuron[off[r]:off[r + 1]]
        measurements.append([t(tt[nn == n]) for n in range(prob["N"])])
    initial_params = {
        "posterior_on_latents": {
            "posterior_on_ind_points": {"mean": [t(a).unsqueeze(-1) for a in prob["mean"]],
                                        "cholVecs": [t(a).unsqueeze(-1) for a in prob["cholVecs"]]},
            "kernels_matrices_store": {"kernels_params0": [t(a) for a in prob["theta"]],
                                       "inducing_points_locs0": [t(a).unsqueeze(-1) for a in prob["Z"]]}},
        "embedding": {"C0": t(prob["C"]), "d0": t(prob["d"])}}
    ell = {"leg_quad_points": t(prob["quad_points"]).unsqueeze(-1),
           "leg_quad_weights": t(prob["quad_weights"]).unsqueeze(-1)}
    model = SVGPFAModelFactory.buildModelPyTorch(kernels=build_kernels(prob), **factory_kw)
    model.setParamsAndData(measurements=measurements, initial_params=initial_params,
                           eLLCalculationParams=ell, priorCovRegParam=prob["reg"])
    return model
