"""Host helpers of the reference's utils/miscUtils.py that callers of the hot path need
(/root/reference/src/svGPFA/utils/miscUtils.py:38-50,226-236,422-445)."""
import numpy as np
import torch

from ..stats import kernels as _kernels
from . import synthetic


def buildKernels(kernels_types, kernels_params):
    """miscUtils.py:38-50."""
    out = []
    for k, kind in enumerate(kernels_types):
        if kind == "exponentialQuadratic":
            kern = _kernels.ExponentialQuadraticKernel()
        elif kind == "periodic":
            kern = _kernels.PeriodicKernel()
        else:
            raise ValueError(f"Invalid kernels type: {kind}")
        kern.setParams(kernels_params[k])
        out.append(kern)
    return out


def getLegQuadPointsAndWeights(n_quad, trials_start_times, trials_end_times, dtype=torch.double):
    """miscUtils.py:226-236 (gcnu_common.leggaussVarLimits == numpy Gauss-Legendre mapped to [a, b])."""
    pts, wts = synthetic.leg_quad(n_quad, np.asarray(trials_start_times, dtype=np.float64),
                                  np.asarray(trials_end_times, dtype=np.float64))
    return (torch.from_numpy(pts).to(dtype).unsqueeze(-1), torch.from_numpy(wts).to(dtype).unsqueeze(-1))


def getVectorRepOfLowerTrianMatrices(lt_matrices):
    """miscUtils.py:422-445: row-major lower-triangle vectorisation, n_trials x P x 1 per latent."""
    out = []
    for L in lt_matrices:
        M = L.shape[1]
        idx = torch.tril_indices(M, M)
        out.append(L[:, idx[0], idx[1]].unsqueeze(-1).contiguous())
    return out


def buildEquidistantIndPointsLocs0(n_latents, n_trials, n_ind_points, trials_start_times, trials_end_times):
    """utils/initUtils.py:913-923."""
    out = []
    for k in range(n_latents):
        Z = torch.empty((n_trials, n_ind_points[k], 1), dtype=torch.double)
        for r in range(n_trials):
            Z[r, :, 0] = torch.linspace(float(trials_start_times[r]), float(trials_end_times[r]), n_ind_points[k])
        out.append(Z)
    return out
