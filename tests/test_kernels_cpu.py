"""The covariance-function objects of svgpfa_amd.stats.kernels (parameter holders for the engine; buildKernelMatrix /
buildKernelMatrixDiag are utilities for callers outside the ELBO path) against the reference's own Kzz
(tests/golden/*.npz: stats/kernelsMatricesStore.py:107-117 output, i.e. kernels.py:33-46 / :73-85 plus eps I) -- the
counterpart of the reference's stats/tests/test_kernels.py, which checks the same methods against MATLAB dumps."""
import numpy as np
import pytest
import torch

from helpers import build_kernels, load_golden, rel_err


@pytest.mark.parametrize("name", ["ragged_small", "matlab_estep"])
def test_build_kernel_matrix_matches_the_reference_kzz(name):
    prob, ref = load_golden(name)
    kernels = build_kernels(prob)
    for k, kern in enumerate(kernels):
        kern.setParams(torch.from_numpy(np.asarray(prob["theta"][k], dtype=np.float64)))
        Z = torch.from_numpy(prob["Z"][k]).unsqueeze(-1)                 # R x M x 1, the reference's layout
        K3 = kern.buildKernelMatrix(X1=Z) + prob["reg"] * torch.eye(Z.shape[1], dtype=torch.double)
        assert rel_err(K3.numpy(), ref[f"Kzz_{k}"]) < 1e-12
        K2 = kern.buildKernelMatrix(X1=Z[0, :, 0], X2=Z[0, :, 0])        # the 2-D form (kernels.py:40-46)
        assert rel_err(K2.numpy() + prob["reg"] * np.eye(Z.shape[1]), ref[f"Kzz_{k}"][0]) < 1e-12
        diag = kern.buildKernelMatrixDiag(X=Z)
        assert diag.shape == Z.shape and torch.allclose(diag[:, :, 0], torch.diagonal(K3, dim1=1, dim2=2) - prob["reg"])
        named = kern.getNamedParams()
        assert float(named["lengthscale"]) == float(prob["theta"][k][0])
        assert ("period" in named) == (len(prob["theta"][k]) == 2)
