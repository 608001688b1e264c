"""Covariance-function objects: parameter holders with the reference's interface
(/root/reference/src/svGPFA/stats/kernels.py:6-108).  The engine evaluates the kernels itself in
HIP (csrc/engine_common.h: kern_val / kern_eval); these objects only carry the hyper-parameter
tensor that the optimiser mutates in place, the fixed `scale`, and the *Scale rescalings."""
import math
from abc import ABC, abstractmethod

import torch

from .. import _lib


class Kernel(ABC):
    _engine_type = None

    def getParams(self):
        return self._params

    def setParams(self, params):
        self._params = params

    @abstractmethod
    def getNamedParams(self):
        pass

    @abstractmethod
    def buildKernelMatrix(self, X1, X2=None):
        pass

    def buildKernelMatrixDiag(self, X):
        """kernels.py:48-51 / :87-90 -- scale**2 everywhere."""
        return float(self._scale) ** 2 * torch.ones(X.shape, dtype=X.dtype, device=X.device)

    @staticmethod
    def _diff(X1, X2):
        if X2 is None:
            X2 = X1
        if X1.ndim == 3:
            return X1 - X2.transpose(1, 2)
        return X1.reshape(-1, 1) - X2.reshape(1, -1)


class ExponentialQuadraticKernel(Kernel):
    """kernels.py:26-64.  params = [lengthscale]."""
    _engine_type = _lib.KERNEL_RBF

    def __init__(self, scale=1.0, lengthscaleScale=1.0, dtype=torch.double):
        self._scale = torch.tensor(scale, dtype=dtype)
        self._lengthscaleScale = lengthscaleScale
        self._periodScale = 1.0

    def buildKernelMatrix(self, X1, X2=None):
        """Utility for callers outside the ELBO path (plots, simulations); the model never calls it."""
        ls = self._params[0] / self._lengthscaleScale
        return float(self._scale) ** 2 * torch.exp(-0.5 * self._diff(X1, X2) ** 2 / ls ** 2)

    def getScaledParams(self):
        return torch.tensor([self._params[0] / self._lengthscaleScale])

    def getNamedParams(self):
        return {"scale": self._scale, "lengthscale": self._params[0]}


class PeriodicKernel(Kernel):
    """kernels.py:66-108.  params = [lengthscale, period]."""
    _engine_type = _lib.KERNEL_PERIODIC

    def __init__(self, scale=1.0, lengthscaleScale=1.0, periodScale=1.0, dtype=torch.double):
        self._scale = torch.tensor(scale, dtype=dtype)
        self._lengthscaleScale = lengthscaleScale
        self._periodScale = periodScale

    def buildKernelMatrix(self, X1, X2=None):
        ls = self._params[0] / self._lengthscaleScale
        period = self._params[1] / self._periodScale
        rr = math.pi * self._diff(X1, X2) / period
        return float(self._scale) ** 2 * torch.exp(-2 * torch.sin(rr) ** 2 / ls ** 2)

    def getScaledParams(self):
        return torch.tensor([self._params[0] / self._lengthscaleScale, self._params[1] / self._periodScale])

    def getNamedParams(self):
        return {"scale": self._scale, "lengthscale": self._params[0], "period": self._params[1]}
