"""KernelsMatricesStore surface of the reference (/root/reference/src/svGPFA/stats/kernelsMatricesStore.py:8-221)
as thin views over a HIP engine.  The matrices are produced by the engine's getter kernels
(svgpfa_get_kzz / _kzz_chol / _ktz_quad / _ktz_spikes); nothing is computed on the host.

A store used stand-alone (as the reference's tests do, stats/tests/test_kernelsMatricesStore.py) creates a
minimal private engine; inside a model the SVLowerBound facade owns the engine and these classes are not
needed on the hot path (Ktz is never materialised there)."""
import numpy as np
import torch

from .. import _lib
from ..engine import Engine, SVGPFAEngineError


class KernelsMatricesStore:

    def __init__(self, device=None):
        self._device = torch.device("cuda", torch.cuda.current_device() if device is None else device)
        self._kernels = None
        self._ind_points_locs = None
        self._engine = None

    def setKernels(self, kernels):
        self._kernels = kernels

    def setInitialParams(self, initial_params):
        self.setIndPointsLocs(ind_points_locs=initial_params["inducing_points_locs0"])
        self.setKernelsParams(kernels_params=initial_params["kernels_params0"])

    def setKernelsParams(self, kernels_params):
        for k in range(len(self._kernels)):
            self._kernels[k].setParams(kernels_params[k])

    def setIndPointsLocs(self, ind_points_locs):
        self._ind_points_locs = ind_points_locs

    def getIndPointsLocs(self):
        return self._ind_points_locs

    def getKernels(self):
        return self._kernels

    def getKernelsParams(self):
        return [k.getParams() for k in self._kernels]

    # -- private: a minimal engine holding (Z, theta) and optionally times
    def _dev(self, t):
        return torch.as_tensor(t).detach().to(device=self._device, dtype=torch.float64).contiguous()

    def _make_engine(self, reg, quad=None, spikes=None):
        Z = [self._dev(z) for z in self._ind_points_locs]
        R = Z[0].shape[0]
        Q = 1 if quad is None else quad.shape[1]
        kt = [k._engine_type for k in self._kernels]
        eng = Engine(R, 1, Q, [z.shape[1] for z in Z], kt, [float(k._scale) for k in self._kernels],
                     [float(k._lengthscaleScale) for k in self._kernels],
                     [float(getattr(k, "_periodScale", 1.0)) for k in self._kernels], device=self._device.index)
        eng.set_reg(reg)
        q = self._dev(quad).reshape(R, Q) if quad is not None else torch.zeros(R, 1, dtype=torch.float64, device=self._device)
        eng.set_quad(q, torch.ones_like(q))
        if spikes is None:
            eng.set_spikes(np.zeros(R + 1, dtype=np.int64), None, None)
        else:
            off = np.concatenate([[0], np.cumsum([len(s) for s in spikes])]).astype(np.int64)
            t = torch.cat([self._dev(s).reshape(-1) for s in spikes]) if off[-1] else None
            n = torch.zeros(int(off[-1]), dtype=torch.int32, device=self._device) if off[-1] else None
            eng.set_spikes(off, t, n)
            self._spike_off = off
        try:
            eng.build_kernels(Z, [self._dev(k.getParams()) for k in self._kernels])
            eng.sync()
        except SVGPFAEngineError as e:
            eng.close()                                        # do not leave the half-built engine to the GC
            if e.code == _lib.ERR_NOT_PD:                      # the reference raises from torch.linalg.cholesky
                raise torch.linalg.LinAlgError("linalg.cholesky: " + str(e)) from None
            raise
        if self._engine is not None:
            self._engine.close()
        self._engine = eng
        return eng


class IndPointsLocsKMS_Chol(KernelsMatricesStore):
    """kernelsMatricesStore.py:90-138."""

    def setRegParam(self, reg_param):
        self._reg_param = reg_param

    def getRegParam(self):
        return self._reg_param

    def buildKernelsMatrices(self):
        eng = self._make_engine(float(self._reg_param))
        K = len(self._kernels)
        self._Kzz = [eng.get_kzz(k) for k in range(K)]
        self._Kzz_inv = [eng.get_kzz_chol(k) for k in range(K)]     # the reference stores the Cholesky factor here
        eng.sync()

    def getKzz(self):
        return self._Kzz

    def solveForLatent(self, input, latentIndex):
        return torch.cholesky_solve(input, self._Kzz_inv[latentIndex])

    def solveForLatentAndTrial(self, input, latentIndex, trialIndex):
        return torch.cholesky_solve(input, self._Kzz_inv[latentIndex][trialIndex, :, :])


class IndPointsLocsKMS_PInv(KernelsMatricesStore):
    """kernelsMatricesStore.py:146-165: Kzz + eps I and its Moore-Penrose pseudo-inverse (utils/miscUtils.py:218-223,
    rcond = 1e-15), solves by multiplication.  Stand-alone store for tests / plots: Kzz comes from the kernel objects'
    own buildKernelMatrix on the GPU and the pseudo-inverse from torch.linalg.pinv (batched over trials) -- no Cholesky
    factor is involved, so a singular Kzz is handled as in the reference.  (Inside a model, kernelMatrixInvPInv runs the
    engine's Cholesky-based whitened algebra: for a positive definite Kzz the pseudo-inverse IS the inverse, see
    SVGPFAModelFactory.)"""

    def setRegParam(self, reg_param):
        self._reg_param = reg_param

    def getRegParam(self):
        return self._reg_param

    def buildKernelsMatrices(self):
        self._Kzz, self._Kzz_inv = [], []
        for k, kern in enumerate(self._kernels):
            Z = self._dev(self._ind_points_locs[k])
            saved = kern.getParams()
            kern.setParams(self._dev(saved))
            try:
                Kzz = kern.buildKernelMatrix(X1=Z) + float(self._reg_param) * torch.eye(
                    Z.shape[1], dtype=Z.dtype, device=Z.device)
            finally:
                kern.setParams(saved)
            self._Kzz.append(Kzz)
            self._Kzz_inv.append(torch.linalg.pinv(Kzz, rcond=1e-15))

    def getKzz(self):
        return self._Kzz

    def solveForLatent(self, input, latentIndex):
        return torch.matmul(self._Kzz_inv[latentIndex], input)

    def solveForLatentAndTrial(self, input, latentIndex, trialIndex):
        return torch.matmul(self._Kzz_inv[latentIndex][trialIndex, :, :], input)


class IndPointsLocsAndTimesKMS(KernelsMatricesStore):

    def setTimes(self, times):
        self._t = times

    def getKtz(self):
        return self._Ktz

    def getKttDiag(self):
        return self._KttDiag


class IndPointsLocsAndAllTimesKMS(IndPointsLocsAndTimesKMS):
    """kernelsMatricesStore.py:183-203: times n_trials x n_quad x 1."""

    def buildKernelsMatrices(self):
        t = self._dev(self._t)
        eng = self._make_engine(1.0, quad=t.reshape(t.shape[0], -1))   # Kzz is not used by this store
        K = len(self._kernels)
        self._Ktz = [eng.get_ktz_quad(k) for k in range(K)]
        eng.sync()
        self._KttDiag = torch.stack([float(k._scale) ** 2 * torch.ones(t.shape[0], t.shape[1], dtype=t.dtype,
                                                                       device=t.device) for k in self._kernels], dim=2)


class IndPointsLocsAndAssocTimesKMS(IndPointsLocsAndTimesKMS):
    """kernelsMatricesStore.py:206-221: times = list over trials of 1-D spike-time tensors."""

    def buildKernelsMatrices(self):
        eng = self._make_engine(1.0, spikes=self._t)   # Kzz is not used by this store
        K, R = len(self._kernels), len(self._t)
        off = self._spike_off
        self._Ktz, self._KttDiag = [], []
        for k in range(K):
            full = eng.get_ktz_spikes(k) if off[-1] else torch.zeros(0, eng.Ms[k], dtype=torch.float64, device=self._device)
            self._Ktz.append([full[off[r]:off[r + 1]] for r in range(R)])
            self._KttDiag.append([float(self._kernels[k]._scale) ** 2 *
                                  torch.ones(int(off[r + 1] - off[r]), dtype=torch.float64, device=self._device)
                                  for r in range(R)])
        eng.sync()
