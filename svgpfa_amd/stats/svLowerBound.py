"""SVLowerBound: the reference's model facade (/root/reference/src/svGPFA/stats/svLowerBound.py:5-120)
over the HIP engine.

What the reference spreads over an object graph (expectedLogLikelihood / klDivergence / svEmbedding /
svPosteriorOnLatents / svPosteriorOnIndPoints / kernelsMatricesStore) is ONE engine handle here; this
class keeps the protocol the EM loop relies on (svEM.py:93,106,202,219-263,290):

  * the lists returned by get*Params() hold the very tensor objects the model reads on the next
    eval(); the optimiser flips their requires_grad and mutates them in place;
  * eval() returns a 0-dim fp64 tensor whose backward() fills .grad of whichever parameter tensors
    have requires_grad=True (the engine computes those gradient groups natively, no autograd graph);
  * buildKernelsMatrices() snapshots Z / kernel params and rebuilds Kzz and its Cholesky factor;
    a non positive-definite Kzz raises torch.linalg.LinAlgError like utils/miscUtils.py:215 does.

Parameter tensors live on the GPU (fp64, contiguous); setInitialParams moves what it is given.
"""

import math
import warnings

import numpy as np
import torch

from .. import _lib
from ..engine import Engine, SVGPFAEngineError


def _to_dev(t, device, dtype=torch.float64):
    t = torch.as_tensor(t)
    return t.detach().to(device=device, dtype=dtype).contiguous()


def _pack_group(tensors, device):
    """One contiguous buffer for a parameter group; returns (flat, views).  The views are what get*Params() hands out:
    ordinary leaf tensors (in-place updates, requires_grad_ and torch.optim work on them as on any tensor) that happen
    to be laid out back to back, so that an optimiser step over the whole group can be ONE axpy on `flat`."""
    ts = [_to_dev(t, device) for t in tensors]
    flat = torch.empty(sum(t.numel() for t in ts), dtype=torch.float64, device=device)
    views, o = [], 0
    for t in ts:
        v = flat[o:o + t.numel()].view(t.shape)
        v.copy_(t)
        views.append(v)
        o += t.numel()
    return flat, views


class _BoundValue(torch.Tensor):
    """The 0-dim tensor eval() returns.  The reference warns inside eval() when the bound is infinite
    (svLowerBound.py:51-53) -- a host read per evaluation; here the same warning is raised where the value reaches the
    host anyway (float(), .item()), so evaluations stay asynchronous and a caller that drives this model with the
    reference's own svEM.py (float() of every closure value) still sees it."""

    @staticmethod
    def _checked(v):
        if isinstance(v, float) and math.isinf(v):
            warnings.warn("infinity lower bound detected")
        return v

    def item(self):
        return _BoundValue._checked(torch.Tensor.item(self))

    def __float__(self):
        return _BoundValue._checked(float(torch.Tensor.item(self)))


class _SumOverRanks(torch.autograd.Function):
    """y = sum over ranks of x (every rank gets y).  Each rank's own x enters the sum with weight one, so the adjoint of
    the local x is the adjoint of y."""

    @staticmethod
    def forward(ctx, model, x):
        y = x.detach().clone().reshape(1)
        model._reduce(y)
        return y.reshape(x.shape)

    @staticmethod
    def backward(ctx, g):
        return None, g


class _SharedParam(torch.autograd.Function):
    """Identity on a parameter that every rank holds (C, d, kernel parameters); its gradient is the sum of the ranks'
    contributions."""

    @staticmethod
    def forward(ctx, model, p):
        ctx.model = model
        return p.view_as(p)

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous().clone()
        ctx.model._reduce(g.view(-1))
        return None, g


class _ELBOFunction(torch.autograd.Function):
    """eval() + hand-derived backward of the engine as one autograd node.
    inputs: K means, K cholVecs, C, d, K kernel-param tensors, K inducing-point tensors."""

    @staticmethod
    def forward(ctx, model, *params):
        K = model._K
        need = ctx.needs_input_grad[1:]
        mask = 0
        if any(need[0:2 * K]):
            mask |= _lib.GRAD_VARIATIONAL
        if any(need[2 * K:2 * K + 2]):
            mask |= _lib.GRAD_EMBEDDING
        if any(need[2 * K + 2:3 * K + 2]):
            mask |= _lib.GRAD_KERNELS
        if any(need[3 * K + 2:4 * K + 2]):
            mask |= _lib.GRAD_INDPOINTS
        eng = model._engine
        # content id of C: torch bumps _version on every in-place update (optimiser steps), a new tensor has a new address
        eng.set_embedding_token(model._embedding_token())
        mean_t = [p.detach() for p in params[0:K]]
        chol_t = [p.detach().contiguous() for p in params[K:2 * K]]
        # nothing here waits for the GPU: the only device status an evaluation can raise (a failed Cholesky) is checked
        # where the factorisation happens, in buildKernelsMatrices()
        shared, gm, gc, gz = eng.elbo(mean_t, chol_t, model._C, model._d, mask)
        if model._reduce is not None:      # trials sharded over ranks: ONE all-reduce of the shared vector
            model._reduce(shared)
        ctx.mask, ctx.K = mask, K
        ctx.shapes = [p.shape for p in params]
        _, _, _, gC, gd, gth = eng.split_shared(shared)
        ctx.grads = (gm, gc, gC, gd, gth, gz)
        model._last_terms = (shared[0].clone(), shared[1].clone(), shared[2].clone())
        return shared[0].clone()

    @staticmethod
    def backward(ctx, gout):
        K, mask = ctx.K, ctx.mask
        gm, gc, gC, gd, gth, gz = ctx.grads
        need = ctx.needs_input_grad[1:]
        sh = ctx.shapes
        out = [None]
        for i in range(4 * K + 2):
            g = None
            if need[i]:
                if i < K:
                    g = gm[i]
                elif i < 2 * K:
                    g = gc[i - K]
                elif i == 2 * K:
                    g = gC
                elif i == 2 * K + 1:
                    g = gd
                elif i < 3 * K + 2:
                    g = gth[i - 2 * K - 2]
                else:
                    g = gz[i - 3 * K - 2]
                g = (gout * g).reshape(sh[i])
            out.append(g)
        return tuple(out)


class _ELLGivenStatsFunction(torch.autograd.Function):
    """ELL with frozen latent stats (the embedding M-step closure, svEM.py:225-232)."""

    @staticmethod
    def forward(ctx, model, C, d):
        want = any(ctx.needs_input_grad[1:])
        out = model._engine.ell_given_stats(C, d, want)
        if model._reduce is not None:
            model._reduce(out)
        N, K = model._N, model._K
        ctx.gC = out[1:1 + N * K].view(N, K)
        ctx.gd = out[1 + N * K:]
        ctx.dshape = d.shape
        return out[0].clone()

    @staticmethod
    def backward(ctx, gout):
        gC = gout * ctx.gC if ctx.needs_input_grad[1] else None
        gd = (gout * ctx.gd).reshape(ctx.dshape) if ctx.needs_input_grad[2] else None
        return None, gC, gd


class _LatentStatsFunction(torch.autograd.Function):
    """Posterior mean / variance of every latent at the quadrature nodes and at the spikes, and the KL term, as ONE
    autograd node whose forward (svgpfa_latent_stats) and backward (svgpfa_latent_stats_backward: whitened chain, Kzz
    path, KL) run in the engine.  A likelihood that the engine does not fuse -- the Gauss-Hermite link -- is written on
    top of it in torch; autograd supplies d ELL / d statistics and this node carries them to the parameters.
    inputs: K means, K cholVecs, K kernel-param tensors, K inducing-point tensors.
    outputs: mu_quad, var_quad (R x Q x K), mu_spk, var_spk (S x K), KL (0-dim)."""

    @staticmethod
    def forward(ctx, model, *params):
        K, eng = model._K, model._engine
        ctx.model, ctx.K = model, K
        ctx.shapes = [p.shape for p in params]
        ctx.mean_t = [p.detach() for p in params[0:K]]
        ctx.chol_t = [p.detach().contiguous() for p in params[K:2 * K]]
        muq, varq, mus, vars_ = eng.latent_stats(ctx.mean_t, ctx.chol_t)
        return muq, varq, mus, vars_, eng.kl()

    @staticmethod
    def backward(ctx, g_muq, g_varq, g_mus, g_vars, g_kl):
        K, eng = ctx.K, ctx.model._engine
        need = ctx.needs_input_grad[1:]
        mask = 0
        if any(need[0:2 * K]):
            mask |= _lib.GRAD_VARIATIONAL
        if any(need[2 * K:3 * K]):
            mask |= _lib.GRAD_KERNELS
        if any(need[3 * K:4 * K]):
            mask |= _lib.GRAD_INDPOINTS
        # the engine returns E(g) = J^T g_stats - dKL; the node needs J^T g_stats + g_kl dKL:
        #   g_kl != 0:  s E(g_stats / s) with s = -g_kl  (one engine call; the usual case, loss = ELL - KL has g_kl = -1)
        #   g_kl == 0 (KL not in the loss, or weighted by 0):  E(g_stats) - E(0)  (two calls; dividing by s would give NaN)
        # g_kl is read on the host: this node is the general-likelihood path, not the fused hot path.
        s_host = float(g_kl) if g_kl is not None else 0.0
        muq, varq, mus, vars_ = (torch.empty(eng.R, eng.Q, K, dtype=torch.float64, device=eng.device),) * 2 + \
                                (torch.empty(eng.S, K, dtype=torch.float64, device=eng.device),) * 2

        def engine_grads(scale):
            z = lambda g, like: (torch.zeros_like(like) if (g is None or scale == 0.0) else g / scale)
            shared, gm, gc, gz = eng.latent_stats_backward(
                ctx.mean_t, ctx.chol_t, z(g_muq, muq).contiguous(), z(g_varq, varq).contiguous(),
                z(g_mus, mus).contiguous(), z(g_vars, vars_).contiguous(), mask)
            gth = eng.split_shared(shared)[5]
            return [(gm[i] if i < K else gc[i - K] if i < 2 * K else gth[i - 2 * K] if i < 3 * K else gz[i - 3 * K])
                    if need[i] else None for i in range(4 * K)]

        if s_host != 0.0:
            s = -s_host
            gs = [None if g is None else s * g for g in engine_grads(s)]
        else:
            a = [None if g is None else g.clone() for g in engine_grads(1.0)]
            b = engine_grads(0.0)
            gs = [None if x is None else x - y for x, y in zip(a, b)]
        return (None,) + tuple(None if g is None else g.reshape(ctx.shapes[i]) for i, g in enumerate(gs))


class SVLowerBound:

    def __init__(self, device=None, rank1_plus_diag=False, precision="fp64", link=None):
        if not torch.cuda.is_available():
            raise RuntimeError("svgpfa_amd.stats.SVLowerBound needs a ROCm GPU; there is no CPU fallback")
        self._device = torch.device("cuda", torch.cuda.current_device() if device is None else device)
        self._engine = None
        self._engine_key = None
        # q(u) covariance representation: Cholesky vectors (default) or S = q q^T + diag(d^2)
        # (reference svPosteriorOnIndPoints.py:86-119, factory switch indPointsCovRank1PlusDiag)
        self._rank1 = bool(rank1_plus_diag)
        # "fp64" = the reference's arithmetic; "mixed" = fp32 spike pass and quadrature ELL, fp64 Kzz / Cholesky / KL
        # (BASELINE config 5; include/svgpfa.h SVGPFA_PRECISION_MIXED)
        if precision not in ("fp64", "mixed"):
            raise ValueError("precision must be 'fp64' or 'mixed'")
        self._precision = precision
        # link = None: exponential link, everything fused in the engine.  A callable: the point-process ELL with
        # Gauss-Hermite quadrature over the embedding's posterior (reference expectedLogLikelihood.py:215-255,
        # PointProcessELLQuad), written in torch on top of the engine's latent statistics (_LatentStatsFunction)
        self._link = link
        self._herm = None
        self._qSVec = self._qSDiag = None
        self._kernels = None
        self._reg = None
        self._quad = None
        self._spikes = None
        self._stats_token = None
        self._reduce = None          # optional callable(tensor): in-place sum over ranks (trial sharding)
        self._agree = None           # optional callable(int) -> int: maximum over ranks (error agreement)
        self._flat = {}              # parameter group -> its one contiguous buffer (see _pack_group)
        self._gflat = {}             # per-trial gradient buffers the engine writes into, same layout
        self._token_salt = 0
        self._last_terms = None
        self._data_version = 0
        self._engine_data_version = -1

    # ----------------------------------------------------------------- setters (svLowerBound.py:13-45,83-99)
    def setParamsAndData(self, measurements, initial_params, eLLCalculationParams, priorCovRegParam):
        self.setMeasurements(measurements=measurements)
        self.setInitialParams(initial_params=initial_params)
        self.setELLCalculationParams(eLLCalculationParams=eLLCalculationParams)
        self.setPriorCovRegParam(priorCovRegParam=priorCovRegParam)
        self.buildKernelsMatrices()

    def setKernels(self, kernels):
        self._kernels = kernels
        self._K = len(kernels)

    def setMeasurements(self, measurements):
        """measurements[r][n]: 1-D array-like of spike times.  Vectorised CSR packing in the reference's
        stacking order (expectedLogLikelihood.py:157-173): neuron-major inside each trial."""
        R = len(measurements)
        N = len(measurements[0])
        times, neuron, offsets = [], [], np.zeros(R + 1, dtype=np.int64)
        for r in range(R):
            if len(measurements[r]) != N:
                raise ValueError("every trial must list the same number of neurons")
            row = measurements[r]
            if N and all(isinstance(x, torch.Tensor) for x in row):
                # the reference's format (a tensor per neuron): one concatenation per trial instead of one host
                # conversion per (trial, neuron) -- 100 000 of them at the north-star size
                lens = np.fromiter((x.numel() for x in row), dtype=np.int64, count=N)
                times.append(torch.cat([x.detach().reshape(-1) for x in row]).to(device="cpu", dtype=torch.float64).numpy())
            else:
                arrs = [np.asarray(x.detach().cpu() if isinstance(x, torch.Tensor) else x, dtype=np.float64).reshape(-1)
                        for x in row]
                lens = np.fromiter((a.size for a in arrs), dtype=np.int64, count=N)
                times.append(np.concatenate(arrs) if N else np.zeros(0))
            neuron.append(np.repeat(np.arange(N, dtype=np.int32), lens))
            offsets[r + 1] = offsets[r] + int(lens.sum())
        self._R_data, self._N = R, N
        self._spikes = (offsets, np.concatenate(times) if R else np.zeros(0),
                        np.concatenate(neuron) if R else np.zeros(0, dtype=np.int32))
        self._data_version += 1

    def setInitialParams(self, initial_params):
        qU = initial_params["posterior_on_latents"]["posterior_on_ind_points"]
        kms = initial_params["posterior_on_latents"]["kernels_matrices_store"]
        emb = initial_params["embedding"]
        dev = self._device
        self._flat, self._gflat = {}, {}
        K = len(qU["mean"])
        if self._rank1:
            self._mean = [_to_dev(t, dev) for t in qU["mean"]]
            self._qSVec = [_to_dev(t, dev) for t in qU["qSVec0"]]
            self._qSDiag = [_to_dev(t, dev) for t in qU["qSDiag0"]]
            self._cholVecs = None
        else:
            # each optimisation group of the EM loop in ONE buffer, in the order its getter lists the tensors
            self._flat["var"], v = _pack_group(list(qU["mean"]) + list(qU["cholVecs"]), dev)
            self._mean, self._cholVecs = v[:K], v[K:]
        self._flat["Z"], self._Z = _pack_group(kms["inducing_points_locs0"], dev)
        self._flat["theta"], theta = _pack_group(kms["kernels_params0"], dev)
        for k, kern in enumerate(self._kernels):
            kern.setParams(theta[k])
        self._flat["emb"], (self._C, self._d) = _pack_group([emb["C0"], emb["d0"]], dev)
        self._token_salt += 1        # new embedding tensors: never reuse spike sums of the old ones

    def setELLCalculationParams(self, eLLCalculationParams):
        self._quad = (_to_dev(eLLCalculationParams["leg_quad_points"], self._device),
                      _to_dev(eLLCalculationParams["leg_quad_weights"], self._device))
        if "hermQuadPoints" in eLLCalculationParams:        # expectedLogLikelihood.py:224-225
            self._herm = (_to_dev(eLLCalculationParams["hermQuadPoints"], self._device).reshape(-1),
                          _to_dev(eLLCalculationParams["hermQuadWeights"], self._device).reshape(-1))
        self._data_version += 1

    def setPriorCovRegParam(self, priorCovRegParam):
        self._reg = float(priorCovRegParam)
        if self._engine is not None:
            self._engine.set_reg(self._reg)

    def setIndPointsLocs(self, locs):
        self._flat["Z"], self._Z = _pack_group(locs, self._device)
        self._gflat.pop("Z", None)

    def _embedding_token(self):
        """Content id of C for the engine's reuse of the spike sums across value-only evaluations: torch bumps _version on
        every in-place update of C or of the buffer it is a view of (optimiser steps); the salt changes whenever C is
        re-created, so a new tensor at a recycled address is never mistaken for the old one.  Contract: C is mutated
        through in-place torch ops (which bump the counter), not through `.data` or raw pointers."""
        C = self._C
        return (hash((C.data_ptr(), C._version, self._token_salt)) & 0x7FFFFFFFFFFFFFFF) | 1

    # ----------------------------------------------------------------- getters (svLowerBound.py:101-120)
    def getSVPosteriorOnIndPointsParams(self):
        if self._rank1:                                             # svPosteriorOnIndPoints.py:93-100
            return list(self._mean) + list(self._qSVec) + list(self._qSDiag)
        return list(self._mean) + list(self._cholVecs)          # svPosteriorOnIndPoints.py:37-42

    def _chol_vecs(self):
        """Cholesky vectors the engine consumes.  Rank-1 + diagonal representation: S = q q^T + diag(d^2)
        (svPosteriorOnIndPoints.py:104-119), factorised by torch on the GPU; the result stays attached to the autograd
        graph, so the engine's gradient wrt the Cholesky vectors flows on to qSVec / qSDiag."""
        if not self._rank1:
            return self._cholVecs
        out = []
        for q, d in zip(self._qSVec, self._qSDiag):
            S = q @ q.transpose(1, 2) + torch.diag_embed(d[:, :, 0] ** 2)
            L = torch.linalg.cholesky(S)
            M = L.shape[1]
            idx = torch.tril_indices(M, M, device=L.device)
            out.append(L[:, idx[0], idx[1]].unsqueeze(-1).contiguous())
        return out

    def getSVEmbeddingParams(self):
        return [self._C, self._d]

    def getKernels(self):
        return self._kernels

    def getKernelsParams(self):
        return [k.getParams() for k in self._kernels]

    def getIndPointsLocs(self):
        return self._Z

    # ------------------------------------------------------------------------------ engine
    def _theta(self):
        return [k.getParams() for k in self._kernels]

    def _ensure_engine(self):
        if self._kernels is None or self._quad is None or self._spikes is None or self._reg is None:
            raise RuntimeError("kernels, measurements, initial params, quadrature and the regularisation "
                               "parameter must be set before building kernel matrices")
        R = self._Z[0].shape[0]
        Ms = tuple(int(z.shape[1]) for z in self._Z)
        Q = int(self._quad[0].shape[1])
        key = (R, self._N, Q, Ms, tuple(k._engine_type for k in self._kernels),
               tuple(float(k._scale) for k in self._kernels),
               tuple(float(k._lengthscaleScale) for k in self._kernels),
               tuple(float(getattr(k, "_periodScale", 1.0)) for k in self._kernels), self._data_version,
               getattr(self, "_precision", "fp64"))
        if self._engine is not None and key == self._engine_key:
            return
        if self._R_data != R or self._quad[0].shape[0] != R:
            raise ValueError("number of trials differs between measurements, quadrature and parameters")
        if self._engine is not None:
            self._engine.close()
        eng = Engine(R, self._N, Q, Ms, key[4], key[5], key[6], key[7], device=self._device.index,
                     precision=getattr(self, "_precision", "fp64"))
        eng.set_reg(self._reg)
        eng.set_quad(self._quad[0], self._quad[1])
        off, t, n = self._spikes
        dev = self._device
        eng.set_spikes(off, torch.from_numpy(t).to(dev), torch.from_numpy(n).to(dev))
        self._engine, self._engine_key = eng, key
        self._stats_token = None

    def _sync(self, agree=False):
        """Wait for the engine's stream and surface its status word.  agree=True (after a factorisation): with trials
        sharded over ranks the inducing points are per trial, so a Cholesky can fail on ONE rank only; the ranks
        exchange their status (one small max-reduction) and raise together -- otherwise the healthy ranks would wait in
        the next collective for a rank that has already left the EM step through the exception."""
        err = None
        self._status_pending = False
        try:
            self._engine.sync()
        except SVGPFAEngineError as e:
            if e.code != _lib.ERR_NOT_PD:
                raise
            err = str(e)
        bad = 1 if err is not None else 0
        if agree and self._agree is not None:
            bad = self._agree(bad)
        if bad:
            # the reference raises from torch.linalg.cholesky (utils/miscUtils.py:215)
            raise torch.linalg.LinAlgError("linalg.cholesky: " + (
                err or "Kzz + eps*I is not positive definite on another rank (trial-sharded run)"))

    def _check_live(self, tensors, what):
        for t in tensors:
            if not (t.is_cuda and t.dtype == torch.float64 and t.is_contiguous()):
                raise ValueError(f"{what} tensors must stay contiguous fp64 tensors on {self._device}")

    # ------------------------------------------------------------------------------ compute
    def buildKernelsMatrices(self, defer_status=False):
        """svLowerBound.py:77-78: Kzz + eps I, its Cholesky (raises LinAlgError if not PD).  Ktz blocks are
        recomputed on the fly by the engine from this snapshot of Z and the kernel parameters.
        defer_status (not part of the reference's signature; the EM loop's flat closures use it): only enqueue -- the
        status of the factorisation is surfaced by the next checkStatus() / any call that waits for the engine, and
        evaluations enqueued in between run on whatever the failed factorisation left (no fault: no address depends on
        it) and are discarded with the exception."""
        self._ensure_engine()
        theta = self._theta()
        self._check_live(self._Z + theta, "inducing point / kernel parameter")
        self._engine.build_kernels(self._Z, theta)
        if defer_status:
            self._status_pending = True
        else:
            self._sync(agree=True)
        self._stats_token = None

    def checkStatus(self):
        """Surface the status of a kernel rebuild enqueued with defer_status=True (raises LinAlgError like
        buildKernelsMatrices(); with trials sharded over ranks every rank must call it at the same point)."""
        if getattr(self, "_status_pending", False):
            self._sync(agree=True)

    def eval(self):
        """svLowerBound.py:47-54."""
        if self._engine is None:
            self.buildKernelsMatrices()
        chol = self._chol_vecs()
        self._check_live(self._mean + [c.detach() for c in chol] + [self._C, self._d], "variational / embedding")
        if getattr(self, "_link", None) is not None:
            return self._eval_gauss_hermite(chol)
        theEval = _ELBOFunction.apply(self, *self._mean, *chol, self._C, self._d,
                                      *self._theta(), *self._Z)
        # the reference warns here when the bound is infinite (svLowerBound.py:51-53); reading the value would stall the
        # host on every evaluation, so the warning is raised where the value reaches the host (_BoundValue)
        return theEval.as_subclass(_BoundValue)

    def _eval_gauss_hermite(self, chol):
        """ELBO with a caller-supplied link function g and Gauss-Hermite quadrature over the Gaussian posterior of the
        embedding h ~ N(m, v) (reference PointProcessELLQuad, expectedLogLikelihood.py:215-255):
            E[g(h)]     at the quadrature nodes = sum_l w_l g(m + sqrt(2 v) x_l)                     (:229-239)
            E[log g(h)] at the spikes           = sum_l w_l log g(m + (2 v) x_l)                     (:243-252)
        The spike term scales the Hermite abscissae by 2 v where the node term (and the Gauss-Hermite rule) has
        sqrt(2 v); that is what the reference computes and what its golden vector pins, so it is replicated here as is.
        Latent statistics and their backward are the engine's; embedding and link are torch on the same GPU."""
        if self._herm is None:
            raise RuntimeError("eLLCalculationParams must hold hermQuadPoints / hermQuadWeights for a non-exponential link")
        theta, C, d = self._theta(), self._C, self._d
        if self._reduce is not None:
            # trials sharded over ranks: the shared parameters' gradients are summed over ranks in the backward, the
            # bound (and its two terms) in the forward; per-trial parameters keep their local gradients
            theta = [_SharedParam.apply(self, t) for t in theta]
            C, d = _SharedParam.apply(self, C), _SharedParam.apply(self, d)
        muq, varq, mus, vars_, kl = _LatentStatsFunction.apply(self, *self._mean, *chol, *theta, *self._Z)
        self._stats_token = None
        ell = self._gauss_hermite_ell(muq, varq, mus, vars_, C, d)
        if self._reduce is not None:
            ell, kl = _SumOverRanks.apply(self, ell), _SumOverRanks.apply(self, kl)
        self._last_terms = ((ell - kl).detach(), ell.detach(), kl.detach())
        return (ell - kl).as_subclass(_BoundValue)

    def _gauss_hermite_ell(self, muq, varq, mus, vars_, C=None, d=None):
        """Point-process ELL with the Gauss-Hermite link (expectedLogLikelihood.py:215-255) from the latent statistics at
        the quadrature nodes (R x Q x K) and at the spikes (S x K); differentiable in C, d and in the statistics."""
        C = self._C if C is None else C
        d = (self._d if d is None else d).reshape(-1)
        x, w = self._herm
        link = self._link
        eMean = torch.matmul(muq, C.T) + d                       # svEmbedding.py:80-84
        eVar = torch.matmul(varq, C.T ** 2)
        sd = torch.sqrt(2.0 * eVar)
        eLink = torch.zeros_like(eMean)
        for l in range(x.numel()):                               # one Hermite node at a time: R x Q x N temporaries only
            eLink = eLink + w[l] * link(eMean + sd * x[l])
        term1 = torch.sum(self._quad[1].reshape(eMean.shape[0], -1, 1) * eLink)
        nid = torch.as_tensor(self._spikes[2], device=self._device).long()
        Cn = C[nid]                                              # svEmbedding.py:137-144
        mS = torch.sum(mus * Cn, dim=1) + d[nid]
        vS = torch.sum(vars_ * Cn ** 2, dim=1)
        eLog = torch.zeros_like(mS)
        for l in range(x.numel()):
            eLog = eLog + w[l] * torch.log(link(mS + 2.0 * vS * x[l]))
        return -term1 + torch.sum(eLog)

    # ------------------------------------------------------------------------------ flat fast path (EM loop)
    GROUPS = ("var", "emb", "theta", "Z")

    def _group_tensors(self, group):
        if group == "var":
            return None if self._rank1 else self._mean + self._cholVecs
        if group == "emb":
            return [self._C, self._d]
        if group == "theta":
            return self._theta()
        return self._Z

    def flatParams(self, group):
        """The ONE contiguous buffer behind a parameter group ("var": means + cholVecs, "emb": C + d, "theta": kernel
        parameters, "Z": inducing points), or None when the group is not laid out that way (rank-1 + diagonal q(u), or
        a caller has rebound one of the tensors, e.g. kernel.setParams(new_tensor))."""
        flat, tens = self._flat.get(group), self._group_tensors(group)
        if flat is None or tens is None or getattr(self, "_link", None) is not None:
            return None
        o = flat.data_ptr()
        for t in tens:
            if not (t.is_cuda and t.dtype == torch.float64 and t.is_contiguous() and t.data_ptr() == o):
                return None
            o += 8 * t.numel()
        return flat if o == flat.data_ptr() + 8 * flat.numel() else None

    def _grad_views(self, group):
        if group not in self._gflat:
            flat = torch.zeros_like(self._flat[group])
            tens = (self._mean + self._cholVecs) if group == "var" else self._Z
            views, o = [], 0
            for t in tens:
                views.append(flat[o:o + t.numel()].view(t.shape[0], -1))
                o += t.numel()
            self._gflat[group] = (flat, views)
        return self._gflat[group]

    def evalFlat(self, group, defer_status=False):
        """(value, gradient) of one EM closure without an autograd graph: value = the lower bound (group "emb": the
        expected log-likelihood with the latent statistics of the latest computeSVPosteriorOnLatentsStats() frozen,
        svEM.py:225-232), gradient = d value / d flatParams(group), a view of an engine-owned buffer that the NEXT
        evaluation overwrites (callers copy or negate it).  Groups "theta" and "Z" rebuild the kernel matrices first
        (svEM.py:238-241, :258-260).  Nothing here waits for the GPU except that rebuild's status check."""
        if self.flatParams(group) is None:
            raise ValueError(f"group {group!r} is not available as a flat buffer")
        if self._engine is None or group in ("theta", "Z"):
            # defer_status: the caller reads the value back anyway and calls checkStatus() behind that wait (ShardedLBFGS
            # after_sync): one synchronisation per closure instead of two, and no gap in the GPU's queue behind the rebuild
            self.buildKernelsMatrices(defer_status=defer_status and self._engine is not None)
        eng, K = self._engine, self._K
        if group == "emb":
            if self._stats_token is None:
                raise RuntimeError("computeSVPosteriorOnLatentsStats() must be called before evalFlat('emb')")
            out = eng.ell_given_stats(self._C, self._d, True)
            if self._reduce is not None:
                self._reduce(out)
            return out[0], out[1:]
        eng.set_embedding_token(self._embedding_token())
        mask = {"var": _lib.GRAD_VARIATIONAL, "theta": _lib.GRAD_KERNELS, "Z": _lib.GRAD_INDPOINTS}[group]
        grads = (None, None, None)
        if group == "var":
            flat, v = self._grad_views("var")
            grads = (v[:K], v[K:], None)
        elif group == "Z":
            flat, v = self._grad_views("Z")
            grads = (None, None, v)
        if not hasattr(self, "_shared_buf") or self._shared_buf.numel() != eng.shared_len:
            self._shared_buf = torch.empty(eng.shared_len, dtype=torch.float64, device=self._device)
        shared = self._shared_buf
        eng.elbo(self._mean, self._cholVecs, self._C, self._d, mask, out=shared, grads=grads)
        if self._reduce is not None:
            self._reduce(shared)
        self._last_terms = (shared[0], shared[1], shared[2])
        if group == "theta":
            flat = shared[3 + self._N * K + self._N:]
        return shared[0], flat

    def computeSVPosteriorOnLatentsStats(self):
        """svLowerBound.py:80-81.  Returns the reference's dict layout; the engine keeps the same numbers
        resident for evalELLSumAcrossTrialsAndNeurons."""
        if self._engine is None:
            self.buildKernelsMatrices()
        muq, varq, mus, vars_ = self._engine.latent_stats(self._mean, [c.detach() for c in self._chol_vecs()])
        self._sync()
        off = self._spikes[0]
        R = len(off) - 1
        stats = {"allTimes": (muq, varq),
                 "assocTimes": ([mus[off[r]:off[r + 1]] for r in range(R)],
                                [vars_[off[r]:off[r + 1]] for r in range(R)])}
        self._stats_token = stats
        return stats

    def freezeLatentsStats(self):
        """What the embedding M-step needs of computeSVPosteriorOnLatentsStats() -- the statistics at the quadrature nodes
        and the per-neuron sums of the posterior means at the spikes, resident in HBM -- without the mean / variance at
        every spike (extension; svEM.py:225-232 passes the statistics back untouched).  Returns the token that
        evalELLSumAcrossTrialsAndNeurons(svPosteriorOnLatentsStats=...) and evalFlat("emb") accept."""
        if getattr(self, "_link", None) is not None:      # the Gauss-Hermite link needs mean and variance at every spike
            return self.computeSVPosteriorOnLatentsStats()
        if self._engine is None:
            self.buildKernelsMatrices()
        self._engine.freeze_stats(self._mean, [c.detach() for c in self._chol_vecs()], self._C.detach())
        self._stats_token = {"frozen": True}        # nothing is factorised here: no status to wait for
        return self._stats_token

    def evalELLSumAcrossTrialsAndNeurons(self, svPosteriorOnLatentsStats=None):
        """svLowerBound.py:72-75 -> expectedLogLikelihood.py:107-135."""
        if svPosteriorOnLatentsStats is None:
            self.computeSVPosteriorOnLatentsStats()
        elif svPosteriorOnLatentsStats is not self._stats_token:
            raise NotImplementedError("svPosteriorOnLatentsStats must be the object returned by the latest "
                                      "computeSVPosteriorOnLatentsStats() call (the stats stay resident in HBM)")
        if getattr(self, "_link", None) is not None:      # Gauss-Hermite link: the same torch expression as in eval()
            st = self._stats_token
            muq, varq = st["allTimes"]
            C, d = self._C, self._d
            if self._reduce is not None:                  # trials sharded: C, d gradients and the value summed over ranks
                C, d = _SharedParam.apply(self, C), _SharedParam.apply(self, d)
            ell = self._gauss_hermite_ell(muq.detach(), varq.detach(), torch.cat(st["assocTimes"][0]).detach(),
                                          torch.cat(st["assocTimes"][1]).detach(), C, d)
            return _SumOverRanks.apply(self, ell) if self._reduce is not None else ell
        return _ELLGivenStatsFunction.apply(self, self._C, self._d)

    def predictLatents(self, times):
        """svPosteriorOnLatents.py:57-77: times n_trials x n_times x 1 -> (means, vars) n_trials x n_times x K."""
        if self._engine is None:
            self.buildKernelsMatrices()
        t = _to_dev(times, self._device)
        mu, var = self._engine.predict(self._mean, [c.detach() for c in self._chol_vecs()],
                                       t.reshape(t.shape[0], -1).contiguous())
        self._sync()
        self._stats_token = None
        return mu, var

    def predictEmbedding(self, times):
        """svEmbedding.py:80-92."""
        mu, var = self.predictLatents(times)
        Ct = self._C.detach().T
        return torch.matmul(mu, Ct) + self._d.detach().reshape(1, 1, -1), torch.matmul(var, Ct ** 2)

    def computeEmbeddingMeansAndVarsAtTimes(self, times):
        return self.predictEmbedding(times)

    def computeExpectedPosteriorCIFs(self, times):
        """expectedLogLikelihood.py:62-73: list[R][N] of link(mean + var / 2) on the grid (link = exp unless the model was
        built with a link callable)."""
        eMean, eVar = self.predictEmbedding(times)
        link = getattr(self, "_link", None) or torch.exp
        cif = link(eMean + 0.5 * eVar)
        return [[cif[r, :, n] for n in range(cif.shape[2])] for r in range(cif.shape[0])]

    def sampleCIFs(self, times, nudget=1e-3):
        raise NotImplementedError("sampleCIFs (GP sampling for plots, svPosteriorOnLatents.py:123-183) is outside "
                                  "the accelerated ELBO path (SURVEY.md section 2, row 4)")

    # engine-backed KernelsMatricesStore getters (kernelsMatricesStore.py:119-120,176-181)
    def getKzz(self):
        return [self._engine.get_kzz(k) for k in range(self._K)]

    def getKzzChol(self):
        return [self._engine.get_kzz_chol(k) for k in range(self._K)]

    def getKtzAllTimes(self):
        return [self._engine.get_ktz_quad(k) for k in range(self._K)]

    def lastTerms(self):
        """(ELBO, ELL, KL) of the most recent eval() as 0-dim tensors."""
        return self._last_terms

    # ------------------------------------------------------------------------------ pickling
    def __getstate__(self):
        """svEM.py:89-92,175-181 pickles {"model": model}: drop the native handle, keep tensors and data."""
        st = dict(self.__dict__)
        st["_engine"] = None
        st["_engine_key"] = None
        st["_stats_token"] = None
        st["_reduce"] = None
        st["_agree"] = None
        st["_gflat"] = {}
        st["_flat"] = {}
        st.pop("_shared_buf", None)
        return st

    def __setstate__(self, st):
        self.__dict__.update(st)
        self._repack()

    def _repack(self):
        """Lay the parameter groups out contiguously again (pickling stores every tensor on its own)."""
        if getattr(self, "_C", None) is None:
            return
        dev, K = self._device, self._K
        self._flat = {}
        if not self._rank1:
            self._flat["var"], v = _pack_group(self._mean + self._cholVecs, dev)
            self._mean, self._cholVecs = v[:K], v[K:]
        self._flat["Z"], self._Z = _pack_group(self._Z, dev)
        self._flat["theta"], theta = _pack_group(self._theta(), dev)
        for k, kern in enumerate(self._kernels):
            kern.setParams(theta[k])
        self._flat["emb"], (self._C, self._d) = _pack_group([self._C, self._d], dev)
        self._token_salt += 1
