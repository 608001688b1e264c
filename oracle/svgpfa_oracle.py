"""CPU oracle for the svGPFA ELBO hot path -- TEST INFRASTRUCTURE, NOT PRODUCT.

This file restates, in plain PyTorch-CPU fp64 with autograd, the algorithm of the
reference's sparse-variational lower bound (point-process likelihood, exponential
link, linear embedding, Cholesky Kzz inverse, Cholesky q(u) covariance).  It keeps
the reference's UNFUSED formulation (materialised Ktz blocks, two Cholesky solves
per variance, slogdet-based KL, autograd gradients) on purpose: it is the checker
for the fused HIP engine, and the `cpu_baseline` leg of bench.py.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it.
The product path (svgpfa_amd/) never does.

Parity pin: tests/golden/*.npz were produced by importing the reference itself
(tests/golden/make_golden.py) and tests/test_oracle_golden.py checks this oracle
against them and against the MATLAB known answers the reference's tests hold
(ELBO -6037.49359022, ELL -5499.83650875, KL 537.65708147).

Reference citations are relative to /root/reference/src/svGPFA/.

A "problem" is a plain dict:
    kernel_types : list[K] of "exponentialQuadratic" | "periodic"
    scales, lengthscale_scales, period_scales : list[K] of float (default 1.0)
    mean[k]      : R x M_k x 1        cholVecs[k] : R x P_k x 1
    Z[k]         : R x M_k x 1        theta[k]    : (1,) or (2,)
    C            : N x K              d           : (N,) or (N,1)
    spike_times[r] : (S_r,)   spike_neuron[r] : (S_r,) int64
    quad_points  : R x Q x 1          quad_weights : R x Q x 1
    reg          : float  (epsilon added to the Kzz diagonal)
"""
import math

import torch

RBF = "exponentialQuadratic"
PERIODIC = "periodic"


# --------------------------------------------------------------------------- kernels
def kernel_matrix(ktype, theta, X1, X2, scale=1.0, lengthscale_scale=1.0,
                  period_scale=1.0):
    """stats/kernels.py:33-46 (RBF) and :73-85 (periodic); 3-D batched or 2-D."""
    if X1.ndim == 3:
        diff = X1 - X2.transpose(1, 2)
    else:
        diff = X1.reshape(-1, 1) - X2.reshape(1, -1)
    ls = theta[0] / lengthscale_scale
    if ktype == RBF:
        return scale**2 * torch.exp(-0.5 * diff**2 / ls**2)
    if ktype == PERIODIC:
        period = theta[1] / period_scale
        rr = math.pi * diff / period
        return scale**2 * torch.exp(-2.0 * torch.sin(rr)**2 / ls**2)
    raise ValueError("invalid kernel type %r" % (ktype,))


def kernel_diag(scale, X):
    """stats/kernels.py:48-51, :87-90 -- the diagonal is scale**2 for both."""
    return scale**2 * torch.ones(X.shape, dtype=X.dtype)


def _kargs(p, k):
    return dict(scale=p.get("scales", [1.0] * len(p["Z"]))[k],
                lengthscale_scale=p.get("lengthscale_scales", [1.0] * len(p["Z"]))[k],
                period_scale=p.get("period_scales", [1.0] * len(p["Z"]))[k])


def _scale(p, k):
    return p.get("scales", [1.0] * len(p["Z"]))[k]


# ------------------------------------------------------------- kernels matrices store
def chol3d(Kmat):
    """utils/miscUtils.py:209-216 -- per-trial torch.linalg.cholesky (raises if not PD)."""
    return torch.stack([torch.linalg.cholesky(Kmat[r]) for r in range(Kmat.shape[0])])


def build_kzz(p):
    """stats/kernelsMatricesStore.py:107-117 -- Kzz[k] = kappa(Z,Z) + eps*I and its Cholesky."""
    Kzz, Kzz_chol = [], []
    for k, Z in enumerate(p["Z"]):
        Kk = kernel_matrix(p["kernel_types"][k], p["theta"][k], Z, Z, **_kargs(p, k))
        Kk = Kk + p["reg"] * torch.eye(Z.shape[1], dtype=Z.dtype)
        Kzz.append(Kk)
        Kzz_chol.append(chol3d(Kk))
    return Kzz, Kzz_chol


def build_ktz_quad(p):
    """stats/kernelsMatricesStore.py:186-195 -- Ktz[k] R x Q x M_k, KttDiag R x Q x K."""
    t = p["quad_points"]
    Ktz = [kernel_matrix(p["kernel_types"][k], p["theta"][k], t, p["Z"][k], **_kargs(p, k))
           for k in range(len(p["Z"]))]
    KttDiag = torch.stack([kernel_diag(_scale(p, k), t).squeeze(-1)
                           for k in range(len(p["Z"]))], dim=2)
    return Ktz, KttDiag


def build_ktz_spikes(p):
    """stats/kernelsMatricesStore.py:208-221 -- nested lists Ktz[k][r] S_r x M_k."""
    K, R = len(p["Z"]), p["Z"][0].shape[0]
    Ktz = [[kernel_matrix(p["kernel_types"][k], p["theta"][k], p["spike_times"][r],
                          p["Z"][k][r], **_kargs(p, k)) for r in range(R)] for k in range(K)]
    KttDiag = [[kernel_diag(_scale(p, k), p["spike_times"][r]) for r in range(R)]
               for k in range(K)]
    return Ktz, KttDiag


# ------------------------------------------------------------------------ q(u) = N(m, S)
def chol_from_vec(vec, M):
    """utils/miscUtils.py:135-139 -- row-major torch.tril_indices fill."""
    L = torch.zeros((M, M), dtype=torch.double)
    idx = torch.tril_indices(M, M)
    L[idx[0], idx[1]] = vec
    return L


def build_covs_from_chol_vecs(cholVecs):
    """utils/miscUtils.py:141-155 -- S[k][r] = L L^T."""
    covs = []
    for cv in cholVecs:
        P = cv.shape[1]
        M = int((-1 + math.sqrt(1 + 8 * P)) / 2)
        rows = []
        for r in range(cv.shape[0]):
            L = chol_from_vec(cv[r, :, 0], M)
            rows.append(L @ L.T)
        covs.append(torch.stack(rows))
    return covs


# --------------------------------------------------------------- posterior on latents
def latents_stats_quad(p, Kzz, Kzz_chol, Ktz, KttDiag, qSigma):
    """stats/svPosteriorOnLatents.py:185-216 -- qKMu, qKVar R x Q x K."""
    mus, vrs = [], []
    for k in range(len(p["Z"])):
        Ak = torch.cholesky_solve(p["mean"][k], Kzz_chol[k])
        mus.append(torch.matmul(Ktz[k], Ak).squeeze(-1))
        Bkf = torch.cholesky_solve(Ktz[k].transpose(1, 2), Kzz_chol[k])
        mm1f = torch.matmul(qSigma[k] - Kzz[k], Bkf)
        vrs.append(KttDiag[:, :, k] + torch.sum(Bkf * mm1f, dim=1))
    return torch.stack(mus, dim=2), torch.stack(vrs, dim=2)


def latents_stats_spikes(p, Kzz, Kzz_chol, Ktz, KttDiag, qSigma):
    """stats/svPosteriorOnLatents.py:265-300 -- lists qKMu[r], qKVar[r] S_r x K."""
    K, R = len(p["Z"]), p["Z"][0].shape[0]
    Ak = [torch.cholesky_solve(p["mean"][k], Kzz_chol[k]) for k in range(K)]
    qKMu, qKVar = [], []
    for r in range(R):
        mu_r, var_r = [], []
        for k in range(K):
            mu_r.append(torch.mm(Ktz[k][r], Ak[k][r]).squeeze(-1))
            Bfk = torch.cholesky_solve(Ktz[k][r].T, Kzz_chol[k][r])
            mm1f = torch.matmul(qSigma[k][r] - Kzz[k][r], Bfk)
            var_r.append(KttDiag[k][r] + torch.sum(Bfk * mm1f, dim=0))
        S_r = p["spike_times"][r].shape[0]
        qKMu.append(torch.stack(mu_r, dim=1) if K else torch.empty((S_r, 0)))
        qKVar.append(torch.stack(var_r, dim=1))
    return qKMu, qKVar


def compute_latents_stats(p):
    """stats/expectedLogLikelihood.py:141-147 -- {"allTimes": (mu,var), "assocTimes": (mu,var)}."""
    Kzz, Kzz_chol = build_kzz(p)
    Ktz_q, Ktt_q = build_ktz_quad(p)
    Ktz_s, Ktt_s = build_ktz_spikes(p)
    qSigma = build_covs_from_chol_vecs(p["cholVecs"])
    return {"allTimes": latents_stats_quad(p, Kzz, Kzz_chol, Ktz_q, Ktt_q, qSigma),
            "assocTimes": latents_stats_spikes(p, Kzz, Kzz_chol, Ktz_s, Ktt_s, qSigma)}


# --------------------------------------------------------------------------- embedding
def embedding_quad(p, mu, var):
    """stats/svEmbedding.py:80-84."""
    C, d = p["C"], p["d"]
    mean = torch.matmul(mu, C.T) + d.reshape(1, 1, -1)
    return mean, torch.matmul(var, (C.T)**2)


def embedding_spikes(p, mu, var):
    """stats/svEmbedding.py:137-144 -- per-spike gather of C rows / d entries."""
    C, d = p["C"], p["d"].reshape(-1)
    means, vrs = [], []
    for r in range(len(mu)):
        idx = p["spike_neuron"][r]
        means.append(torch.sum(mu[r] * C[idx, :], dim=1) + d[idx])
        vrs.append(torch.sum(var[r] * C[idx, :]**2, dim=1))
    return means, vrs


# ------------------------------------------------------------------------------- terms
def ell_given_stats(p, stats):
    """stats/expectedLogLikelihood.py:107-135 with the exp link of :205-213."""
    eMean, eVar = embedding_quad(p, *stats["allTimes"])
    sMean, _ = embedding_spikes(p, *stats["assocTimes"])
    eLink = torch.exp(eMean + 0.5 * eVar)
    term1 = torch.sum(torch.matmul(p["quad_weights"].transpose(1, 2), eLink))
    term2 = torch.sum(torch.cat(sMean)) if len(sMean) else torch.zeros((), dtype=torch.double)
    return -term1 + term2


def kl_divergence(p, Kzz, Kzz_chol, qSigma):
    """stats/klDivergence.py:18-44 -- slogdet (LU) of Kzz and S, trace of a Cholesky solve."""
    total = torch.zeros((), dtype=torch.double)
    for k in range(len(p["Z"])):
        m = p["mean"][k]
        ESS = qSigma[k] + torch.matmul(m, m.permute(0, 2, 1))
        for r in range(m.shape[0]):
            _, ldK = torch.linalg.slogdet(Kzz[k][r])
            _, ldS = torch.linalg.slogdet(qSigma[k][r])
            tr = torch.trace(torch.cholesky_solve(ESS[r], Kzz_chol[k][r]))
            total = total + 0.5 * (tr + ldK - ldS - ESS.shape[1])
    return total


def elbo_terms(p):
    """stats/svLowerBound.py:47-54 -- returns (elbo, ell, kl) as 0-dim tensors."""
    Kzz, Kzz_chol = build_kzz(p)
    Ktz_q, Ktt_q = build_ktz_quad(p)
    Ktz_s, Ktt_s = build_ktz_spikes(p)
    # the reference calls buildCov three times per eval; once is the same maths
    qSigma = build_covs_from_chol_vecs(p["cholVecs"])
    stats = {"allTimes": latents_stats_quad(p, Kzz, Kzz_chol, Ktz_q, Ktt_q, qSigma),
             "assocTimes": latents_stats_spikes(p, Kzz, Kzz_chol, Ktz_s, Ktt_s, qSigma)}
    ell = ell_given_stats(p, stats)
    kl = kl_divergence(p, Kzz, Kzz_chol, qSigma)
    return ell - kl, ell, kl


# --------------------------------------------------------------------------- gradients
GRAD_VARIATIONAL = 1   # mean + cholVecs   (stats/svEM.py:218-223)
GRAD_EMBEDDING = 2     # C + d             (stats/svEM.py:225-232)
GRAD_KERNELS = 4       # theta_k           (stats/svEM.py:234-254)
GRAD_INDPOINTS = 8     # Z_k               (stats/svEM.py:256-264)
GRAD_ALL = 15


def _leaves(p, mask):
    leaves = []
    if mask & GRAD_VARIATIONAL:
        leaves += list(p["mean"]) + list(p["cholVecs"])
    if mask & GRAD_EMBEDDING:
        leaves += [p["C"], p["d"]]
    if mask & GRAD_KERNELS:
        leaves += list(p["theta"])
    if mask & GRAD_INDPOINTS:
        leaves += list(p["Z"])
    return leaves


def elbo_and_grads(p, mask=GRAD_ALL):
    """ELBO and autograd gradients, as stats/svEM.py:274-288 obtains them
    (the reference minimises -ELBO; gradients returned here are of +ELBO)."""
    leaves = _leaves(p, mask)
    for t in leaves:
        t.requires_grad_(True)
        t.grad = None
    try:
        elbo, ell, kl = elbo_terms(p)
        grads = torch.autograd.grad(elbo, leaves, allow_unused=True) if leaves else []
    finally:
        for t in leaves:
            t.requires_grad_(False)
    K = len(p["Z"])
    out = {"elbo": elbo.detach(), "ell": ell.detach(), "kl": kl.detach()}
    it = iter([g if g is not None else torch.zeros_like(t) for g, t in zip(grads, leaves)])
    if mask & GRAD_VARIATIONAL:
        out["g_mean"] = [next(it) for _ in range(K)]
        out["g_cholVecs"] = [next(it) for _ in range(K)]
    if mask & GRAD_EMBEDDING:
        out["g_C"], out["g_d"] = next(it), next(it)
    if mask & GRAD_KERNELS:
        out["g_theta"] = [next(it) for _ in range(K)]
    if mask & GRAD_INDPOINTS:
        out["g_Z"] = [next(it) for _ in range(K)]
    return out


def ell_given_stats_and_grads(p, stats):
    """The embedding M-step closure (stats/svEM.py:225-232): ELL only, stats frozen."""
    leaves = [p["C"], p["d"]]
    for t in leaves:
        t.requires_grad_(True)
    try:
        ell = ell_given_stats(p, stats)
        gC, gd = torch.autograd.grad(ell, leaves)
    finally:
        for t in leaves:
            t.requires_grad_(False)
    return ell.detach(), gC, gd


def predict_latents(p, times):
    """stats/svPosteriorOnLatents.py:57-77 -- posterior mean/var on an R x T x 1 grid."""
    Kzz, Kzz_chol = build_kzz(p)
    q = dict(p)
    q["quad_points"] = times
    Ktz, KttDiag = build_ktz_quad(q)
    qSigma = build_covs_from_chol_vecs(p["cholVecs"])
    return latents_stats_quad(p, Kzz, Kzz_chol, Ktz, KttDiag, qSigma)


def from_numpy_problem(prob):
    """numpy problem (svgpfa_amd.utils.synthetic layout) -> the dict of torch CPU fp64 tensors used here."""
    import numpy as np
    R = prob["R"]
    off = prob["spike_offsets"]
    t = torch.from_numpy(np.ascontiguousarray(prob["spike_times"])).double()
    n = torch.from_numpy(np.ascontiguousarray(prob["spike_neuron"])).long()
    return dict(
        kernel_types=list(prob["kernel_types"]), reg=float(prob["reg"]),
        scales=list(prob.get("scales", [1.0] * prob["K"])),
        lengthscale_scales=list(prob.get("lengthscale_scales", [1.0] * prob["K"])),
        period_scales=list(prob.get("period_scales", [1.0] * prob["K"])),
        mean=[torch.from_numpy(a.copy()).double().unsqueeze(-1) for a in prob["mean"]],
        cholVecs=[torch.from_numpy(a.copy()).double().unsqueeze(-1) for a in prob["cholVecs"]],
        Z=[torch.from_numpy(a.copy()).double().unsqueeze(-1) for a in prob["Z"]],
        theta=[torch.from_numpy(a.copy()).double() for a in prob["theta"]],
        C=torch.from_numpy(prob["C"].copy()).double(),
        d=torch.from_numpy(prob["d"].copy()).double(),
        spike_times=[t[off[r]:off[r + 1]] for r in range(R)],
        spike_neuron=[n[off[r]:off[r + 1]] for r in range(R)],
        quad_points=torch.from_numpy(prob["quad_points"].copy()).double().unsqueeze(-1),
        quad_weights=torch.from_numpy(prob["quad_weights"].copy()).double().unsqueeze(-1),
    )
