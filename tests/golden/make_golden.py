#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE implementation itself.

Runs only in the build container (needs /root/reference, which never travels to the
GPU box).  Usage:   python tests/golden/make_golden.py

The reference's only missing dependency on this path is the un-vendored, unpinned
`gcnu_common` package (setup.cfg:21; imported at utils/miscUtils.py:13-14).  Its one
function used here, leggaussVarLimits, is numpy Gauss-Legendre mapped to [a,b]
(the reference's own .mat quadrature nodes equal it to 1e-14), so an
in-memory stand-in module is registered before the import; nothing is written to disk
and no reference source is copied.

What is written (all float64 numpy arrays, inputs AND reference outputs):
  matlab_estep.npz  - the 5x50x3x10 geometry of the reference's MATLAB fixtures
                      (Estep_Objective_PointProcess_svGPFA.mat + YNonStacked.mat):
                      inputs, MATLAB known answers (obj/Elik/KLd), reference ELBO/ELL/KL,
                      every intermediate (Kzz, chol, Ktz, latent and embedding stats)
                      and autograd gradients for the four parameter groups.
  ragged_small.npz  - tiny ragged case: M=[4,6,5], mixed kernels, non-unit kernel
                      scales, a silent neuron, an empty trial.
  cfg2.npz          - BASELINE config 2 (50x50x5x16 RBF) on SURVEY 8(d) inputs:
                      scalars + full gradients.
  estep_embedding.npz - embedding M-step closure (ELL with frozen latent stats) on the
                      ragged case.
  em_small.npz      - the reference's own SVEM_PyTorch.maximize (ECM, 2 iterations) on a small
                      seeded problem: lowerBoundHist and final shared parameters.
  rank1_small.npz   - the rank-1 + diagonal representation of the variational covariances
                      (indPointsCovRep = indPointsCovRank1PlusDiag, stats/svPosteriorOnIndPoints.py:86-119)
                      on the ragged case: ELBO / ELL / KL and gradients wrt mean, qSVec, qSDiag.
  pinv_small.npz    - the pseudo-inverse Kzz model (kernelMatrixInvMethod = kernelMatrixInvPInv,
                      stats/kernelsMatricesStore.py:146-165) on the ragged case: ELBO / ELL / KL and all gradients.
  gh_small.npz      - the Gauss-Hermite (non-exponential link) point-process ELL of stats/expectedLogLikelihood.py:215-255
                      with a softplus link on the ragged case: ELBO / ELL / KL and all gradients.
  cfg3_shape.npz    - the north-star geometry (BASELINE config 3: 200 neurons, 10 latents RBF + periodic, 32 inducing
                      points, Q = 200, ~4000 spikes per trial) on the first 6 trials of the SURVEY 8(d) inputs:
                      scalars + gradients of all four groups (inputs are regenerated from the seed in the tests).
  cfg1_bundled.npz  - BASELINE config 1: the bundled simulated data
                      (examples/data/32451751_simRes.pickle: spike times only, as CSR) with the
                      parameters of examples/params/00000545_estimation_metaData.ini, and the
                      lowerBoundHist of the reference's EM loop for em_max_iter = 2.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF_SRC = "/root/reference/src"
REF_DATA = os.path.join(REF_SRC, "svGPFA/stats/tests/data")
sys.path.insert(0, REPO)


def _install_gcnu_common_standin():
    def mod(name):
        m = types.ModuleType(name)
        sys.modules[name] = m
        return m
    root = mod("gcnu_common")
    nm = mod("gcnu_common.numerical_methods")
    nmu = mod("gcnu_common.numerical_methods.utils")
    st = mod("gcnu_common.stats")
    gp = mod("gcnu_common.stats.gaussianProcesses")
    ev = mod("gcnu_common.stats.gaussianProcesses.eval")
    root.numerical_methods, nm.utils = nm, nmu
    root.stats, st.gaussianProcesses, gp.eval = st, gp, ev

    def leggaussVarLimits(n, a, b):
        x, w = np.polynomial.legendre.leggauss(n)
        return (torch.from_numpy((a * (1 - x) + b * (1 + x)) / 2),
                torch.from_numpy((b - a) / 2 * w))
    nmu.leggaussVarLimits = leggaussVarLimits


_install_gcnu_common_standin()
sys.path.insert(0, REF_SRC)
import svGPFA.stats.kernels  # noqa: E402
import svGPFA.stats.svGPFAModelFactory  # noqa: E402
import svGPFA.utils.miscUtils  # noqa: E402

from svgpfa_amd.utils import synthetic  # noqa: E402


# ------------------------------------------------------------------ reference driving
def ref_model_from_problem(prob):
    """numpy problem (svgpfa_amd.utils.synthetic layout) -> reference model + live tensors."""
    K, R, N = prob["K"], prob["R"], prob["N"]
    scales = prob.get("scales", [1.0] * K)
    lss = prob.get("lengthscale_scales", [1.0] * K)
    pss = prob.get("period_scales", [1.0] * K)
    kernels = []
    for k in range(K):
        if prob["kernel_types"][k] == synthetic.RBF:
            kernels.append(svGPFA.stats.kernels.ExponentialQuadraticKernel(
                scale=scales[k], lengthscaleScale=lss[k]))
        else:
            kernels.append(svGPFA.stats.kernels.PeriodicKernel(
                scale=scales[k], lengthscaleScale=lss[k], periodScale=pss[k]))
    t = lambda a: torch.from_numpy(np.array(a, dtype=np.float64, copy=True, order="C"))
    params = dict(
        mean=[t(a).unsqueeze(-1) for a in prob["mean"]],
        cholVecs=[t(a).unsqueeze(-1) for a in prob["cholVecs"]],
        Z=[t(a).unsqueeze(-1) for a in prob["Z"]],
        theta=[t(a) for a in prob["theta"]],
        C=t(prob["C"]), d=t(prob["d"]))
    off, times, neuron = prob["spike_offsets"], prob["spike_times"], prob["spike_neuron"]
    measurements = []
    for r in range(R):
        tr_t, tr_n = times[off[r]:off[r + 1]], neuron[off[r]:off[r + 1]]
        measurements.append([t(tr_t[tr_n == n]) for n in range(N)])
    initial_params = {
        "posterior_on_latents": {
            "posterior_on_ind_points": {"mean": params["mean"], "cholVecs": params["cholVecs"]},
            "kernels_matrices_store": {"kernels_params0": params["theta"],
                                       "inducing_points_locs0": params["Z"]}},
        "embedding": {"C0": params["C"], "d0": params["d"]}}
    ell_params = {"leg_quad_points": t(prob["quad_points"]).unsqueeze(-1),
                  "leg_quad_weights": t(prob["quad_weights"]).unsqueeze(-1)}
    model = svGPFA.stats.svGPFAModelFactory.SVGPFAModelFactory.buildModelPyTorch(kernels=kernels)
    model.setParamsAndData(measurements=measurements, initial_params=initial_params,
                           eLLCalculationParams=ell_params, priorCovRegParam=prob["reg"])
    return model, params


def ref_outputs(model, params, intermediates=True):
    """Run the reference: ELBO/ELL/KL, intermediates, autograd gradients of +ELBO."""
    out = {}
    K = len(params["Z"])
    leaves = (params["mean"] + params["cholVecs"] + [params["C"], params["d"]]
              + params["theta"] + params["Z"])
    for x in leaves:
        x.requires_grad_(True)
    model.buildKernelsMatrices()
    ell = model._eLL.evalSumAcrossTrialsAndNeurons()
    kl = model._klDiv.evalSumAcrossLatentsAndTrials()
    elbo = model.eval()
    grads = torch.autograd.grad(elbo, leaves, allow_unused=True)
    for x in leaves:
        x.requires_grad_(False)
    grads = [g if g is not None else torch.zeros_like(x) for g, x in zip(grads, leaves)]
    out["ref_elbo"], out["ref_ell"], out["ref_kl"] = (float(elbo), float(ell), float(kl))
    names = ([f"g_mean_{k}" for k in range(K)] + [f"g_cholVecs_{k}" for k in range(K)]
             + ["g_C", "g_d"] + [f"g_theta_{k}" for k in range(K)] + [f"g_Z_{k}" for k in range(K)])
    for nme, g in zip(names, grads):
        out[nme] = g.detach().numpy().reshape(g.shape[:-1] if (g.ndim == 3) else g.shape)
    if intermediates:
        model.buildKernelsMatrices()
        qK = model._eLL._svEmbeddingAllTimes._svPosteriorOnLatents
        qKs = model._eLL._svEmbeddingAssocTimes._svPosteriorOnLatents
        kms = qK._indPointsLocsKMS
        for k in range(K):
            out[f"Kzz_{k}"] = kms.getKzz()[k].detach().numpy()
            out[f"KzzChol_{k}"] = kms._Kzz_inv[k].detach().numpy()
            out[f"KtzQuad_{k}"] = qK._indPointsLocsAndTimesKMS.getKtz()[k].detach().numpy()
            ktz_s = qKs._indPointsLocsAndTimesKMS.getKtz()[k]
            out[f"KtzSpikes_{k}"] = np.concatenate(
                [x.detach().numpy().reshape(-1, x.shape[-1]) for x in ktz_s], axis=0)
        out["KttDiagQuad"] = qK._indPointsLocsAndTimesKMS.getKttDiag().detach().numpy()
        stats = model.computeSVPosteriorOnLatentsStats()
        out["muQuad"] = stats["allTimes"][0].detach().numpy()
        out["varQuad"] = stats["allTimes"][1].detach().numpy()
        out["muSpikes"] = np.concatenate([x.detach().numpy() for x in stats["assocTimes"][0]], 0)
        out["varSpikes"] = np.concatenate([x.detach().numpy() for x in stats["assocTimes"][1]], 0)
        eM, eV = model._eLL._svEmbeddingAllTimes.computeMeansAndVars()
        out["embMeanQuad"], out["embVarQuad"] = eM.detach().numpy(), eV.detach().numpy()
        sM, sV = model._eLL._svEmbeddingAssocTimes.computeMeansAndVars()
        out["embMeanSpikes"] = np.concatenate([x.detach().numpy().reshape(-1) for x in sM])
        out["embVarSpikes"] = np.concatenate([x.detach().numpy().reshape(-1) for x in sV])
        qS = qK._svPosteriorOnIndPoints.buildCov()
        for k in range(K):
            out[f"qSigma_{k}"] = qS[k].detach().numpy()
    return out


def problem_inputs(prob):
    """Flatten a numpy problem into npz-storable arrays."""
    K = prob["K"]
    d = dict(R=prob["R"], N=prob["N"], K=K, Q=prob["Q"], Ms=np.array(prob["Ms"]),
             reg=prob["reg"],
             kernel_types=np.array([0 if kt == synthetic.RBF else 1 for kt in prob["kernel_types"]]),
             scales=np.array(prob.get("scales", [1.0] * K), dtype=np.float64),
             lengthscale_scales=np.array(prob.get("lengthscale_scales", [1.0] * K), dtype=np.float64),
             period_scales=np.array(prob.get("period_scales", [1.0] * K), dtype=np.float64),
             spike_offsets=prob["spike_offsets"], spike_times=prob["spike_times"],
             spike_neuron=prob["spike_neuron"], quad_points=prob["quad_points"],
             quad_weights=prob["quad_weights"], C=prob["C"], d=prob["d"])
    for k in range(K):
        d[f"Z_{k}"], d[f"mean_{k}"] = prob["Z"][k], prob["mean"][k]
        d[f"cholVecs_{k}"], d[f"theta_{k}"] = prob["cholVecs"][k], prob["theta"][k]
    return d


# ------------------------------------------------------------------------ case builders
def matlab_problem():
    """Inputs of the reference's own ELBO test (stats/tests/test_svLowerBound.py:18-106)."""
    from scipy.io import loadmat
    mat = loadmat(os.path.join(REF_DATA, "Estep_Objective_PointProcess_svGPFA.mat"))
    ymat = loadmat(os.path.join(REF_DATA, "YNonStacked.mat"))["YNonStacked"]
    K = len(mat["Z"])
    R = mat["Z"][0, 0].shape[2]
    perm = lambda a: torch.from_numpy(a).double().permute(2, 0, 1)
    qMu = [perm(mat["q_mu"][(0, k)]) for k in range(K)]
    qSVec = [perm(mat["q_sqrt"][(0, k)]) for k in range(K)]
    qSDiag = [perm(mat["q_diag"][(0, k)]) for k in range(K)]
    # MATLAB stores q(u) covariance as rank-1 + diag; the reference's tests convert it to
    # Cholesky vectors with utils/miscUtils.py:367-381 -- do the same with the reference code
    chol = svGPFA.utils.miscUtils.getSRQSigmaVec(qSVec=qSVec, qSDiag=qSDiag)
    Z = [perm(mat["Z"][(k, 0)]) for k in range(K)]
    N = ymat[0, 0].shape[0]
    times, neuron, offsets = [], [], [0]
    for r in range(R):
        for n in range(N):
            sp = np.asarray(ymat[r, 0][n, 0][:, 0], dtype=np.float64)
            times.append(sp)
            neuron.append(np.full(sp.shape, n, dtype=np.int32))
        offsets.append(sum(len(x) for x in times))
    ktypes, theta = [], []
    for k in range(K):
        name = str(mat["kernelNames"][0, k][0])
        hp = np.asarray(mat["hprs"][k, 0], dtype=np.float64).reshape(-1)
        if name == "PeriodicKernel":
            ktypes.append(synthetic.PERIODIC)
            theta.append(hp[:2].copy())
        else:
            ktypes.append(synthetic.RBF)
            theta.append(hp[:1].copy())
    prob = dict(R=R, N=N, K=K, Ms=[z.shape[1] for z in Z], Q=mat["ttQuad"].shape[0],
                kernel_types=ktypes, reg=1e-5,
                spike_offsets=np.array(offsets, dtype=np.int64),
                spike_times=np.concatenate(times), spike_neuron=np.concatenate(neuron),
                quad_points=np.ascontiguousarray(np.transpose(mat["ttQuad"], (2, 0, 1))[:, :, 0]),
                quad_weights=np.ascontiguousarray(np.transpose(mat["wwQuad"], (2, 0, 1))[:, :, 0]),
                Z=[z.numpy()[:, :, 0].copy() for z in Z],
                mean=[m.numpy()[:, :, 0].copy() for m in qMu],
                cholVecs=[c.numpy()[:, :, 0].copy() for c in chol],
                theta=theta, C=np.asarray(mat["C"], dtype=np.float64).copy(),
                d=np.asarray(mat["b"], dtype=np.float64).reshape(-1).copy())
    known = dict(matlab_obj=float(mat["obj"][0, 0]))
    for key in ("Elik", "KLd"):
        if key in mat:
            known["matlab_" + key] = float(np.asarray(mat[key]).reshape(-1)[0])
    return prob, known


def ragged_problem():
    prob = synthetic.make_problem(R=4, N=7, K=3, M=[4, 6, 5], kernels="mix", seed=77,
                                  n_quad=24, rate=6.0, reg=1e-3, jitter=1.0)
    prob["scales"] = [1.3, 0.8, 1.0]
    prob["lengthscale_scales"] = [2.0, 1.0, 0.5]
    prob["period_scales"] = [1.0, 1.5, 1.0]
    # kernel params are stored UNSCALED (kernels.py:37,75-76 divide by the *Scale attributes)
    prob["theta"][0] = prob["theta"][0] * 2.0
    prob["theta"][1] = prob["theta"][1] * np.array([1.0, 1.5])
    prob["theta"][2] = prob["theta"][2] * 0.5
    off, t, n = prob["spike_offsets"], prob["spike_times"], prob["spike_neuron"]
    # neuron 3 silent everywhere; trial 2 has no spikes at all
    keep = n != 3
    trial_of = np.repeat(np.arange(prob["R"]), np.diff(off))
    keep &= trial_of != 2
    prob["spike_times"], prob["spike_neuron"] = t[keep], n[keep]
    cnt = np.bincount(trial_of[keep], minlength=prob["R"])
    prob["spike_offsets"] = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int64)
    return prob


def main():
    torch.manual_seed(0)
    # ---- 1. MATLAB geometry
    prob, known = matlab_problem()
    model, params = ref_model_from_problem(prob)
    out = ref_outputs(model, params)
    print("matlab: ref elbo %.8f ell %.8f kl %.8f | matlab obj %.8f" %
          (out["ref_elbo"], out["ref_ell"], out["ref_kl"], known["matlab_obj"]))
    np.savez_compressed(os.path.join(HERE, "matlab_estep.npz"), **problem_inputs(prob), **known, **out)

    # ---- 2. ragged small (+ the embedding M-step closure on it)
    prob = ragged_problem()
    model, params = ref_model_from_problem(prob)
    out = ref_outputs(model, params)
    print("ragged: ref elbo %.10f ell %.10f kl %.10f" % (out["ref_elbo"], out["ref_ell"], out["ref_kl"]))
    np.savez_compressed(os.path.join(HERE, "ragged_small.npz"), **problem_inputs(prob), **out)
    stats = model.computeSVPosteriorOnLatentsStats()
    for x in (params["C"], params["d"]):
        x.requires_grad_(True)
    ell = model.evalELLSumAcrossTrialsAndNeurons(svPosteriorOnLatentsStats=stats)
    gC, gd = torch.autograd.grad(ell, [params["C"], params["d"]])
    np.savez_compressed(os.path.join(HERE, "estep_embedding.npz"), ref_ell=float(ell),
                        g_C=gC.numpy(), g_d=gd.numpy())
    # predictLatents on a 9-point grid (stats/svPosteriorOnLatents.py:57-77)
    grid = torch.linspace(0.05, 0.95, 9, dtype=torch.double).reshape(1, -1, 1).repeat(prob["R"], 1, 1)
    mu, var = model.predictLatents(times=grid)
    emu, evar = model.predictEmbedding(times=grid)
    np.savez_compressed(os.path.join(HERE, "ragged_predict.npz"), times=grid.numpy()[:, :, 0],
                        mu=mu.detach().numpy(), var=var.detach().numpy(),
                        emb_mean=emu.detach().numpy(), emb_var=evar.detach().numpy())

    # ---- 3. BASELINE config 2 on SURVEY 8(d) inputs
    prob = synthetic.make_config("cfg2")
    model, params = ref_model_from_problem(prob)
    out = ref_outputs(model, params, intermediates=False)
    print("cfg2: ref elbo %.8f ell %.8f kl %.8f" % (out["ref_elbo"], out["ref_ell"], out["ref_kl"]))
    # inputs are regenerated from the seed in tests; store only what pins them + outputs
    np.savez_compressed(os.path.join(HERE, "cfg2.npz"),
                        n_spikes=prob["spike_offsets"][-1],
                        spike_times_sum=prob["spike_times"].sum(),
                        C_sum=prob["C"].sum(), **out)


LBFGS = dict(max_iter=20, lr=1.0, tolerance_grad=1e-7, tolerance_change=1e-9, line_search_fn="strong_wolfe")


def optim_params(em_max_iter):
    op = {"em_max_iter": em_max_iter}
    for step in ("estep", "mstep_embedding", "mstep_kernels", "mstep_indpointslocs"):
        op[f"{step}_estimate"] = True
        op[f"{step}_optim_params"] = dict(LBFGS)
    return op


def run_reference_em(prob, em_max_iter):
    import io
    import svGPFA.stats.svEM
    model, params = ref_model_from_problem(prob)
    hist, _, info, _ = svGPFA.stats.svEM.SVEM_PyTorch().maximize(
        model=model, optim_params=optim_params(em_max_iter), method="ECM", out=io.StringIO())
    assert not hasattr(info, "_error"), info.message
    return hist, params


def em_small():
    prob = synthetic.make_problem(R=3, N=8, K=2, M=[5, 6], kernels="mix", seed=123, n_quad=40, rate=12.0,
                                  reg=1e-3, jitter=1.0)
    hist, params = run_reference_em(prob, 2)
    print("em_small lowerBoundHist", hist)
    np.savez_compressed(os.path.join(HERE, "em_small.npz"), **problem_inputs(prob), lowerBoundHist=np.array(hist),
                        final_C=params["C"].detach().numpy(), final_d=params["d"].detach().numpy(),
                        final_theta0=params["theta"][0].detach().numpy(),
                        final_theta1=params["theta"][1].detach().numpy())


def cfg1_bundled():
    import pickle
    with open("/root/reference/examples/data/32451751_simRes.pickle", "rb") as f:
        sim = pickle.load(f)
    spikes = sim["spikes"]
    R, N = len(spikes), len(spikes[0])
    times, neuron, off = [], [], [0]
    for r in range(R):
        for n in range(N):
            sp = np.asarray(spikes[r][n], dtype=np.float64).reshape(-1)   # float32 values, exact in float64
            times.append(sp)
            neuron.append(np.full(sp.shape, n, dtype=np.int32))
        off.append(sum(len(x) for x in times))
    pdir = "/root/reference/examples/params/"
    m0 = np.loadtxt(pdir + "uniform_0.00_1.00_len09.csv", delimiter=",").reshape(-1)
    S0 = np.loadtxt(pdir + "identity_scaled1e-2_09x09.csv", delimiter=",")
    C0 = np.loadtxt(pdir + "C_constant_1.00constant_100neurons_02latents.csv", delimiter=",").reshape(N, 2)
    d0 = np.loadtxt(pdir + "d_constant_0.00constant_100neurons.csv", delimiter=",").reshape(-1)
    K, M, Q = 2, 9, 200
    L0 = np.linalg.cholesky(S0)
    pts, wts = synthetic.leg_quad(Q, np.zeros(R), np.ones(R))
    prob = dict(R=R, N=N, K=K, Ms=[M, M], Q=Q, kernel_types=[synthetic.RBF, synthetic.RBF], reg=1e-3,
                spike_offsets=np.array(off, dtype=np.int64), spike_times=np.concatenate(times),
                spike_neuron=np.concatenate(neuron), quad_points=pts, quad_weights=wts,
                Z=[np.tile(np.linspace(0.0, 1.0, M)[None, :], (R, 1)) for _ in range(K)],
                mean=[np.tile(m0[None, :], (R, 1)) for _ in range(K)],
                cholVecs=[np.tile(synthetic.tril_vec(L0)[None, :], (R, 1)) for _ in range(K)],
                theta=[np.array([2.0]), np.array([1.0])], C=C0.copy(), d=d0.copy())
    hist, params = run_reference_em(prob, 2)
    print("cfg1 lowerBoundHist", hist)
    np.savez_compressed(os.path.join(HERE, "cfg1_bundled.npz"), **problem_inputs(prob),
                        lowerBoundHist=np.array(hist))


def cfg3_shape():
    prob = synthetic.make_config("cfg3", n_trials=6)
    model, params = ref_model_from_problem(prob)
    out = ref_outputs(model, params, intermediates=False)
    print("cfg3_shape: ref elbo %.8f ell %.8f kl %.8f" % (out["ref_elbo"], out["ref_ell"], out["ref_kl"]))
    np.savez_compressed(os.path.join(HERE, "cfg3_shape.npz"), n_trials=6, n_spikes=prob["spike_offsets"][-1],
                        spike_times_sum=prob["spike_times"].sum(), C_sum=prob["C"].sum(), **out)


def pinv_small():
    """Non-default Kzz inversion: kernelMatrixInvPInv (kernelsMatricesStore.py:146-165) on the ragged case."""
    prob = ragged_problem()
    F = svGPFA.stats.svGPFAModelFactory
    model0, params = ref_model_from_problem(prob)
    K, R, N = prob["K"], prob["R"], prob["N"]
    model = F.SVGPFAModelFactory.buildModelPyTorch(kernels=model0.getKernels(),
                                                   kernelMatrixInvMethod=F.kernelMatrixInvPInv)
    t = lambda a: torch.from_numpy(np.array(a, dtype=np.float64, copy=True, order="C"))
    off, times, neuron = prob["spike_offsets"], prob["spike_times"], prob["spike_neuron"]
    measurements = []
    for r in range(R):
        tr_t, tr_n = times[off[r]:off[r + 1]], neuron[off[r]:off[r + 1]]
        measurements.append([t(tr_t[tr_n == n]) for n in range(N)])
    initial_params = {
        "posterior_on_latents": {
            "posterior_on_ind_points": {"mean": params["mean"], "cholVecs": params["cholVecs"]},
            "kernels_matrices_store": {"kernels_params0": params["theta"], "inducing_points_locs0": params["Z"]}},
        "embedding": {"C0": params["C"], "d0": params["d"]}}
    ell_params = {"leg_quad_points": t(prob["quad_points"]).unsqueeze(-1),
                  "leg_quad_weights": t(prob["quad_weights"]).unsqueeze(-1)}
    model.setParamsAndData(measurements=measurements, initial_params=initial_params,
                           eLLCalculationParams=ell_params, priorCovRegParam=prob["reg"])
    out = ref_outputs(model, params, intermediates=False)
    print("pinv: ref elbo %.10f ell %.10f kl %.10f" % (out["ref_elbo"], out["ref_ell"], out["ref_kl"]))
    np.savez_compressed(os.path.join(HERE, "pinv_small.npz"), **problem_inputs(prob), **out)


def _softplus(input=None, x=None):
    """Link function for the Gauss-Hermite golden; the reference calls its link with `input=` at the quadrature nodes
    (expectedLogLikelihood.py:235) and with `x=` at the spikes (:249)."""
    return torch.nn.functional.softplus(input if input is not None else x)


def gh_small():
    """Non-exponential link with Gauss-Hermite quadrature (PointProcessELLQuad, expectedLogLikelihood.py:215-255).  The
    reference's factory never builds this model (svGPFAModelFactory.py:96-98 raises), so the object graph is assembled by
    hand, as the reference's own test does (stats/tests/test_expectedLogLikelihood.py:106-198), here with a softplus
    link on the ragged case: ELBO / ELL / KL and autograd gradients of every group."""
    import svGPFA.stats.expectedLogLikelihood
    import svGPFA.stats.kernelsMatricesStore
    import svGPFA.stats.klDivergence
    import svGPFA.stats.svEmbedding
    import svGPFA.stats.svLowerBound
    import svGPFA.stats.svPosteriorOnIndPoints
    import svGPFA.stats.svPosteriorOnLatents
    prob = ragged_problem()
    model0, params = ref_model_from_problem(prob)
    K, R, N = prob["K"], prob["R"], prob["N"]
    S = svGPFA.stats
    qU = S.svPosteriorOnIndPoints.SVPosteriorOnIndPointsChol()
    kmsZ = S.kernelsMatricesStore.IndPointsLocsKMS_Chol()
    kmsAll = S.kernelsMatricesStore.IndPointsLocsAndAllTimesKMS()
    kmsAssoc = S.kernelsMatricesStore.IndPointsLocsAndAssocTimesKMS()
    qKAll = S.svPosteriorOnLatents.SVPosteriorOnLatentsAllTimes(
        svPosteriorOnIndPoints=qU, indPointsLocsKMS=kmsZ, indPointsLocsAndTimesKMS=kmsAll)
    qKAssoc = S.svPosteriorOnLatents.SVPosteriorOnLatentsAssocTimes(
        svPosteriorOnIndPoints=qU, indPointsLocsKMS=kmsZ, indPointsLocsAndTimesKMS=kmsAssoc)
    qHAll = S.svEmbedding.LinearSVEmbeddingAllTimes(svPosteriorOnLatents=qKAll)
    qHAssoc = S.svEmbedding.LinearSVEmbeddingAssocTimes(svPosteriorOnLatents=qKAssoc)
    eLL = S.expectedLogLikelihood.PointProcessELLQuad(svEmbeddingAllTimes=qHAll, svEmbeddingAssocTimes=qHAssoc,
                                                      linkFunction=_softplus)
    klDiv = S.klDivergence.KLDivergence(indPointsLocsKMS=kmsZ, svPosteriorOnIndPoints=qU)
    model = S.svLowerBound.SVLowerBound(eLL=eLL, klDiv=klDiv)
    model.setKernels(kernels=model0.getKernels())
    t = lambda a: torch.from_numpy(np.array(a, dtype=np.float64, copy=True, order="C"))
    off, times, neuron = prob["spike_offsets"], prob["spike_times"], prob["spike_neuron"]
    measurements = []
    for r in range(R):
        tr_t, tr_n = times[off[r]:off[r + 1]], neuron[off[r]:off[r + 1]]
        measurements.append([t(tr_t[tr_n == n]) for n in range(N)])
    xh, wh = np.polynomial.hermite.hermgauss(20)
    herm_x, herm_w = xh.reshape(-1, 1), (wh / np.sqrt(np.pi)).reshape(-1, 1)
    initial_params = {
        "posterior_on_latents": {
            "posterior_on_ind_points": {"mean": params["mean"], "cholVecs": params["cholVecs"]},
            "kernels_matrices_store": {"kernels_params0": params["theta"], "inducing_points_locs0": params["Z"]}},
        "embedding": {"C0": params["C"], "d0": params["d"]}}
    ell_params = {"leg_quad_points": t(prob["quad_points"]).unsqueeze(-1),
                  "leg_quad_weights": t(prob["quad_weights"]).unsqueeze(-1),
                  "hermQuadPoints": t(herm_x), "hermQuadWeights": t(herm_w)}
    model.setParamsAndData(measurements=measurements, initial_params=initial_params,
                           eLLCalculationParams=ell_params, priorCovRegParam=prob["reg"])
    out = ref_outputs(model, params, intermediates=False)
    print("gh: ref elbo %.10f ell %.10f kl %.10f" % (out["ref_elbo"], out["ref_ell"], out["ref_kl"]))
    np.savez_compressed(os.path.join(HERE, "gh_small.npz"), **problem_inputs(prob), herm_x=herm_x, herm_w=herm_w, **out)


def rank1_small():
    """Non-default q(u) covariance: S = q q^T + diag(d^2) (svPosteriorOnIndPoints.py:104-119)."""
    prob = ragged_problem()
    rng = np.random.default_rng(5)
    K, R, N = prob["K"], prob["R"], prob["N"]
    qSVec = [0.3 * rng.standard_normal((R, M)) for M in prob["Ms"]]
    qSDiag = [0.2 + 0.1 * rng.random((R, M)) for M in prob["Ms"]]
    model0, params = ref_model_from_problem(prob)          # for kernels / measurements plumbing
    t = lambda a: torch.from_numpy(np.array(a, dtype=np.float64, copy=True, order="C"))
    qv = [t(a).unsqueeze(-1) for a in qSVec]
    qd = [t(a).unsqueeze(-1) for a in qSDiag]
    scales = prob.get("scales", [1.0] * K)
    lss = prob.get("lengthscale_scales", [1.0] * K)
    pss = prob.get("period_scales", [1.0] * K)
    kernels = []
    for k in range(K):
        if prob["kernel_types"][k] == synthetic.RBF:
            kernels.append(svGPFA.stats.kernels.ExponentialQuadraticKernel(scale=scales[k], lengthscaleScale=lss[k]))
        else:
            kernels.append(svGPFA.stats.kernels.PeriodicKernel(scale=scales[k], lengthscaleScale=lss[k],
                                                               periodScale=pss[k]))
    off, times, neuron = prob["spike_offsets"], prob["spike_times"], prob["spike_neuron"]
    measurements = []
    for r in range(R):
        tr_t, tr_n = times[off[r]:off[r + 1]], neuron[off[r]:off[r + 1]]
        measurements.append([t(tr_t[tr_n == n]) for n in range(N)])
    initial_params = {
        "posterior_on_latents": {
            "posterior_on_ind_points": {"mean": params["mean"], "qSVec0": qv, "qSDiag0": qd},
            "kernels_matrices_store": {"kernels_params0": params["theta"], "inducing_points_locs0": params["Z"]}},
        "embedding": {"C0": params["C"], "d0": params["d"]}}
    ell_params = {"leg_quad_points": t(prob["quad_points"]).unsqueeze(-1),
                  "leg_quad_weights": t(prob["quad_weights"]).unsqueeze(-1)}
    F = svGPFA.stats.svGPFAModelFactory
    model = F.SVGPFAModelFactory.buildModelPyTorch(kernels=kernels, indPointsCovRep=F.indPointsCovRank1PlusDiag)
    model.setParamsAndData(measurements=measurements, initial_params=initial_params,
                           eLLCalculationParams=ell_params, priorCovRegParam=prob["reg"])
    leaves = params["mean"] + qv + qd
    for x in leaves:
        x.requires_grad_(True)
    model.buildKernelsMatrices()
    ell = model._eLL.evalSumAcrossTrialsAndNeurons()
    kl = model._klDiv.evalSumAcrossLatentsAndTrials()
    elbo = model.eval()
    grads = torch.autograd.grad(elbo, leaves)
    out = {"ref_elbo": float(elbo), "ref_ell": float(ell), "ref_kl": float(kl)}
    names = [f"g_mean_{k}" for k in range(K)] + [f"g_qSVec_{k}" for k in range(K)] + [f"g_qSDiag_{k}" for k in range(K)]
    for nme, g in zip(names, grads):
        out[nme] = g.detach().numpy()[:, :, 0]
    for k in range(K):
        out[f"qSVec_{k}"], out[f"qSDiag_{k}"] = qSVec[k], qSDiag[k]
    print("rank1: ref elbo %.10f ell %.10f kl %.10f" % (out["ref_elbo"], out["ref_ell"], out["ref_kl"]))
    np.savez_compressed(os.path.join(HERE, "rank1_small.npz"), **problem_inputs(prob), **out)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "rank1":
        rank1_small()
    elif len(sys.argv) > 1 and sys.argv[1] == "gh":
        gh_small()
    elif len(sys.argv) > 1 and sys.argv[1] == "pinv":
        pinv_small()
    elif len(sys.argv) > 1 and sys.argv[1] == "cfg3":
        cfg3_shape()
    elif len(sys.argv) > 1 and sys.argv[1] == "em":
        em_small()
        cfg1_bundled()
    else:
        main()
        em_small()
        cfg1_bundled()
        rank1_small()
        cfg3_shape()
        pinv_small()
        gh_small()
