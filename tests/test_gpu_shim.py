"""GPU tests of the reference-API mirror (svgpfa_amd.stats): same call sequences as the reference's own
tests (stats/tests/test_svLowerBound.py, test_kernelsMatricesStore.py, test_svEM.py), expected values from
the golden vectors produced by running the reference (tests/golden/make_golden.py)."""
import io
import pickle

import numpy as np
import pytest
import torch

from helpers import load_golden, rel_err
from svgpfa_amd.utils import synthetic

pytestmark = pytest.mark.gpu


from helpers import build_kernels as _kernels, build_model as _model  # noqa: E402


def test_eval_pointProcess():
    """stats/tests/test_svLowerBound.py:18-106: |lbEval + obj| < 3e-4 on the MATLAB fixture."""
    prob, ref = load_golden("matlab_estep")
    model = _model(prob)
    lbEval = model.eval()
    assert abs(float(lbEval) + float(ref["matlab_obj"])) < 3e-4
    assert abs(float(lbEval) - float(ref["ref_elbo"])) < 1e-9 * abs(float(ref["ref_elbo"]))
    elbo, ell, kl = model.lastTerms()
    assert abs(float(ell) - (-5499.83650875)) < 3e-4      # test_expectedLogLikelihood.py:104
    assert abs(float(kl) - 537.65708147) < 1e-5           # test_klDivergence.py:62


@pytest.mark.parametrize("group", ["estep", "embedding", "kernels", "indpoints"])
def test_backward_fills_grad_of_live_tensors(group):
    """svEM.py:266-288: requires_grad on one group, closure, backward -> .grad of the very tensors returned
    by the getters; compared with the reference's autograd gradients."""
    prob, ref = load_golden("ragged_small")
    model = _model(prob)
    K = prob["K"]
    x = {"estep": model.getSVPosteriorOnIndPointsParams, "embedding": model.getSVEmbeddingParams,
         "kernels": model.getKernelsParams, "indpoints": model.getIndPointsLocs}[group]()
    for t in x:
        t.requires_grad_(True)
    if group in ("kernels", "indpoints"):
        model.buildKernelsMatrices()
    loss = -model.eval()
    loss.backward()
    names = {"estep": [f"g_mean_{k}" for k in range(K)] + [f"g_cholVecs_{k}" for k in range(K)],
             "embedding": ["g_C", "g_d"], "kernels": [f"g_theta_{k}" for k in range(K)],
             "indpoints": [f"g_Z_{k}" for k in range(K)]}[group]
    for t, nme in zip(x, names):
        assert t.grad is not None and t.grad.shape == t.shape
        got = -t.grad.cpu().numpy().reshape(ref[nme].shape)
        assert rel_err(got, ref[nme]) < 1e-9, (nme, rel_err(got, ref[nme]))
    others = [p for p in (model.getSVPosteriorOnIndPointsParams() + model.getSVEmbeddingParams()
                          + model.getKernelsParams() + model.getIndPointsLocs()) if all(p is not q for q in x)]
    assert all(p.grad is None for p in others)


@pytest.mark.parametrize("group", ["var", "emb", "theta", "Z"])
def test_flat_closure_equals_protocol_path(group):
    """SVLowerBound.evalFlat (the EM loop's fast closure: no autograd graph, gradients written by the engine into one flat
    buffer laid out like flatParams(group)) gives bit for bit what eval() + backward() puts into .grad."""
    prob, _ = load_golden("ragged_small")
    model = _model(prob)
    getter = {"var": model.getSVPosteriorOnIndPointsParams, "emb": model.getSVEmbeddingParams,
              "theta": model.getKernelsParams, "Z": model.getIndPointsLocs}[group]
    x = getter()
    flat = model.flatParams(group)
    assert flat is not None and flat.numel() == sum(t.numel() for t in x)
    assert torch.equal(flat, torch.cat([t.reshape(-1) for t in x]))        # views of one buffer, in getter order
    x[0].mul_(1.0)                                                          # an in-place update is seen through both
    assert flat.data_ptr() == x[0].data_ptr()
    for t in x:
        t.requires_grad_(True)
    if group == "emb":
        stats = model.computeSVPosteriorOnLatentsStats()
        val = model.evalELLSumAcrossTrialsAndNeurons(svPosteriorOnLatentsStats=stats)
    else:
        if group in ("theta", "Z"):
            model.buildKernelsMatrices()
        val = model.eval()
    val.backward()
    want = torch.cat([t.grad.reshape(-1) for t in x]).clone()
    for t in x:
        t.requires_grad_(False)
        t.grad = None
    v2, g2 = model.evalFlat(group)
    assert torch.equal(v2, val.detach())
    assert torch.equal(g2.reshape(-1), want)
    # a caller that rebinds a tensor of the group loses the flat layout, and is told so
    if group == "theta":
        model.getKernels()[0].setParams(model.getKernelsParams()[0].clone())
        assert model.flatParams("theta") is None
        with pytest.raises(ValueError):
            model.evalFlat("theta")


def test_stale_embedding_token_cases():
    """The spike sums are reused across value-only evaluations while C is unchanged (engine token).  In-place updates
    of C -- directly or through the flat buffer it is a view of -- and re-created parameters must all bring the
    spike pass back."""
    prob, _ = load_golden("ragged_small")
    model = _model(prob)
    ref_model = _model(prob)

    def check():
        a, b = model.eval(), ref_model.eval()
        ref_model._engine.set_embedding_token(0)           # the reference model never reuses
        assert torch.equal(a, b)
    check()
    check()                                                 # reuse path
    model.getSVEmbeddingParams()[0].mul_(1.3)
    ref_model.getSVEmbeddingParams()[0].mul_(1.3)
    check()
    model.flatParams("emb").add_(0.01)                      # optimiser step through the flat buffer
    ref_model.flatParams("emb").add_(0.01)
    check()
    st = pickle.loads(pickle.dumps({"model": model}))["model"]   # re-created tensors (new salt, new buffers)
    assert st.flatParams("var") is not None and st.flatParams("emb") is not None
    st.buildKernelsMatrices()
    assert torch.equal(st.eval(), ref_model.eval())


def test_embedding_mstep_closure():
    """svEM.py:225-232: frozen latent stats, ELL only."""
    prob, _ = load_golden("ragged_small")
    ref = np.load(__import__("os").path.join(__import__("helpers").GOLDEN, "estep_embedding.npz"))
    model = _model(prob)
    stats = model.computeSVPosteriorOnLatentsStats()
    C, d = model.getSVEmbeddingParams()
    C.requires_grad_(True), d.requires_grad_(True)
    ell = model.evalELLSumAcrossTrialsAndNeurons(svPosteriorOnLatentsStats=stats)
    ell.backward()
    assert abs(float(ell) - float(ref["ref_ell"])) < 1e-10 * max(1.0, abs(float(ref["ref_ell"])))
    assert rel_err(C.grad.cpu().numpy(), ref["g_C"]) < 1e-9
    assert rel_err(d.grad.cpu().numpy(), ref["g_d"]) < 1e-9
    with pytest.raises(NotImplementedError):
        model.evalELLSumAcrossTrialsAndNeurons(svPosteriorOnLatentsStats={"allTimes": None, "assocTimes": None})


def test_kernels_matrices_store_standalone():
    """stats/tests/test_kernelsMatricesStore.py:12-183: Kzz (+eps), Ktz / KttDiag at quad and spike times."""
    from svgpfa_amd.stats import kernelsMatricesStore as kms
    prob, ref = load_golden("matlab_estep")
    t = lambda a: torch.from_numpy(np.array(a, dtype=np.float64, copy=True, order="C"))
    Z = [t(a).unsqueeze(-1) for a in prob["Z"]]
    kernels = _kernels(prob)
    params0 = {"kernels_params0": [t(a) for a in prob["theta"]], "inducing_points_locs0": Z}
    s1 = kms.IndPointsLocsKMS_Chol()
    s1.setKernels(kernels=kernels)
    s1.setInitialParams(initial_params=params0)
    s1.setRegParam(reg_param=prob["reg"])
    s1.buildKernelsMatrices()
    for k in range(prob["K"]):
        assert rel_err(s1.getKzz()[k].cpu().numpy(), ref[f"Kzz_{k}"]) < 1e-12
    # solveForLatent == cholesky_solve with the stored factor (kernelsMatricesStore.py:132-134)
    m0 = t(prob["mean"][0]).unsqueeze(-1).to(s1.getKzz()[0].device)
    sol = s1.solveForLatent(input=m0, latentIndex=0)
    assert rel_err(torch.matmul(s1.getKzz()[0], sol).cpu().numpy(), m0.cpu().numpy()) < 1e-8
    s2 = kms.IndPointsLocsAndAllTimesKMS()
    s2.setKernels(kernels=kernels)
    s2.setInitialParams(initial_params=params0)
    s2.setTimes(times=t(prob["quad_points"]).unsqueeze(-1))
    s2.buildKernelsMatrices()
    for k in range(prob["K"]):
        assert rel_err(s2.getKtz()[k].cpu().numpy(), ref[f"KtzQuad_{k}"]) < 1e-12
    assert rel_err(s2.getKttDiag().cpu().numpy(), ref["KttDiagQuad"]) < 1e-15
    off = prob["spike_offsets"]
    s3 = kms.IndPointsLocsAndAssocTimesKMS()
    s3.setKernels(kernels=kernels)
    s3.setInitialParams(initial_params=params0)
    s3.setTimes(times=[t(prob["spike_times"][off[r]:off[r + 1]]) for r in range(prob["R"])])
    s3.buildKernelsMatrices()
    for k in range(prob["K"]):
        got = torch.cat(s3.getKtz()[k]).cpu().numpy()
        assert rel_err(got, ref[f"KtzSpikes_{k}"]) < 1e-12


def test_predict_and_cifs(golden_dir):
    prob, _ = load_golden("ragged_small")
    pr = np.load(golden_dir + "/ragged_predict.npz")
    model = _model(prob)
    times = torch.from_numpy(pr["times"]).unsqueeze(-1)
    mu, var = model.predictLatents(times=times)
    assert rel_err(mu.cpu().numpy(), pr["mu"]) < 1e-9 and rel_err(var.cpu().numpy(), pr["var"]) < 1e-9
    em, evar = model.predictEmbedding(times=times)
    assert rel_err(em.cpu().numpy(), pr["emb_mean"]) < 1e-9 and rel_err(evar.cpu().numpy(), pr["emb_var"]) < 1e-9
    cifs = model.computeExpectedPosteriorCIFs(times=times)
    assert len(cifs) == prob["R"] and len(cifs[0]) == prob["N"]
    assert rel_err(cifs[1][2].cpu().numpy(), np.exp(pr["emb_mean"][1, :, 2] + 0.5 * pr["emb_var"][1, :, 2])) < 1e-9


def _optim_params(em_max_iter):
    lb = dict(max_iter=20, lr=1.0, tolerance_grad=1e-7, tolerance_change=1e-9, line_search_fn="strong_wolfe")
    op = {"em_max_iter": em_max_iter}
    for step in ("estep", "mstep_embedding", "mstep_kernels", "mstep_indpointslocs"):
        op[f"{step}_estimate"] = True
        op[f"{step}_optim_params"] = dict(lb)
    return op


@pytest.mark.parametrize("optimizer", ["sharded", "torch"])
def test_em_small_matches_reference_history(optimizer):
    """The reference's SVEM_PyTorch.maximize (ECM, 2 iterations) vs the same loop over the HIP model, with the default
    Gram-form L-BFGS and with torch.optim.LBFGS (the reference's optimiser)."""
    from svgpfa_amd.stats.svEM import SVEM_PyTorch, ErrorTerminationInfo
    prob, ref = load_golden("em_small")
    model = _model(prob)
    out = io.StringIO()
    hist, elapsed, info, _ = SVEM_PyTorch(optimizer=optimizer).maximize(model=model, optim_params=_optim_params(2),
                                                                        method="ECM", out=out)
    assert not isinstance(info, ErrorTerminationInfo), info.message
    assert len(hist) == 3 and len(elapsed) == 3
    want = ref["lowerBoundHist"]
    assert abs(hist[0] - want[0]) < 1e-10 * abs(want[0])
    # L-BFGS line searches amplify rounding differences: loose tolerance on the iterates
    assert abs(hist[1] - want[1]) < 1e-6 * abs(want[1]) and abs(hist[2] - want[2]) < 2e-5 * abs(want[2])
    assert hist[2] > hist[1] > hist[0]
    assert "Iteration 01, estep start" in out.getvalue() and "nfeval" in out.getvalue()
    assert rel_err(model.getSVEmbeddingParams()[0].cpu().numpy(), ref["final_C"]) < 1e-4


@pytest.mark.parametrize("optimizer", ["sharded", "torch"])
def test_config1_bundled_data_two_em_iterations(optimizer):
    """BASELINE config 1: bundled simulated data, ini 545, em_max_iter = 2."""
    from svgpfa_amd.stats.svEM import SVEM_PyTorch, ErrorTerminationInfo
    prob, ref = load_golden("cfg1_bundled")
    model = _model(prob)
    hist, _, info, _ = SVEM_PyTorch(optimizer=optimizer).maximize(model=model, optim_params=_optim_params(2),
                                                                  method="ECM", out=io.StringIO())
    assert not isinstance(info, ErrorTerminationInfo), info.message
    want = ref["lowerBoundHist"]          # [277018.8745717274, 1003993.0510170484, 1007863.0982122809]
    assert abs(hist[0] - 277018.8745717274) < 1e-10 * 277018.8745717274
    assert abs(hist[1] - want[1]) < 1e-6 * abs(want[1])
    assert abs(hist[2] - want[2]) < 1e-6 * abs(want[2])


def test_not_pd_raises_linalg_error_and_em_reports_it():
    """utils/miscUtils.py:215 raises torch.linalg.LinAlgError; svEM.py:150-163 turns it into ErrorTerminationInfo."""
    prob = synthetic.make_problem(R=2, N=3, K=1, M=6, kernels="rbf", seed=9, n_quad=16, rate=5.0, reg=-0.5)
    prob["Z"][0][1, 3] = prob["Z"][0][1, 2]
    with pytest.raises(torch.linalg.LinAlgError):
        _model(prob)


def test_pickle_roundtrip():
    """svEM.py:89-92 pickles {"model": model}."""
    prob, ref = load_golden("ragged_small")
    model = _model(prob)
    v0 = float(model.eval())
    model2 = pickle.loads(pickle.dumps({"model": model}))["model"]
    assert model2._engine is None
    assert float(model2.eval()) == v0


def test_rank1_plus_diag_variational_covariance():
    """Factory switch indPointsCovRank1PlusDiag (stats/svGPFAModelFactory.py:65-67, svPosteriorOnIndPoints.py:86-119):
    q(u) covariances as q q^T + diag(d^2); value and gradients wrt mean, qSVec, qSDiag against the reference."""
    from svgpfa_amd.stats import svGPFAModelFactory as F
    from helpers import build_kernels
    prob, ref = load_golden("rank1_small")
    K, R, N = prob["K"], prob["R"], prob["N"]
    t = lambda a: torch.from_numpy(np.array(a, dtype=np.float64, copy=True, order="C"))
    off, times, neuron = prob["spike_offsets"], prob["spike_times"], prob["spike_neuron"]
    measurements = []
    for r in range(R):
        tt, nn = times[off[r]:off[r + 1]], neuron[off[r]:off[r + 1]]
        measurements.append([t(tt[nn == n]) for n in range(N)])
    initial_params = {
        "posterior_on_latents": {
            "posterior_on_ind_points": {"mean": [t(a).unsqueeze(-1) for a in prob["mean"]],
                                        "qSVec0": [t(ref[f"qSVec_{k}"]).unsqueeze(-1) for k in range(K)],
                                        "qSDiag0": [t(ref[f"qSDiag_{k}"]).unsqueeze(-1) for k in range(K)]},
            "kernels_matrices_store": {"kernels_params0": [t(a) for a in prob["theta"]],
                                       "inducing_points_locs0": [t(a).unsqueeze(-1) for a in prob["Z"]]}},
        "embedding": {"C0": t(prob["C"]), "d0": t(prob["d"])}}
    ell = {"leg_quad_points": t(prob["quad_points"]).unsqueeze(-1),
           "leg_quad_weights": t(prob["quad_weights"]).unsqueeze(-1)}
    model = F.SVGPFAModelFactory.buildModelPyTorch(kernels=build_kernels(prob), indPointsCovRep=F.indPointsCovRank1PlusDiag)
    model.setParamsAndData(measurements=measurements, initial_params=initial_params, eLLCalculationParams=ell,
                           priorCovRegParam=prob["reg"])
    x = model.getSVPosteriorOnIndPointsParams()
    assert len(x) == 3 * K
    for p in x:
        p.requires_grad_(True)
    val = model.eval()
    assert abs(float(val) - float(ref["ref_elbo"])) < 1e-9 * abs(float(ref["ref_elbo"]))
    _, ell_v, kl_v = model.lastTerms()
    assert abs(float(ell_v) - float(ref["ref_ell"])) < 1e-9 * abs(float(ref["ref_ell"]))
    assert abs(float(kl_v) - float(ref["ref_kl"])) < 1e-9 * abs(float(ref["ref_kl"]))
    val.backward()
    names = [f"g_mean_{k}" for k in range(K)] + [f"g_qSVec_{k}" for k in range(K)] + [f"g_qSDiag_{k}" for k in range(K)]
    for p, nme in zip(x, names):
        assert p.grad is not None
        assert rel_err(p.grad.cpu().numpy()[:, :, 0], ref[nme]) < 1e-8, (nme, rel_err(p.grad.cpu().numpy()[:, :, 0], ref[nme]))
    # the E-step of the EM loop runs on this representation too
    from svgpfa_amd.stats.svEM import SVEM_PyTorch
    for p in x:
        p.requires_grad_(False)
    before = float(model.eval())
    res = SVEM_PyTorch()._eStep(model=model, optim_params=dict(max_iter=5, lr=1.0, tolerance_grad=1e-7,
                                                                tolerance_change=1e-9, line_search_fn="strong_wolfe"))
    assert float(res["lowerBound"]) > before


def test_kms_not_positive_definite_raises_linalgerror():
    """The stand-alone KernelsMatricesStore raises what the reference raises (torch.linalg.LinAlgError from
    utils/miscUtils.py:215), not the engine's own exception type."""
    from svgpfa_amd.stats import kernelsMatricesStore as kms_mod
    prob, _ = load_golden("ragged_small")
    kms = kms_mod.IndPointsLocsKMS_Chol()
    kms.setKernels(_kernels(prob))
    kms.setKernelsParams([torch.from_numpy(np.array(t)) for t in prob["theta"]])
    Z = [torch.from_numpy(np.array(z)).unsqueeze(-1) for z in prob["Z"]]
    Z[0][1, 2, 0] = Z[0][1, 1, 0]
    kms.setIndPointsLocs(Z)
    kms.setRegParam(-0.5)
    with pytest.raises(torch.linalg.LinAlgError):
        kms.buildKernelsMatrices()


def test_pinv_model_against_the_reference_pinv_model():
    """kernelMatrixInvPInv (reference kernelsMatricesStore.py:146-165) through the factory: the engine's whitened
    algebra against the reference's own pseudo-inverse model (tests/golden/pinv_small.npz), value and every gradient."""
    from svgpfa_amd.stats import svGPFAModelFactory as F
    prob, ref = load_golden("pinv_small")
    model = _model(prob, kernelMatrixInvMethod=F.kernelMatrixInvPInv)
    K = prob["K"]
    groups = (model.getSVPosteriorOnIndPointsParams() + model.getSVEmbeddingParams() + model.getKernelsParams()
              + model.getIndPointsLocs())
    for t in groups:
        t.requires_grad_(True)
    model.buildKernelsMatrices()
    val = model.eval()
    val.backward()
    assert abs(float(val) - float(ref["ref_elbo"])) <= 1e-9 * abs(float(ref["ref_elbo"]))
    names = ([f"g_mean_{k}" for k in range(K)] + [f"g_cholVecs_{k}" for k in range(K)] + ["g_C", "g_d"]
             + [f"g_theta_{k}" for k in range(K)] + [f"g_Z_{k}" for k in range(K)])
    for t, nme in zip(groups, names):
        assert rel_err(t.grad.cpu().numpy().reshape(ref[nme].shape), ref[nme]) < 1e-8, nme


def test_pinv_kms_store():
    """The stand-alone pseudo-inverse store: Kzz as the Cholesky store builds it, solves by multiplication, and a
    singular Kzz (duplicated inducing point, no jitter) is handled instead of raising."""
    from svgpfa_amd.stats import kernelsMatricesStore as kms_mod
    prob, ref = load_golden("ragged_small")
    Z = [torch.from_numpy(np.array(z)).unsqueeze(-1) for z in prob["Z"]]
    kms = kms_mod.IndPointsLocsKMS_PInv()
    kms.setKernels(_kernels(prob))
    kms.setKernelsParams([torch.from_numpy(np.array(t)) for t in prob["theta"]])
    kms.setIndPointsLocs(Z)
    kms.setRegParam(prob["reg"])
    kms.buildKernelsMatrices()
    for k in range(prob["K"]):
        Kzz = kms.getKzz()[k]
        assert rel_err(Kzz.cpu().numpy(), ref[f"Kzz_{k}"]) < 1e-12
        x = torch.randn(Kzz.shape[0], Kzz.shape[1], 2, dtype=torch.float64, device=Kzz.device)
        assert torch.allclose(Kzz @ kms.solveForLatent(x, k), x, rtol=1e-7, atol=1e-8)
    Z[0][1, 2, 0] = Z[0][1, 1, 0]
    kms.setIndPointsLocs(Z)
    kms.setRegParam(0.0)
    kms.buildKernelsMatrices()                      # singular Kzz: the pseudo-inverse still exists
    P = kms._Kzz_inv[0][1]
    Kz = kms.getKzz()[0][1]
    assert torch.allclose(Kz @ P @ Kz, Kz, rtol=1e-6, atol=1e-8)


def _build_gh_model(prob, ref, link):
    from svgpfa_amd.stats import svGPFAModelFactory as F
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
    ell = {"leg_quad_points": t(prob["quad_points"]).unsqueeze(-1), "leg_quad_weights": t(prob["quad_weights"]).unsqueeze(-1),
           "hermQuadPoints": t(ref["herm_x"]), "hermQuadWeights": t(ref["herm_w"])}
    model = SVGPFAModelFactory.buildModelPyTorch(kernels=_kernels(prob), linkFunction=F.NonExponentialLink,
                                                 linkFunctionCallable=link)
    model.setParamsAndData(measurements=measurements, initial_params=initial_params, eLLCalculationParams=ell,
                           priorCovRegParam=prob["reg"])
    return model


def test_gauss_hermite_link_against_the_reference():
    """Non-exponential link (reference PointProcessELLQuad, expectedLogLikelihood.py:215-255) with a softplus link: value
    and the gradients of all four groups against the reference's own outputs (tests/golden/gh_small.npz).  The latent
    statistics, their backward (quadrature AND spike adjoints) and the KL term are the engine's."""
    prob, ref = load_golden("gh_small")
    model = _build_gh_model(prob, ref, torch.nn.functional.softplus)
    K = prob["K"]
    groups = (model.getSVPosteriorOnIndPointsParams() + model.getSVEmbeddingParams() + model.getKernelsParams()
              + model.getIndPointsLocs())
    for t in groups:
        t.requires_grad_(True)
    model.buildKernelsMatrices()
    val = model.eval()
    elbo, ell, kl = model.lastTerms()
    assert abs(float(val) - float(ref["ref_elbo"])) <= 1e-9 * abs(float(ref["ref_elbo"]))
    assert abs(float(ell) - float(ref["ref_ell"])) <= 1e-9 * abs(float(ref["ref_ell"]))
    assert abs(float(kl) - float(ref["ref_kl"])) <= 1e-9 * abs(float(ref["ref_kl"]))
    (-val).backward()                       # the EM loop minimises the negative bound: exercises the scaling path
    names = ([f"g_mean_{k}" for k in range(K)] + [f"g_cholVecs_{k}" for k in range(K)] + ["g_C", "g_d"]
             + [f"g_theta_{k}" for k in range(K)] + [f"g_Z_{k}" for k in range(K)])
    for t, nme in zip(groups, names):
        assert rel_err(-t.grad.cpu().numpy().reshape(ref[nme].shape), ref[nme]) < 1e-8, (nme, rel_err(-t.grad.cpu().numpy().reshape(ref[nme].shape), ref[nme]))


def test_gauss_hermite_with_exp_link_equals_the_fused_model():
    """With g = exp the Gauss-Hermite ELL equals the closed form the engine fuses (the reference's own check,
    stats/tests/test_expectedLogLikelihood.py:106-198): 20 Hermite nodes reproduce E[exp(h)] to quadrature accuracy, and
    the (2 v) scaling of the spike term cancels for log(exp(.))."""
    prob, ref = load_golden("gh_small")
    gh = _build_gh_model(prob, ref, torch.exp)
    fused = _model(prob)
    a, b = float(gh.eval()), float(fused.eval())
    assert abs(a - b) <= 1e-6 * abs(b), (a, b)


def test_gauss_hermite_embedding_closure_and_em_step():
    """The embedding M-step closure of a Gauss-Hermite model (evalELLSumAcrossTrialsAndNeurons with the statistics frozen,
    svEM.py:225-232) uses the Gauss-Hermite ELL, not the fused exponential one: its value is the ELL term of eval() at the
    same parameters (the reference's own value for it), its gradients are those of eval() with respect to C and d
    (the KL term does not depend on them), and the EM loop's embedding step raises the bound."""
    prob, ref = load_golden("gh_small")
    model = _build_gh_model(prob, ref, torch.nn.functional.softplus)
    model.buildKernelsMatrices()
    C, d = model.getSVEmbeddingParams()
    for t in (C, d):
        t.requires_grad_(True)
    val = model.eval()
    val.backward()
    gC, gd = C.grad.clone(), d.grad.clone()
    C.grad = d.grad = None
    for make in (model.computeSVPosteriorOnLatentsStats, model.freezeLatentsStats):   # frozen: falls back to the full stats
        stats = make()
        ell = model.evalELLSumAcrossTrialsAndNeurons(svPosteriorOnLatentsStats=stats)
        assert abs(float(ell) - float(ref["ref_ell"])) <= 1e-9 * abs(float(ref["ref_ell"]))
        ell.backward()
        assert rel_err(C.grad.cpu().numpy(), gC.cpu().numpy()) < 1e-10 and rel_err(d.grad.cpu().numpy(), gd.cpu().numpy()) < 1e-10
        C.grad = d.grad = None
    for t in (C, d):
        t.requires_grad_(False)
    times = torch.linspace(0.05, 0.95, 7, dtype=torch.double).reshape(1, -1, 1).repeat(prob["R"], 1, 1)
    eMean, eVar = model.predictEmbedding(times)
    cifs = model.computeExpectedPosteriorCIFs(times)          # expectedLogLikelihood.py:62-73: link(mean + var / 2)
    want = torch.nn.functional.softplus(eMean + 0.5 * eVar)
    assert len(cifs) == prob["R"] and len(cifs[0]) == prob["N"] and torch.equal(cifs[1][2], want[1, :, 2])
    from svgpfa_amd.stats.svEM import SVEM_PyTorch
    before = float(model.eval())
    step = dict(max_iter=5, lr=1.0, tolerance_grad=1e-7, tolerance_change=1e-9, line_search_fn="strong_wolfe")
    res = SVEM_PyTorch()._mStepEmbedding(model=model, optim_params=step)
    assert res["nfeval"] >= 2
    assert float(model.eval()) > before


@pytest.mark.parametrize("name", ["ragged_small", "oversize_neurons", "ragged_M", "no_spikes"])
def test_frozen_stats_equal_full_statistics(name):
    """freezeLatentsStats() (quadrature statistics + per-neuron sums of the spike means through the spike pass, no
    per-spike statistics materialised) leaves the embedding closure with the same value and gradients as
    computeSVPosteriorOnLatentsStats() (svLowerBound.py:72-81), also after the embedding has moved."""
    if name == "ragged_small":
        prob, _ = load_golden(name)
    elif name == "oversize_neurons":     # 12 000 spikes per (trial, neuron): dedicated chunks, split-chunk sums
        prob = synthetic.make_problem(R=2, N=3, K=2, M=[8, 12], kernels="mix", n_quad=40, rate=12000.0, seed=9, jitter=1.0)
    elif name == "no_spikes":            # a recording without a single spike: the spike-side sums are empty
        prob = synthetic.make_problem(R=2, N=3, K=2, M=[6, 9], kernels="mix", n_quad=30, rate=5.0, seed=3, jitter=1.0)
        prob["spike_offsets"] = np.zeros(3, dtype=np.int64)
        prob["spike_times"], prob["spike_neuron"] = np.zeros(0), np.zeros(0, dtype=np.int32)
        from oracle import svgpfa_oracle as orc
        want = float(orc.elbo_and_grads(orc.from_numpy_problem(prob), 0)["elbo"])
        assert abs(float(_model(prob).eval()) - want) <= 1e-9 * abs(want)
    else:                                # M up to 48 (NT = 3), latents that fall back to the pair loops
        prob = synthetic.make_problem(R=3, N=40, K=7, M=[48, 20, 33, 9, 16, 40, 5], kernels="mix", n_quad=50, rate=10.0,
                                      seed=7, jitter=1.0)
    m_full, m_frozen = _model(prob), _model(prob)
    s_full = m_full.computeSVPosteriorOnLatentsStats()
    s_frozen = m_frozen.freezeLatentsStats()
    for step in range(2):
        v1, g1 = m_full.evalFlat("emb")
        v2, g2 = m_frozen.evalFlat("emb")
        assert abs(float(v1) - float(v2)) <= 1e-10 * max(1.0, abs(float(v1)))
        assert rel_err(g2.cpu().numpy(), g1.cpu().numpy()) < 1e-10
        e1 = m_full.evalELLSumAcrossTrialsAndNeurons(svPosteriorOnLatentsStats=s_full)
        e2 = m_frozen.evalELLSumAcrossTrialsAndNeurons(svPosteriorOnLatentsStats=s_frozen)
        assert abs(float(e1) - float(e2)) <= 1e-10 * max(1.0, abs(float(e1)))
        for m in (m_full, m_frozen):                      # an optimiser step on the embedding; the statistics stay frozen
            m.flatParams("emb").mul_(1.01)
    with pytest.raises(NotImplementedError):
        m_frozen.evalELLSumAcrossTrialsAndNeurons(svPosteriorOnLatentsStats=s_full)


def test_infinite_bound_warns_where_the_value_is_read():
    """The reference warns inside eval() when the bound is infinite (svLowerBound.py:51-53).  Here eval() stays
    asynchronous and the tensor it returns raises the same warning when its value reaches the host -- float(), .item() --
    which is what a caller driving this model with the reference's own svEM.py does with every closure value."""
    import warnings
    from svgpfa_amd.stats.svLowerBound import _BoundValue
    import math
    prob, _ = load_golden("ragged_small")
    model = _model(prob)
    C = model.getSVEmbeddingParams()[0]
    C.requires_grad_(True)
    v = model.eval()
    assert isinstance(v, _BoundValue)
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        assert math.isfinite(float(v)) and math.isfinite((-v).item())      # finite: silent
    inf = torch.full((), float("inf"), dtype=torch.float64, device=v.device).as_subclass(_BoundValue)
    with pytest.warns(UserWarning, match="infinity lower bound"):
        float(inf)
    with pytest.warns(UserWarning, match="infinity lower bound"):
        (-inf).item()
    v.backward()                                                          # still an autograd node
    assert C.grad is not None and torch.isfinite(C.grad).all()


def test_latent_stats_node_without_the_kl_term():
    """The statistics node carries adjoints of ANY loss built on its outputs: a loss that leaves the KL term out (its
    adjoint is 0) must give finite gradients -- grad(ELL) = grad(ELL - KL) - grad(-KL) -- not the NaN a division by the
    KL adjoint would produce."""
    from svgpfa_amd.stats.svLowerBound import _LatentStatsFunction
    prob, ref = load_golden("gh_small")
    model = _build_gh_model(prob, ref, torch.nn.functional.softplus)
    params = model.getSVPosteriorOnIndPointsParams() + model.getKernelsParams() + model.getIndPointsLocs()
    for t in params:
        t.requires_grad_(True)
    model.buildKernelsMatrices()

    def grads(which):
        chol = model._chol_vecs()
        muq, varq, mus, vars_, kl = _LatentStatsFunction.apply(model, *model._mean, *chol, *model._theta(), *model._Z)
        ell = model._gauss_hermite_ell(muq, varq, mus, vars_)
        loss = {"ell": ell, "elbo": ell - kl, "mkl": -kl}[which]
        return [g.clone() for g in torch.autograd.grad(loss, params)]

    g_ell, g_elbo, g_mkl = grads("ell"), grads("elbo"), grads("mkl")
    for a, b, c in zip(g_ell, g_elbo, g_mkl):
        assert torch.isfinite(a).all()
        want = (b - c).cpu().numpy()
        assert rel_err(a.cpu().numpy(), want) < 1e-9


def test_gauss_hermite_model_with_trials_sharded():
    """The Gauss-Hermite model under trial sharding (the model's _reduce hook = an in-place sum over ranks): the bound and
    its two terms are summed in the forward, the gradients of the shared parameters (C, d, kernel parameters) in the
    backward, per-trial parameters keep their local gradients.  Emulated in one process by a reduction that doubles its
    argument (two ranks holding identical shards): bound and shared gradients double, per-trial gradients do not."""
    prob, ref = load_golden("gh_small")
    K = prob["K"]
    outs = []
    for shard in (False, True):
        model = _build_gh_model(prob, ref, torch.nn.functional.softplus)
        if shard:
            model._reduce = lambda t: t.mul_(2.0)
        groups = (model.getSVPosteriorOnIndPointsParams() + model.getSVEmbeddingParams() + model.getKernelsParams()
                  + model.getIndPointsLocs())
        for t in groups:
            t.requires_grad_(True)
        model.buildKernelsMatrices()
        val = model.eval()
        val.backward()
        outs.append((float(val), [t.grad.clone() for t in groups]))
    (v1, g1), (v2, g2) = outs
    assert abs(v2 - 2.0 * v1) <= 1e-12 * abs(v1)
    shared_idx = set(range(2 * K, 2 * K + 2 + K))                  # C, d, kernel parameters
    for i, (a, b) in enumerate(zip(g1, g2)):
        f = 2.0 if i in shared_idx else 1.0
        assert rel_err(b.cpu().numpy(), f * a.cpu().numpy()) < 1e-12, i
