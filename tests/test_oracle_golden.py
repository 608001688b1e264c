"""Pin the CPU oracle (oracle/svgpfa_oracle.py) to the reference: the MATLAB known answers
the reference's own tests hold, and golden vectors produced by running the reference
(tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

from oracle import svgpfa_oracle as orc
from svgpfa_amd.utils import synthetic
from helpers import load_golden, rel_err

# fp64 tolerances: the oracle restates the same arithmetic in the same precision; only
# summation order differs from the reference.
TOL_SCALAR = 1e-11
TOL_ARRAY = 1e-9


def test_matlab_known_answers():
    """stats/tests/test_svLowerBound.py:106, test_expectedLogLikelihood.py:104, test_klDivergence.py:62."""
    prob, ref = load_golden("matlab_estep")
    elbo, ell, kl = orc.elbo_terms(orc.from_numpy_problem(prob))
    assert abs(float(elbo) + float(ref["matlab_obj"])) < 3e-4
    assert abs(float(elbo) - (-6037.49359022)) < 3e-4
    assert abs(float(ell) - (-5499.83650875)) < 3e-4
    assert abs(float(kl) - 537.65708147) < 1e-5
    if "matlab_KLd" in ref:
        assert abs(float(kl) - float(ref["matlab_KLd"])) < 1e-5


@pytest.mark.parametrize("name", ["matlab_estep", "ragged_small"])
def test_scalars_and_intermediates(name):
    prob, ref = load_golden(name)
    p = orc.from_numpy_problem(prob)
    elbo, ell, kl = orc.elbo_terms(p)
    assert abs(float(elbo) - float(ref["ref_elbo"])) <= TOL_SCALAR * abs(float(ref["ref_elbo"])) + 1e-10
    assert abs(float(ell) - float(ref["ref_ell"])) <= TOL_SCALAR * abs(float(ref["ref_ell"])) + 1e-10
    assert abs(float(kl) - float(ref["ref_kl"])) <= TOL_SCALAR * abs(float(ref["ref_kl"])) + 1e-10
    Kzz, Kchol = orc.build_kzz(p)
    Ktz_q, Ktt_q = orc.build_ktz_quad(p)
    Ktz_s, _ = orc.build_ktz_spikes(p)
    qS = orc.build_covs_from_chol_vecs(p["cholVecs"])
    for k in range(prob["K"]):
        assert rel_err(Kzz[k].numpy(), ref[f"Kzz_{k}"]) < TOL_ARRAY
        assert rel_err(Kchol[k].numpy(), ref[f"KzzChol_{k}"]) < TOL_ARRAY
        assert rel_err(Ktz_q[k].numpy(), ref[f"KtzQuad_{k}"]) < TOL_ARRAY
        got = np.concatenate([x.numpy().reshape(-1, prob["Ms"][k]) for x in Ktz_s[k]], 0)
        assert rel_err(got, ref[f"KtzSpikes_{k}"]) < TOL_ARRAY
        assert rel_err(qS[k].numpy(), ref[f"qSigma_{k}"]) < TOL_ARRAY
    assert rel_err(Ktt_q.numpy(), ref["KttDiagQuad"]) < TOL_ARRAY
    stats = orc.compute_latents_stats(p)
    assert rel_err(stats["allTimes"][0].numpy(), ref["muQuad"]) < TOL_ARRAY
    assert rel_err(stats["allTimes"][1].numpy(), ref["varQuad"]) < TOL_ARRAY
    assert rel_err(torch.cat(stats["assocTimes"][0]).numpy(), ref["muSpikes"]) < TOL_ARRAY
    assert rel_err(torch.cat(stats["assocTimes"][1]).numpy(), ref["varSpikes"]) < TOL_ARRAY
    eM, eV = orc.embedding_quad(p, *stats["allTimes"])
    assert rel_err(eM.numpy(), ref["embMeanQuad"]) < TOL_ARRAY
    assert rel_err(eV.numpy(), ref["embVarQuad"]) < TOL_ARRAY
    sM, sV = orc.embedding_spikes(p, *stats["assocTimes"])
    assert rel_err(torch.cat(sM).numpy(), ref["embMeanSpikes"]) < TOL_ARRAY
    assert rel_err(torch.cat(sV).numpy(), ref["embVarSpikes"]) < TOL_ARRAY


@pytest.mark.parametrize("name", ["matlab_estep", "ragged_small"])
def test_gradients(name):
    prob, ref = load_golden(name)
    out = orc.elbo_and_grads(orc.from_numpy_problem(prob), orc.GRAD_ALL)
    for k in range(prob["K"]):
        assert rel_err(out["g_mean"][k].numpy()[:, :, 0], ref[f"g_mean_{k}"]) < TOL_ARRAY
        assert rel_err(out["g_cholVecs"][k].numpy()[:, :, 0], ref[f"g_cholVecs_{k}"]) < TOL_ARRAY
        assert rel_err(out["g_Z"][k].numpy()[:, :, 0], ref[f"g_Z_{k}"]) < TOL_ARRAY
        assert rel_err(out["g_theta"][k].numpy(), ref[f"g_theta_{k}"]) < TOL_ARRAY
    assert rel_err(out["g_C"].numpy(), ref["g_C"]) < TOL_ARRAY
    assert rel_err(out["g_d"].numpy(), ref["g_d"]) < TOL_ARRAY


def test_cfg2_seeded_inputs_and_outputs(golden_dir):
    """BASELINE config 2: inputs regenerate bit-identically from the seed; ELBO/grad parity."""
    ref = np.load(golden_dir + "/cfg2.npz")
    prob = synthetic.make_config("cfg2")
    assert int(prob["spike_offsets"][-1]) == int(ref["n_spikes"])
    assert prob["spike_times"].sum() == float(ref["spike_times_sum"])
    assert prob["C"].sum() == float(ref["C_sum"])
    out = orc.elbo_and_grads(orc.from_numpy_problem(prob), orc.GRAD_ALL)
    assert abs(float(out["elbo"]) - float(ref["ref_elbo"])) < 1e-10 * abs(float(ref["ref_elbo"]))
    for k in range(prob["K"]):
        assert rel_err(out["g_mean"][k].numpy()[:, :, 0], ref[f"g_mean_{k}"]) < TOL_ARRAY
        assert rel_err(out["g_cholVecs"][k].numpy()[:, :, 0], ref[f"g_cholVecs_{k}"]) < TOL_ARRAY
        assert rel_err(out["g_Z"][k].numpy()[:, :, 0], ref[f"g_Z_{k}"]) < TOL_ARRAY
        assert rel_err(out["g_theta"][k].numpy(), ref[f"g_theta_{k}"]) < TOL_ARRAY
    assert rel_err(out["g_C"].numpy(), ref["g_C"]) < TOL_ARRAY
    assert rel_err(out["g_d"].numpy(), ref["g_d"]) < TOL_ARRAY


def test_cfg3_shape_seeded_inputs_and_outputs(golden_dir):
    """North-star geometry (BASELINE config 3 shapes, first 6 trials of the SURVEY 8(d) inputs): the oracle equals the
    reference on ELBO / ELL / KL and every gradient group."""
    ref = np.load(golden_dir + "/cfg3_shape.npz")
    prob = synthetic.make_config("cfg3", n_trials=int(ref["n_trials"]))
    assert int(prob["spike_offsets"][-1]) == int(ref["n_spikes"])
    assert prob["spike_times"].sum() == float(ref["spike_times_sum"])
    assert prob["C"].sum() == float(ref["C_sum"])
    out = orc.elbo_and_grads(orc.from_numpy_problem(prob), orc.GRAD_ALL)
    for key in ("elbo", "ell", "kl"):
        assert abs(float(out[key]) - float(ref["ref_" + key])) < 1e-10 * abs(float(ref["ref_" + key]))
    for k in range(prob["K"]):
        assert rel_err(out["g_mean"][k].numpy()[:, :, 0], ref[f"g_mean_{k}"]) < TOL_ARRAY
        assert rel_err(out["g_cholVecs"][k].numpy()[:, :, 0], ref[f"g_cholVecs_{k}"]) < TOL_ARRAY
        assert rel_err(out["g_Z"][k].numpy()[:, :, 0], ref[f"g_Z_{k}"]) < TOL_ARRAY
        assert rel_err(out["g_theta"][k].numpy(), ref[f"g_theta_{k}"]) < TOL_ARRAY
    assert rel_err(out["g_C"].numpy(), ref["g_C"]) < TOL_ARRAY
    assert rel_err(out["g_d"].numpy(), ref["g_d"]) < TOL_ARRAY


def test_embedding_mstep_closure(golden_dir):
    prob, _ = load_golden("ragged_small")
    ref = np.load(golden_dir + "/estep_embedding.npz")
    p = orc.from_numpy_problem(prob)
    stats = orc.compute_latents_stats(p)
    ell, gC, gd = orc.ell_given_stats_and_grads(p, stats)
    assert abs(float(ell) - float(ref["ref_ell"])) < 1e-11 * max(1.0, abs(float(ref["ref_ell"])))
    assert rel_err(gC.numpy(), ref["g_C"]) < TOL_ARRAY
    assert rel_err(gd.numpy(), ref["g_d"]) < TOL_ARRAY


def test_predict_latents(golden_dir):
    prob, _ = load_golden("ragged_small")
    ref = np.load(golden_dir + "/ragged_predict.npz")
    p = orc.from_numpy_problem(prob)
    mu, var = orc.predict_latents(p, torch.from_numpy(ref["times"]).unsqueeze(-1))
    assert rel_err(mu.numpy(), ref["mu"]) < TOL_ARRAY
    assert rel_err(var.numpy(), ref["var"]) < TOL_ARRAY


def test_rank1_plus_diag_covariance():
    """q(u) covariance S = q q^T + diag(d^2) (reference svPosteriorOnIndPoints.py:104-119): the oracle fed with the
    Cholesky vectors of that S reproduces the reference's ELBO / ELL / KL for the rank-1 + diagonal model."""
    prob, ref = load_golden("rank1_small")
    K = prob["K"]
    chol = []
    for k in range(K):
        q, d = ref[f"qSVec_{k}"], ref[f"qSDiag_{k}"]
        S = q[:, :, None] * q[:, None, :] + np.stack([np.diag(x ** 2) for x in d])
        L = np.linalg.cholesky(S)
        i, j = np.tril_indices(L.shape[-1])
        chol.append(np.ascontiguousarray(L[:, i, j]))
    prob["cholVecs"] = chol
    elbo, ell, kl = orc.elbo_terms(orc.from_numpy_problem(prob))
    assert abs(float(elbo) - float(ref["ref_elbo"])) <= 1e-10 * abs(float(ref["ref_elbo"]))
    assert abs(float(ell) - float(ref["ref_ell"])) <= 1e-10 * abs(float(ref["ref_ell"]))
    assert abs(float(kl) - float(ref["ref_kl"])) <= 1e-10 * abs(float(ref["ref_kl"]))
