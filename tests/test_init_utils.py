"""Configuration-file parameter assembly (svgpfa_amd/utils/initUtils.py) against the parameters the reference's own
initUtils produced for its bundled estimation ini (number 545), which tests/golden/cfg1_bundled.npz holds."""
import os

import numpy as np
import pytest
import torch

from helpers import load_golden
from svgpfa_amd.utils import initUtils

INI = """
[model_structure_params]
n_latents = 2
[data_structure_params]
trials_start_time = 0.0
trials_end_time = 1.0
[variational_params0]
variational_means0_filename = {d}/means.csv
variational_covs0_filename = {d}/covs.csv
[embedding_params0]
c0_filename = {d}/C.csv
d0_filename = {d}/d.csv
[kernels_params0]
k_type_latent0 = exponentialQuadratic
k_lengthscale0_latent0 = 2.0
k_type_latent1 = exponentialQuadratic
k_lengthscale0_latent1 = 1.0
[ind_points_locs_params0]
common_n_ind_points = 9
ind_points_locs0_layout = equidistant
[optim_params]
prior_cov_reg_param = 1e-3
n_quad = 200
optim_method = ecm
em_max_iter = 200
estep_estimate = True
estep_max_iter = 20
estep_lr = 1.0
estep_tolerance_grad = 1e-7
estep_tolerance_change = 1e-9
estep_line_search_fn = strong_wolfe
mstep_embedding_estimate = True
mstep_embedding_max_iter = 20
mstep_embedding_lr = 1.0
mstep_embedding_tolerance_grad = 1e-7
mstep_embedding_tolerance_change = 1e-9
mstep_embedding_line_search_fn = strong_wolfe
mstep_kernels_estimate = True
mstep_kernels_max_iter = 20
mstep_kernels_lr = 1.0
mstep_kernels_tolerance_grad = 1e-7
mstep_kernels_tolerance_change = 1e-9
mstep_kernels_line_search_fn = strong_wolfe
mstep_indpointslocs_estimate = False
mstep_indpointslocs_max_iter = 20
mstep_indpointslocs_lr = 1.0
mstep_indpointslocs_tolerance_grad = 1e-7
mstep_indpointslocs_tolerance_change = 1e-9
mstep_indpointslocs_line_search_fn = strong_wolfe
verbose = False
"""


def write_config(tmp_path, prob):
    d = str(tmp_path)
    np.savetxt(os.path.join(d, "means.csv"), prob["mean"][0][0], delimiter=",")
    np.savetxt(os.path.join(d, "covs.csv"), 1e-2 * np.eye(9), delimiter=",")
    np.savetxt(os.path.join(d, "C.csv"), prob["C"], delimiter=",")
    np.savetxt(os.path.join(d, "d.csv"), prob["d"], delimiter=",")
    ini = os.path.join(d, "init.ini")
    with open(ini, "w") as f:
        f.write(INI.format(d=d))
    return ini


def test_bundled_ini_gives_reference_parameters(tmp_path):
    prob, _ = load_golden("cfg1_bundled")
    ini = write_config(tmp_path, prob)
    R, N, K = prob["R"], prob["N"], prob["K"]
    params, ktypes = initUtils.getParamsAndKernelsTypes(
        n_neurons=N, n_trials=R, n_latents=K, trials_start_times=[0.0] * R, trials_end_times=[1.0] * R,
        config_file_params_spec=initUtils.readConfigFile(ini),
        dynamic_params_spec={"optim_params": {"em_max_iter": "2"}})
    assert ktypes == ["exponentialQuadratic"] * 2
    ip = params["initial_params"]
    qu = ip["posterior_on_latents"]["posterior_on_ind_points"]
    kms = ip["posterior_on_latents"]["kernels_matrices_store"]
    for k in range(K):
        assert qu["mean"][k].shape == (R, 9, 1) and qu["cholVecs"][k].shape == (R, 45, 1)
        assert np.allclose(qu["mean"][k][:, :, 0].numpy(), prob["mean"][k], rtol=0, atol=1e-15)
        assert np.allclose(qu["cholVecs"][k][:, :, 0].numpy(), prob["cholVecs"][k], rtol=0, atol=1e-15)
        assert np.allclose(kms["inducing_points_locs0"][k][:, :, 0].numpy(), prob["Z"][k], rtol=0, atol=1e-15)
        assert np.allclose(kms["kernels_params0"][k].numpy(), prob["theta"][k])
    assert np.allclose(ip["embedding"]["C0"].numpy(), prob["C"]) and ip["embedding"]["C0"].shape == (N, K)
    assert np.allclose(ip["embedding"]["d0"].numpy().reshape(-1), prob["d"]) and ip["embedding"]["d0"].shape == (N, 1)
    ell = params["ell_calculation_params"]
    assert np.allclose(ell["leg_quad_points"][:, :, 0].numpy(), prob["quad_points"], rtol=0, atol=1e-14)
    assert np.allclose(ell["leg_quad_weights"][:, :, 0].numpy(), prob["quad_weights"], rtol=0, atol=1e-14)
    op = params["optim_params"]
    assert op["em_max_iter"] == 2 and op["n_quad"] == 200 and op["prior_cov_reg_param"] == 1e-3
    assert op["optim_method"] == "ecm" and op["verbose"] is False
    assert op["mstep_indpointslocs_estimate"] is False and op["estep_estimate"] is True
    assert op["estep_optim_params"] == dict(max_iter=20, lr=1.0, tolerance_grad=1e-7, tolerance_change=1e-9,
                                            line_search_fn="strong_wolfe")


def test_defaults_short_kernel_format_and_errors():
    R, N, K = 3, 4, 2
    defaults = initUtils.getDefaultParamsDict(n_neurons=N, n_trials=R, n_latents=K, common_n_ind_points=5)
    params, ktypes = initUtils.getParamsAndKernelsTypes(
        n_neurons=N, n_trials=R, n_latents=K, trials_start_times=[0.0, 0.5, 1.0], trials_end_times=[1.0, 2.0, 3.0],
        default_params_spec=defaults,
        config_file_params_spec={"kernels_params0": {"k_types": "periodic", "k_lengthscales0": "1.5", "k_periods0": "0.4"},
                                 "embedding_params0": {"c0_random_seed": "3", "d0_random_seed": "4"}})
    assert ktypes == ["periodic", "periodic"]
    kms = params["initial_params"]["posterior_on_latents"]["kernels_matrices_store"]
    assert torch.equal(kms["kernels_params0"][1], torch.tensor([1.5, 0.4], dtype=torch.double))
    Z = kms["inducing_points_locs0"][0]
    assert Z.shape == (R, 5, 1) and float(Z[1, 0, 0]) == 0.5 and float(Z[1, -1, 0]) == 2.0
    q = params["ell_calculation_params"]["leg_quad_points"]
    assert q.shape == (R, 200, 1) and 1.0 < float(q[2].min()) and float(q[2].max()) < 3.0
    assert abs(float(params["ell_calculation_params"]["leg_quad_weights"][1].sum()) - 1.5) < 1e-12
    qu = params["initial_params"]["posterior_on_latents"]["posterior_on_ind_points"]
    assert qu["cholVecs"][0].shape == (R, 15, 1) and abs(float(qu["cholVecs"][0][0, 0, 0]) - 0.1) < 1e-15
    assert params["initial_params"]["embedding"]["C0"].shape == (N, K)
    with pytest.raises(ValueError, match="k_periods0"):
        initUtils.getParamsAndKernelsTypes(
            n_neurons=N, n_trials=R, n_latents=K, trials_start_times=[0.0] * R, trials_end_times=[1.0] * R,
            default_params_spec=defaults,
            config_file_params_spec={"kernels_params0": {"k_types": "periodic", "k_lengthscales0": "1.5"}})


@pytest.mark.gpu
def test_driver_reproduces_bundled_em_history(tmp_path):
    """BASELINE config 1 through the driver: ini file + spikes -> 2 EM iterations; history pinned by the reference's
    own run (tests/golden/make_golden.py em): tight on the initial bound, loose on the L-BFGS iterates."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))
    import doEstimateSVGPFA as driver
    prob, ref = load_golden("cfg1_bundled")
    ini = write_config(tmp_path, prob)
    with open(ini) as f:
        text = f.read().replace("mstep_indpointslocs_estimate = False", "mstep_indpointslocs_estimate = True")
    with open(ini, "w") as f:
        f.write(text)
    spikes = os.path.join(str(tmp_path), "spikes.npz")
    np.savez(spikes, spike_offsets=prob["spike_offsets"], spike_times=prob["spike_times"],
             spike_neuron=prob["spike_neuron"], N=prob["N"])
    cwd = os.getcwd()
    try:
        hist = driver.main(["--spikes_filename", spikes, "--est_init_config_filename", ini, "--em_max_iter", "2",
                            "--results_prefix", os.path.join(str(tmp_path), "out", "run")])
    finally:
        os.chdir(cwd)
    want = ref["lowerBoundHist"]
    assert abs(hist[0] - want[0]) <= 1e-10 * abs(want[0])
    for a, b in zip(hist[1:], want[1:]):
        assert abs(a - b) <= 2e-5 * abs(b), (hist, want)
    assert os.path.exists(os.path.join(str(tmp_path), "out", "run_estimatedModel.pickle"))
