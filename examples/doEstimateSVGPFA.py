"""Estimate an svGPFA model on the MI355X engine from a spikes file and a configuration file -- the counterpart
of the reference's examples/scripts/doEstimateSVGPFA.py:22-139 (same flow: load spikes -> parameters from
the ini file and command-line overrides -> kernels -> model -> SVEM_PyTorch.maximize -> pickle the results).

Spikes file: a pickle holding {"spikes": spikes_times[r][n]} (the reference's simRes format) or an .npz with the
CSR arrays spike_offsets / spike_times / spike_neuron and N (the layout of tests/golden/cfg1_bundled.npz).

    python examples/doEstimateSVGPFA.py --spikes_filename data.pickle --est_init_config_filename init.ini \
        --n_latents 2 --em_max_iter 2 --results_prefix results/run1
Extra --section.key=value arguments override configuration-file entries (e.g. --optim_params.em_max_iter=5).
"""
import argparse
import os
import pickle
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

import svgpfa_amd.stats.svEM as svEM                                   # noqa: E402
import svgpfa_amd.stats.svGPFAModelFactory as factory                  # noqa: E402
from svgpfa_amd.utils import initUtils, miscUtils                      # noqa: E402


def load_spikes(filename):
    if filename.endswith(".npz"):
        z = np.load(filename)
        off, times, neuron, N = z["spike_offsets"], z["spike_times"], z["spike_neuron"], int(z["N"])
        out = []
        for r in range(len(off) - 1):
            tt, nn = times[off[r]:off[r + 1]], neuron[off[r]:off[r + 1]]
            out.append([torch.from_numpy(np.ascontiguousarray(tt[nn == n])) for n in range(N)])
        return out
    with open(filename, "rb") as f:
        return pickle.load(f)["spikes"]


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--spikes_filename", required=True)
    ap.add_argument("--est_init_config_filename", required=True)
    ap.add_argument("--n_latents", type=int, default=None, help="default: model_structure_params.n_latents of the ini")
    ap.add_argument("--trials_start_time", type=float, default=None)
    ap.add_argument("--trials_end_time", type=float, default=None)
    ap.add_argument("--em_max_iter", type=int, default=None)
    ap.add_argument("--results_prefix", default=None, help="write <prefix>_estimatedModel.pickle")
    args, remaining = ap.parse_known_args(argv)

    spikes_times = load_spikes(args.spikes_filename)
    n_trials, n_neurons = len(spikes_times), len(spikes_times[0])
    config = initUtils.readConfigFile(args.est_init_config_filename)
    dynamic = {}
    for item in remaining:                       # --section.key=value
        key, _, value = item.lstrip("-").partition("=")
        section, _, name = key.partition(".")
        if not name or not value:
            raise SystemExit(f"cannot parse override {item!r}: expected --section.key=value")
        dynamic.setdefault(section, {})[name] = value
    if args.em_max_iter is not None:
        dynamic.setdefault("optim_params", {})["em_max_iter"] = args.em_max_iter
    n_latents = args.n_latents or int(config.get("model_structure_params", {}).get("n_latents", 0))
    if n_latents <= 0:
        raise SystemExit("n_latents: give --n_latents or [model_structure_params] n_latents")
    ds = config.get("data_structure_params", {})
    t0 = args.trials_start_time if args.trials_start_time is not None else float(ds.get("trials_start_time", 0.0))
    t1 = args.trials_end_time if args.trials_end_time is not None else float(ds.get("trials_end_time", 1.0))
    os.chdir(os.path.dirname(os.path.abspath(args.est_init_config_filename)) or ".")   # relative *_filename entries
    params, kernels_types = initUtils.getParamsAndKernelsTypes(
        n_trials=n_trials, n_neurons=n_neurons, n_latents=n_latents, trials_start_times=[t0] * n_trials,
        trials_end_times=[t1] * n_trials, dynamic_params_spec=dynamic, config_file_params_spec=config,
        default_params_spec=initUtils.getDefaultParamsDict(n_neurons=n_neurons, n_trials=n_trials, n_latents=n_latents))

    kernels = miscUtils.buildKernels(
        kernels_types=kernels_types,
        kernels_params=params["initial_params"]["posterior_on_latents"]["kernels_matrices_store"]["kernels_params0"])
    model = factory.SVGPFAModelFactory.buildModelPyTorch(kernels=kernels)
    model.setParamsAndData(measurements=spikes_times, initial_params=params["initial_params"],
                           eLLCalculationParams=params["ell_calculation_params"],
                           priorCovRegParam=params["optim_params"]["prior_cov_reg_param"])
    start = time.time()
    lowerBoundHist, elapsedTimeHist, terminationInfo, iterationsModelParams = svEM.SVEM_PyTorch().maximize(
        model=model, optim_params=params["optim_params"], method=params["optim_params"]["optim_method"],
        verbose=params["optim_params"]["verbose"])
    print("lowerBoundHist:", lowerBoundHist)
    print("elapsed: {:.2f} s ({:s})".format(time.time() - start, terminationInfo.message))
    if args.results_prefix:
        os.makedirs(os.path.dirname(os.path.abspath(args.results_prefix)) or ".", exist_ok=True)
        with open(args.results_prefix + "_estimatedModel.pickle", "wb") as f:
            pickle.dump({"lowerBoundHist": lowerBoundHist, "elapsedTimeHist": elapsedTimeHist,
                         "terminationInfo": terminationInfo, "iterationModelParams": iterationsModelParams,
                         "model": model}, f)
        print("Saved results to {:s}".format(args.results_prefix + "_estimatedModel.pickle"))
    return lowerBoundHist


if __name__ == "__main__":
    main()
