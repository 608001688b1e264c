"""Random small problems (shapes, kernel mixes, lengthscales, rates, trial lengths) through the engine and the oracle:
a sweep beyond the fixed cases of tests/test_gpu_parity.py.  usage: python tests/tools/fuzz_parity.py [cases] [seed]
[fp64|mixed] [largest N]   (mixed: the tolerances of tests/test_gpu_mixed.py; largest N: 40, above 112 the fused ELL kernel
-- SVGPFA_ELL_FUSED=1 SVGPFA_ELL_CLS=0 on problems this small -- runs two neuron groups)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import torch
from helpers import rel_err
from oracle import svgpfa_oracle as orc
from svgpfa_amd.engine import engine_from_problem
from svgpfa_amd.utils import synthetic

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
precision = sys.argv[3] if len(sys.argv) > 3 else "fp64"
n_max = int(sys.argv[4]) if len(sys.argv) > 4 else 40
tol_s, tol_a = (1e-9, 1e-7) if precision == "fp64" else (1e-5, 1e-2)
worst = {"scalar": 0.0, "array": 0.0}
for c in range(n_cases):
    K = int(rng.integers(1, 7))
    case = dict(R=int(rng.integers(1, 6)), N=int(rng.integers(1, n_max + 1)), K=K,
                M=[int(rng.integers(1, 65)) for _ in range(K)],
                kernels=[synthetic.RBF if rng.random() < 0.5 else synthetic.PERIODIC for _ in range(K)],
                n_quad=int(rng.integers(8, 120)), rate=float(np.exp(rng.uniform(np.log(0.5), np.log(400.0)))),
                trial_len=float(rng.uniform(0.5, 8.0)), seed=int(rng.integers(0, 1 << 30)), jitter=1.0, reg=1e-3)
    prob = synthetic.make_problem(**case)
    for k in range(K):            # lengthscales from short (pair-loop fallback) to long
        prob["theta"][k][0] = float(np.exp(rng.uniform(np.log(0.03), np.log(2.0)))) * case["trial_len"] ** 0.5
    mask = int(rng.choice([15, 15, 1, 4, 8, 12, 6, 9]))
    eng, P = engine_from_problem(prob, precision=precision)
    shared, gm, gc, gz = eng.elbo(P["mean"], P["cholVecs"], P["C"], P["d"], mask)
    eng.sync()
    if mask == 1:     # E-step evaluation: the second one loads the whitened tiles the first one stored -- bit-identical
        sh2, gm2, gc2, _ = eng.elbo(P["mean"], P["cholVecs"], P["C"], P["d"], mask)
        eng.sync()
        assert torch.equal(sh2, shared) and all(torch.equal(a, b) for a, b in zip(gm2 + gc2, gm + gc)), "cached E-step evaluation differs"
    elbo, ell, kl, gC, gd, gth = eng.split_shared(shared)
    ref = orc.elbo_and_grads(orc.from_numpy_problem(prob), mask)
    scale = max(1.0, abs(float(ref["elbo"]))) if precision == "fp64" else max(1.0, abs(float(ref["elbo"])), abs(float(ref["ell"])),
                                                                             float(prob["spike_offsets"][-1]))
    es = max(abs(float(elbo) - float(ref["elbo"])) / scale,
             abs(float(kl) - float(ref["kl"])) / max(1.0, abs(float(ref["kl"]))))
    ea = 0.0
    if mask & 1:
        ea = max([ea] + [rel_err(gm[k].cpu().numpy(), ref["g_mean"][k].numpy()[:, :, 0]) for k in range(K)]
                 + [rel_err(gc[k].cpu().numpy(), ref["g_cholVecs"][k].numpy()[:, :, 0]) for k in range(K)])
    if mask & 2:
        ea = max(ea, rel_err(gC.cpu().numpy(), ref["g_C"].numpy()), rel_err(gd.cpu().numpy(), ref["g_d"].numpy()))
    if mask & 4:
        ea = max([ea] + [rel_err(gth[k].cpu().numpy(), ref["g_theta"][k].numpy()) for k in range(K)])
    if mask & 8:
        ea = max([ea] + [rel_err(gz[k].cpu().numpy(), ref["g_Z"][k].numpy()[:, :, 0]) for k in range(K)])
    worst["scalar"], worst["array"] = max(worst["scalar"], es), max(worst["array"], ea)
    flag = "" if (es < tol_s and ea < tol_a) else "   <-- OUT OF TOLERANCE"
    print(f"case {c:3d} R{case['R']} N{case['N']} K{K} M{case['M']} mask {mask:2d} spikes {int(prob['spike_offsets'][-1]):6d} "
          f"harm {eng.series_harmonics()}  scalar {es:.1e}  arrays {ea:.1e}{flag}", flush=True)
    eng.close()
print("worst", worst)
