"""The drop-in boundary without Python in the loop: examples/c_abi_demo.c (plain C compiled by gcc + HIP runtime) is built against
include/svgpfa.h, linked with libsvgpfa_hip.so and run on a seeded problem; its ELBO / ELL / KL and gradient-group sums
must match the oracle (tolerances of test_gpu_parity.py)."""
import os
import subprocess

import numpy as np
import pytest

from svgpfa_amd.utils import synthetic

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_program_through_the_c_abi(tmp_path):
    from oracle import svgpfa_oracle as orc
    prob = synthetic.make_problem(R=3, N=7, K=3, M=[9, 12, 6], kernels="mix", seed=41, n_quad=48, rate=14.0, jitter=1.0)
    K = prob["K"]
    path = os.path.join(str(tmp_path), "problem.bin")
    with open(path, "wb") as f:
        S = int(prob["spike_offsets"][-1])
        np.array([prob["R"], prob["N"], K, prob["Q"], S], dtype=np.int32).tofile(f)
        mk = []
        for k in range(K):
            mk += [prob["Ms"][k], 0 if prob["kernel_types"][k] == synthetic.RBF else 1]
        np.array(mk, dtype=np.int32).tofile(f)
        np.array([prob["reg"]], dtype=np.float64).tofile(f)
        prob["quad_points"].astype(np.float64).tofile(f)
        prob["quad_weights"].astype(np.float64).tofile(f)
        prob["spike_offsets"].astype(np.int64).tofile(f)
        prob["spike_times"].astype(np.float64).tofile(f)
        prob["spike_neuron"].astype(np.int32).tofile(f)
        for k in range(K):
            for key in ("Z", "mean", "cholVecs", "theta"):
                np.ascontiguousarray(prob[key][k], dtype=np.float64).tofile(f)
        np.ascontiguousarray(prob["C"], dtype=np.float64).tofile(f)
        np.ascontiguousarray(prob["d"], dtype=np.float64).tofile(f)
    exe = os.path.join(str(tmp_path), "c_abi_demo")
    csrc = os.path.join(REPO, "svgpfa_amd", "csrc")
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    subprocess.run(["gcc", "-std=c11", "-O2", "-D__HIP_PLATFORM_AMD__", os.path.join(REPO, "examples", "c_abi_demo.c"),
                    "-I", os.path.join(rocm, "include"), "-I", os.path.join(REPO, "include"), "-L", csrc, "-lsvgpfa_hip",
                    "-L", os.path.join(rocm, "lib"), "-lamdhip64", f"-Wl,-rpath,{csrc}",
                    f"-Wl,-rpath,{os.path.join(rocm, 'lib')}", "-o", exe], check=True, cwd=REPO)
    out = subprocess.run([exe, path], check=True, capture_output=True, text=True, timeout=300).stdout.split()
    got = [float(x) for x in out]
    ref = orc.elbo_and_grads(orc.from_numpy_problem(prob), 15)
    want = [float(ref["elbo"]), float(ref["ell"]), float(ref["kl"]), float(ref["g_C"].sum()), float(ref["g_d"].sum()),
            float(sum(g.sum() for g in ref["g_theta"])), float(sum(g.sum() for g in ref["g_mean"])),
            float(sum(g.sum() for g in ref["g_cholVecs"])), float(sum(g.sum() for g in ref["g_Z"]))]
    scale = [abs(w) for w in want[:3]] + [float(ref["g_C"].abs().sum()), float(ref["g_d"].abs().sum()),
                                          float(sum(g.abs().sum() for g in ref["g_theta"])),
                                          float(sum(g.abs().sum() for g in ref["g_mean"])),
                                          float(sum(g.abs().sum() for g in ref["g_cholVecs"])),
                                          float(sum(g.abs().sum() for g in ref["g_Z"]))]
    assert len(got) == 9
    for i, (g, w, s) in enumerate(zip(got, want, scale)):
        tol = 1e-9 if i < 3 else 1e-7
        assert abs(g - w) <= tol * max(s, 1e-300), (i, g, w)
