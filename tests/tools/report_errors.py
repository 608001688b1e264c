"""Print HIP-vs-reference relative errors per output group for the golden fixtures (GPU box)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from helpers import load_golden, rel_err
from svgpfa_amd.engine import engine_from_problem

for name in ("matlab_estep", "ragged_small"):
    prob, ref = load_golden(name)
    eng, P = engine_from_problem(prob)
    shared, gm, gc, gz = eng.elbo(P["mean"], P["cholVecs"], P["C"], P["d"], 15)
    eng.sync()
    elbo, ell, kl, gC, gd, gth = eng.split_shared(shared)
    print(name, "elbo rel", abs(float(elbo) - float(ref["ref_elbo"])) / abs(float(ref["ref_elbo"])),
          "ell", abs(float(ell) - float(ref["ref_ell"])) / abs(float(ref["ref_ell"])),
          "kl", abs(float(kl) - float(ref["ref_kl"])) / abs(float(ref["ref_kl"])))
    muq, varq, mus, vars_ = eng.latent_stats(P["mean"], P["cholVecs"])
    eng.sync()
    print("  muQuad", rel_err(muq.cpu().numpy(), ref["muQuad"]), "varQuad", rel_err(varq.cpu().numpy(), ref["varQuad"]),
          "muSpk", rel_err(mus.cpu().numpy(), ref["muSpikes"]), "varSpk", rel_err(vars_.cpu().numpy(), ref["varSpikes"]))
    for k in range(prob["K"]):
        print("  k", k, "chol", rel_err(eng.get_kzz_chol(k).cpu().numpy(), ref[f"KzzChol_{k}"]),
              "g_mean", rel_err(gm[k].cpu().numpy(), ref[f"g_mean_{k}"]),
              "g_chol", rel_err(gc[k].cpu().numpy(), ref[f"g_cholVecs_{k}"]),
              "g_Z", rel_err(gz[k].cpu().numpy(), ref[f"g_Z_{k}"]),
              "g_theta", rel_err(gth[k].cpu().numpy(), ref[f"g_theta_{k}"]))
    print("  g_C", rel_err(gC.cpu().numpy(), ref["g_C"]), "g_d", rel_err(gd.cpu().numpy(), ref["g_d"]))
    if name == "matlab_estep":
        print("  g_Z[0][0,:5]", gz[0][0, :5].cpu().numpy())
