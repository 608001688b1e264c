"""Condense gpurun_out/profile_<tag>/ (made by collect_profiles.sh) into the committed artefacts:
  profiles/<name>_kernel_stats.csv   rocprofv3 --kernel-trace --stats per-kernel summary (copied)
  profiles/<name>_pmc.json           per-kernel, per-dispatch averages of the PMC passes
  profiles/<name>_pmc_summary.txt    the same as a table
HBM bytes follow MI355X_MICROARCH.md: FETCH_SIZE / WRITE_SIZE are reported in KB; on gfx950 FETCH_SIZE tallies
128-B requests of wide coalesced streams at 64 B.  Our loads are 8 B per lane (not the calibrated 16 B pattern),
so both the raw figure and the doubled upper bound are stored; WRITE_SIZE is taken as reported.

usage: python tests/tools/digest_profiles.py <tag> <name> [config variant]
"""
import csv
import glob
import json
import os
import re
import shutil
import sys
from collections import defaultdict

REPO = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")


def short(name):
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*$", "", name)


def per_kernel(path_glob):
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for f in glob.glob(path_glob, recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                a = acc[short(row["Kernel_Name"])][row["Counter_Name"]]
                a[0] += float(row["Counter_Value"])
                a[1] += 1
    return {k: {c: v[0] / v[1] for c, v in d.items()} for k, d in acc.items()}


def main():
    tag, name = sys.argv[1], sys.argv[2]
    src = os.path.join(REPO, "gpurun_out", f"profile_{tag}")
    dst = os.path.join(REPO, "profiles")
    stats = glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True)
    shutil.copy(stats[0], os.path.join(dst, f"{name}_kernel_stats.csv"))
    avg_ns = {}
    with open(stats[0]) as fh:
        for row in csv.DictReader(fh):
            avg_ns[short(row["Name"])] = float(row["AverageNs"])
    merged = defaultdict(dict)
    for sub in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_sq2"):
        for k, d in per_kernel(os.path.join(src, sub, "**", "*counter_collection.csv")).items():
            merged[k].update(d)
    out = {}
    for k, d in merged.items():
        e = dict(d)
        if "FETCH_SIZE" in d:
            e["hbm_read_bytes_raw"] = d["FETCH_SIZE"] * 1024.0
            e["hbm_read_bytes_x2_upper"] = 2.0 * d["FETCH_SIZE"] * 1024.0
        if "WRITE_SIZE" in d:
            e["hbm_write_bytes"] = d["WRITE_SIZE"] * 1024.0
        if k in avg_ns:
            e["avg_ns_trace"] = avg_ns[k]
        out[k] = e
    sha = None
    if os.path.exists(os.path.join(src, "csrc_sha.txt")):
        sha = open(os.path.join(src, "csrc_sha.txt")).read().strip()
    with open(os.path.join(dst, f"{name}_pmc.json"), "w") as fh:
        json.dump({"source": f"collect_profiles.sh {tag}: separate rocprofv3 --pmc passes, per-dispatch averages",
                   "kernel_source_sha": sha,
                   "config": sys.argv[3] if len(sys.argv) > 3 else "cfg3",
                   "variant": sys.argv[4] if len(sys.argv) > 4 else "all",
                   "kernels": out}, fh, indent=1, sort_keys=True)
    cols = ["FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU", "SQ_INSTS_VALU_MFMA_MOPS_F64", "SQ_ACTIVE_INST_VALU",
            "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY", "SQ_INSTS_LDS",
            "SQ_LDS_BANK_CONFLICT"]
    with open(os.path.join(dst, f"{name}_pmc_summary.txt"), "w") as fh:
        fh.write("per-dispatch averages; FETCH_SIZE / WRITE_SIZE in KB as reported by rocprofv3 (see *_pmc.json)\n")
        fh.write("%-34s %10s " % ("kernel", "avg_us") + " ".join("%12s" % c[-12:] for c in cols) + "\n")
        for k in sorted(out, key=lambda k: -out[k].get("avg_ns_trace", 0)):
            fh.write("%-34s %10.1f " % (k[:34], out[k].get("avg_ns_trace", 0) / 1e3)
                     + " ".join("%12.4g" % out[k].get(c, float("nan")) for c in cols) + "\n")
    print("wrote", name)


if __name__ == "__main__":
    main()
