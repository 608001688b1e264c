#!/usr/bin/env python3
"""bench.py -- ELBO+gradient evaluations per second of the HIP engine (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config cfg3] [--variant all|estep] [--dtype f64|mixed]

One "step" = one complete evaluation of the sparse-variational lower bound and its gradients on
the named synthetic workload (seeded), inputs already resident in HBM:
  variant "all"   : rebuild kernels (Kzz, Cholesky) + ELBO + gradients of all four parameter
                    groups -- the closure of the reference's svLBFGS.py:162-174 (headline)
  variant "estep" : ELBO + gradients of the variational parameters, kernels not rebuilt --
                    the closure of the reference's svEM.py:218-223
With N > 1 -- one rank per GPU, started either by torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in
the environment; WORLD_SIZE must equal --gpus) or, when `python bench.py --gpus N` is run bare, by this script itself
(launch_ranks: N child interpreters, the parent never touches a GPU and relays rank 0's line; any rank failing, or a
line whose n_gpus is not N, is a non-zero exit) -- the 500 trials are sharded by
spike count across ranks (total work fixed: "strong" scaling) and every step ends with ONE RCCL
all-reduce of the shared vector [ELBO, ELL, KL, dC, dd, dtheta].  `value` is the BLOCKING form
(svgpfa_amd.distributed.allreduce_shared: the compute stream waits for the reduced vector before the
next evaluation starts -- what an optimiser's closure needs, reference stats/svEM.py:274-294); two
more loops of K steps each run behind the headline's timed region and are reported beside it:
`value_host_synced` (the host also waits for the reduced vector every step, as a closure that reads
the bound does) and `value_pipelined` (alternating output buffers, the exchange of evaluation i
enqueued beside the kernels of evaluation i + 1: throughput of INDEPENDENT evaluations, an upper
bound no dependent loop reaches).

Rank 0 prints one JSON line.  Extra objects:
  roofline     : the kernel with the largest device time in this run -- k_item_bwd (f64 MFMA + VALU) since
                 round 3; the spike pass (class path: k_spike_pre / k_spike_cls / k_spike_comb, plus the
                 pair-by-pair k_spike for items without a series; fp64 VALU bound) follows as
                 `roofline_second`.  fp64 operations the kernel(s) must execute per launch -- counted
                 from the instruction mix (class_path_flops / PAIR_FLOPS / item_bwd_flops below, FMA = 2,
                 other fp64 arithmetic = 1, ldexp / cvt = 0) -- over the mean device
                 time from hipEvents recorded inside the timed region on the stream of the launch,
                 against the 78.6 TFLOP/s fp64 peak; frac <= 1 by construction.  SURVEY 8(d)'s flop-EQUIVALENT convention (24 per kernel evaluation,
                 x3 for value + gradients) is reported beside it as `survey_convention_*`, never
                 as `frac`.
  cpu_baseline : the CPU oracle (oracle/svgpfa_oracle.py, PyTorch-CPU fp64, the reference's
                 unfused algorithm) timed on this host on a bounded sample of the same workload:
                 1 warm-up + 3 repetitions, min and mean, CPU model stated.
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

FP64_PEAK_TFLOPS = 78.6   # MI355X fp64 vector = fp64 matrix peak (AMD public figure)
HBM_PEAK_GBS = 8000.0

# fp64 operations per (spike, inducing point) pair of the spike pass, from the instruction mix of
# svgpfa_amd/csrc/kernels_spike.h (FMA = 2 flops, add / mul / rndne = 1, ldexp / cvt = 0):
#   2^y kernel (sv_math.h sv_exp2_neg): rndne + sub + 10 FMA                      = 22
#   RBF      full : u = t - z, y = -u u, 2^y, ck = c kt, p0 += ck, 3 FMA (p1, p2, mu)      = 32
#   periodic full : y = 2 FMA, 2^y, ck, p0 += ck, 5 FMA (qa, qb, qc, qd, mu)                = 38
#   RBF      value: u, y, 2^y, 1 FMA (p0)                                                   = 26
#   periodic value: cs = mul + FMA, y = FMA, 2^y, 1 FMA (p0)                                = 29
# Per-spike set-up (loads, one sincos per spike of a periodic latent, the dC scan) is NOT counted:
# the figure is a lower bound of the executed operations, so the fraction cannot exceed 1.
PAIR_FLOPS = {("exponentialQuadratic", True): 32, ("periodic", True): 38,
              ("exponentialQuadratic", False): 26, ("periodic", False): 29}
# --dtype mixed (BASELINE config 5): the same pair loop in fp32 with ONE v_exp_f32 (counted as 1 operation) for 2^y:
#   RBF full: sub, mul, exp, mul, add, 3 FMA = 11;  periodic full: 2 FMA, exp, mul, add, 5 FMA = 17
#   RBF value: sub, mul, exp, FMA = 5;              periodic value: mul + FMA, FMA, exp, FMA = 8
PAIR_FLOPS_F32 = {("exponentialQuadratic", True): 11, ("periodic", True): 17,
                  ("exponentialQuadratic", False): 5, ("periodic", False): 8}
FP32_PEAK_TFLOPS = 157.3  # MI355X fp32 vector peak (MI355X_MICROARCH.md)


# Class path of the spike pass (svgpfa_amd/csrc/kernels_spike_cls.h): latents that share the angular frequency of their
# cosine series form a class, and a spike costs, per harmonic and CLASS, a rotation (2 mul + 2 FMA = 6 flops) and two adds
# (8 flops; 12 with the time-weighted moments of a periodic class when Z / theta gradients are wanted).  On top, per
# (row, member latent, harmonic): 4 flops of the A-moment product (f64 MFMA) and, with dC, 4 flops of the row's dC sum.
# Per-spike set-up (one sincos, loads) and the harmonics a register pass computes beyond H are NOT counted: a lower
# bound of the executed operations, so the fraction cannot exceed 1.
CLASS_FLOPS = {2: 8, 4: 12}


def class_plan(prob, harmonics):
    """The device's class table (k_spike_classes) re-derived on the host: a list of classes, each
    dict(type, members, H = [harmonics per trial, -1: pair loop])."""
    R, K = prob["R"], prob["K"]
    off = prob["spike_offsets"]
    ls_sc = prob.get("lengthscale_scales", [1.0] * K)
    p_sc = prob.get("period_scales", [1.0] * K)
    classes = []
    per = {}
    for k in range(K):                                   # periodic: one class per distinct period (<= 16 members)
        if prob["kernel_types"][k] != "periodic" or harmonics[k] < 0:
            continue
        key = float(prob["theta"][k][1]) / float(p_sc[k])
        lst = per.setdefault(key, [])
        if len(lst) < 16:
            lst.append(k)
    for key, mem in per.items():
        classes.append(dict(type="periodic", members=mem, H=[max(harmonics[k] for k in mem)] * R))
    hs = {k: 0.5 / (float(prob["theta"][k][0]) / float(ls_sc[k])) ** 2
          for k in range(K) if prob["kernel_types"][k] != "periodic" and harmonics[k] == -2}
    left = dict(hs)
    while left:                                          # RBF: greedy from the longest lengthscale, members within x2
        hmin = min(left.values())
        mem = [k for k in sorted(left) if left[k] <= 4.0 * hmin][:16]
        for k in mem:
            left.pop(k)
        Hs = []
        sumM = sum(prob["Ms"][k] for k in mem)
        for r in range(R):
            t = prob["spike_times"][off[r]:off[r + 1]]
            lo, hi = (float(t.min()), float(t.max())) if len(t) else (0.0, 0.0)
            for k in mem:
                lo, hi = min(lo, float(prob["Z"][k][r].min())), max(hi, float(prob["Z"][k][r].max()))
            L = (hi - lo) + math.sqrt(41.5 / hmin)
            H = 0
            for k in mem:
                nmax = math.sqrt(46.0 * hs[k] * L * L) / math.pi
                H = 65 if not (nmax < 64) else max(H, int(nmax) + 1)
            Hs.append(-1 if (H > 64 or 10 * (H + 2) > 21 * sumM) else H)
        classes.append(dict(type="rbf", members=mem, H=Hs))
    return classes


def spike_pass_flops(prob, full, want_c, table=None, harmonics=None):
    """Operations one spike pass must execute, this rank's shard: the class path's (CLASS_FLOPS per (spike, harmonic,
    class) + the per-member products) for the (trial, latent) items it serves, PAIR_FLOPS x (spike, inducing point) pairs
    for the rest.  Returns (class path flops, pair flops)."""
    table = table or PAIR_FLOPS
    off = prob["spike_offsets"]
    Sr = np.diff(off).astype(float)
    Nrows = prob["N"]                                    # ~ one row per (trial, neuron) with spikes
    cls_fl, pair_fl = 0.0, 0.0
    served = np.zeros((prob["R"], prob["K"]), dtype=bool)
    for c in class_plan(prob, harmonics if harmonics is not None else [-1] * prob["K"]):
        nv = 4 if (c["type"] == "periodic" and full) else 2
        for r, H in enumerate(c["H"]):
            if H < 0:
                continue
            served[r, c["members"]] = True
            cls_fl += Sr[r] * (H + 1) * CLASS_FLOPS[nv]
            cls_fl += Nrows * len(c["members"]) * (H + 1) * nv * 2                       # A-moment product
            if want_c:
                cls_fl += Nrows * len(c["members"]) * (H + 1) * 4                        # dC rows
    for k, (M, t) in enumerate(zip(prob["Ms"], prob["kernel_types"])):
        pair_fl += float(Sr[~served[:, k]].sum()) * M * table[(t, bool(full))]
    return float(cls_fl), float(pair_fl)


# k_item_bwd (kernels_item_bwd.h): fp64 operations per (quadrature node, inducing point) of phase 1 on the VALU, from its
# instruction mix -- kernel value (difference / angle addition, 2^y = 22, scale) + the adjoint accumulations:
#   RBF 28 (value, gu) + 11 with Z / theta gradients (Gk, gk ktil, delta, dz, th0);  periodic 30 + 21 (cos / sin of the
#   difference again, dz, th0, th1)
ITEM_VALU_FLOPS = {("exponentialQuadratic", True): 39, ("exponentialQuadratic", False): 28,
                   ("periodic", True): 51, ("periodic", False): 30}


def item_bwd_flops(prob, mask):
    """fp64 operations one k_item_bwd launch must execute: its v_mfma_f64_16x16x4_f64 (2 048 flops each, counted from the
    tile structure: NT = ceil(M / 16) tiles per side, triangular factors skip the zero tiles) plus the kernel
    evaluations and adjoint accumulations of phase 1 on the VALU (ITEM_VALU_FLOPS).  Phase 2's and phase 3's VALU work is
    not counted: a lower bound, so the fraction cannot exceed 1."""
    R, K, Q = prob["R"], prob["K"], prob["Q"]
    if not (mask & 13):
        return 0.0
    full, var = bool(mask & 12), bool(mask & 1)
    NT = (max(prob["Ms"]) + 15) // 16
    tri = lambda n: n * (n + 1) // 2
    tiles = (Q + 15) // 16
    tp = tiles * (2 * tri(NT) + (NT * NT if full else 0))            # per node tile: Linv k, [D kt], Gt += kt gv kt^T
    if full:
        tp += NT * 2 * tri(NT)                                         # D = Linv^T (Vt Vt^T - I), one panel per wave
        tp += 3 * tri(NT) * NT + NT * tri(NT) + tri(NT) * NT           # C, Vt Vt^T; At Linv; Linv^T (At Linv)
    if var:
        tp += sum((NT - c) ** 2 for c in range(NT)) + sum(tri(NT - c) for c in range(NT))   # dL_S
    mfma = tp * 4 * 2048.0 * R * K
    valu = sum(R * Q * M * ITEM_VALU_FLOPS[(t, full)] for M, t in zip(prob["Ms"], prob["kernel_types"]))
    return mfma + float(valu)


def kernel_source_sha():
    """Content hash of the ELBO path's device code: a PMC digest only describes the kernels it was measured on.
    Comments and blank lines do not count (a reworded comment does not invalidate a measurement); lbfgs_*: the
    optimiser's vector kernels, a separate translation unit that no evaluation launches."""
    import hashlib
    import re
    h = hashlib.sha256()
    csrc = os.path.join(REPO, "svgpfa_amd", "csrc")
    for f in sorted(os.listdir(csrc)):
        if f.endswith((".h", ".hip")) and not f.startswith("lbfgs_"):
            with open(os.path.join(csrc, f), "r") as fh:
                text = fh.read()
            text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
            text = re.sub(r"//[^\n]*", "", text)
            lines = [" ".join(ln.split()) for ln in text.splitlines()]
            code = "\n".join(ln for ln in lines if ln)
            h.update(f.encode() + b"\0" + code.encode())
    return h.hexdigest()[:16]


def algorithmic_work(prob, variant):
    """flop-equivalents and bytes per evaluation, and the spike kernel's share."""
    R, N, K, Q = prob["R"], prob["N"], prob["K"], prob["Q"]
    S = int(prob["spike_offsets"][-1])
    Sbar = S / max(R, 1)
    Ms = prob["Ms"]
    T_fwd = sum(R * M * (M + Q + Sbar) for M in Ms)
    F_fwd = (sum(R * (3 * M**3 + Q * (2 * M**2 + 4 * M) + 2 * M * Sbar) for M in Ms)
             + R * (4 * Q * K * N + 3 * Q * N) + 2 * S * K)
    flops_eval = 3 * (F_fwd + 24 * T_fwd)
    T_spk = sum(M * S for M in Ms)
    F_spk = sum(2 * M * S for M in Ms) + 2 * S * K
    flops_spike = (3 if variant == "all" else 1) * (F_spk + 24 * T_spk)   # value-only pass: forward work only
    ngrad = sum(R * (M + M * (M + 1) // 2) for M in Ms)
    if variant == "all":
        ngrad += sum(R * M for M in Ms) + N * K + N + sum(1 if t == "exponentialQuadratic" else 2
                                                           for t in prob["kernel_types"])
    bytes_eval = 8 * (S + 2 * R * Q + sum(R * (2 * M + M * (M + 1) // 2) for M in Ms) + N * K + N) \
        + 4 * S + 8 * ngrad + 8
    # spike pass, compulsory traffic of what it actually moves: spike times (8 S; the neuron ids are only used at ingest) and
    # the row table once, C, Z and a = Kzz^-1 m in; per (trial, latent) the sums P0 / P1 (2 M) and T (2) out; with dC the
    # two planes of per-(trial, neuron, latent) sums out.  (The class path reads the times once per CLASS: measured
    # traffic above this figure by that factor is not waste of a kernel, it is the price of separate launches per class.)
    bytes_spike = 8 * S + 16 * R * N + 8 * (N * K + sum(R * 2 * M for M in Ms)) + 8 * sum(R * (2 * M + 2) for M in Ms)
    if variant == "all":
        bytes_spike += 2 * 8 * R * N * K
    return dict(flops_eval=flops_eval, flops_spike=flops_spike, bytes_eval=bytes_eval, bytes_spike=bytes_spike,
                kernel_evals=T_fwd)


def cpu_baseline(config, variant, sample_trials, reps=3):
    """Time the CPU oracle on the first `sample_trials` trials of the workload (same seed)."""
    import torch
    from oracle import svgpfa_oracle as orc
    from svgpfa_amd.utils import synthetic
    full = synthetic.make_config(config)
    R = full["R"]
    sample = synthetic.shard_trials(full, 0, 1)
    ns = min(sample_trials, R)
    sub = dict(sample)
    off = sample["spike_offsets"]
    sub["R"] = ns
    sub["spike_offsets"] = off[:ns + 1].copy()
    sub["spike_times"] = sample["spike_times"][:off[ns]]
    sub["spike_neuron"] = sample["spike_neuron"][:off[ns]]
    for key in ("quad_points", "quad_weights"):
        sub[key] = sample[key][:ns]
    for key in ("Z", "mean", "cholVecs"):
        sub[key] = [a[:ns] for a in sample[key]]
    mask = 15 if variant == "all" else 1
    # threads = the PHYSICAL cores this process may run on: hyper-thread siblings are counted once
    torch.set_num_threads(_physical_cores())
    warm = dict(sub)   # tiny warm-up (2 trials) so that one-off torch initialisation is not timed
    warm["R"] = 2
    warm["spike_offsets"] = off[:3].copy()
    warm["spike_times"], warm["spike_neuron"] = sample["spike_times"][:off[2]], sample["spike_neuron"][:off[2]]
    for key in ("quad_points", "quad_weights"):
        warm[key] = sample[key][:2]
    for key in ("Z", "mean", "cholVecs"):
        warm[key] = [a[:2] for a in sample[key]]
    orc.elbo_and_grads(orc.from_numpy_problem(warm), mask)
    p = orc.from_numpy_problem(sub)
    times = []
    for _ in range(max(1, reps)):       # 1 warm-up + >= 3 timed repetitions, min and mean
        t0 = time.perf_counter()
        orc.elbo_and_grads(p, mask)
        times.append(time.perf_counter() - t0)
    frac = ns / R
    tmin, tmean = min(times), sum(times) / len(times)
    return dict(value=frac / tmin, value_mean=frac / tmean, unit="evals/s", cores=int(torch.get_num_threads()),
                kind="port", cpu_model=_cpu_model(), repetitions=len(times), seconds=[round(t, 3) for t in times],
                sample=f"oracle ({'ALL' if mask == 15 else 'E-step'} closure) on the first {ns} of {R} trials of "
                       f"{config} ({int(off[ns])} spikes), {len(times)} repetitions after a warm-up, min {tmin:.2f} s / "
                       f"mean {tmean:.2f} s; value = ({ns}/{R}) / min time, value_mean from the mean "
                       f"(linear in trials; the reference itself is super-linear, SURVEY 3.4)")


def _physical_cores():
    """Threads for the CPU baseline: the PHYSICAL cores this process may use -- distinct
    (package, core) pairs of its affinity mask, capped by the cgroup CPU quota when there is one and by 16, the share of
    the host a one-GPU job has on this pool (the affinity mask of the GPU box shows every core of the host; the first
    round-3 run with one thread per visible core did not finish one repetition in seven minutes)."""
    cores = set()
    for c in os.sched_getaffinity(0):
        base = "/sys/devices/system/cpu/cpu%d/topology/" % c
        try:
            with open(base + "physical_package_id") as f1, open(base + "core_id") as f2:
                cores.add((f1.read().strip(), f2.read().strip()))
        except OSError:
            cores.add(("?", str(c)))
    n = max(1, len(cores))
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:
            q, per = fh.read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except (OSError, ValueError):
        pass
    return min(n, 16)


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def measured_traffic(config, variant, full_mask, prefix="k_spike"):
    """HBM bytes per launch of the kernel `prefix` from a committed PMC digest (profiles/*_pmc.json, made by
    tests/tools/collect_profiles.sh + digest_profiles.py in separate rocprofv3 --pmc passes of this same
    command).  Counters cannot be read inside the timed run, so the figure comes from a digest -- and only from one
    taken on THESE kernels: the digest records the content hash of svgpfa_amd/csrc at collection time and is
    ignored (traffic = null) when the device code has changed since.  Returns (bytes, file, valu_insts, sha)."""
    import glob
    sha = kernel_source_sha()
    best = None
    for f in sorted(glob.glob(os.path.join(REPO, "profiles", "*_pmc.json"))):
        try:
            with open(f) as fh:
                d = json.load(fh)
        except (OSError, ValueError):
            continue
        if d.get("config") != config or d.get("variant") != variant or d.get("kernel_source_sha") != sha:
            continue
        tot, n = 0.0, 0     # the spike pass of one evaluation = the k_spike launches of all covariance types
        insts = 0.0
        for k, e in d.get("kernels", {}).items():
            # prefix "k_spike": every kernel of the spike pass (k_spike_pre / _cls / _comb and the pair-by-pair k_spike; the
            # digest is per config and variant, so they all belong to it); "k_item_bwd<": the FULL template argument (second)
            # must match the mask
            targs = [a.strip() for a in k[k.find("<") + 1:k.rfind(">")].split(",")] if "<" in k else []
            ok = k.startswith(prefix)
            if ok and prefix.startswith("k_item_bwd"):
                ok = len(targs) > 1 and ((targs[1] == "true") == bool(full_mask))
            if ok and k.startswith("k_spike_classes"):
                ok = False                      # belongs to the kernel build, not to the pass
            if ok and "hbm_read_bytes_raw" in e and "hbm_write_bytes" in e:
                tot += e["hbm_read_bytes_raw"] + e["hbm_write_bytes"]
                insts += e.get("SQ_INSTS_VALU", 0.0)
                n += 1
        if n:
            best = (tot, os.path.basename(f), insts, sha)
    return best


def rank_environments(n, port, base=None):
    """The environment of each of the n ranks of one node, as torch.distributed.run would set it."""
    envs = []
    for r in range(n):
        env = dict(os.environ if base is None else base)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), SVGPFA_BENCH_LAUNCHED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: what RCCL needs on this pool's host driver
        envs.append(env)
    return envs


def launch_ranks(n, argv, script=None, timeout=None):
    """`python bench.py --gpus N` without torch.distributed.run: start N fresh interpreters of this script (one rank per
    GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set), relay what they print (only rank 0 prints the JSON line) and return
    0 only if EVERY rank did and rank 0 printed a line with n_gpus == N.  The launcher itself never imports torch and never
    touches a GPU; children are new processes (no exec of a process that has initialised one).  A rank that fails takes
    the others down with it (they would wait in a collective for ever)."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    script = script or os.path.abspath(__file__)
    procs = []
    for r, env in enumerate(rank_environments(n, port)):
        procs.append(subprocess.Popen([sys.executable, script] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else None, text=True))
    t_end = None if timeout is None else time.monotonic() + timeout
    out0, rcs = None, [None] * n
    try:
        while any(rc is None for rc in rcs):
            for r, p in enumerate(procs):
                if rcs[r] is not None:
                    continue
                if r == 0 and out0 is None:
                    try:
                        out0 = p.communicate(timeout=0.2)[0]
                    except subprocess.TimeoutExpired:
                        pass
                rc = p.poll()
                if rc is not None:
                    rcs[r] = rc
                    if r == 0 and out0 is None:
                        out0 = p.communicate()[0]
                    if rc != 0:
                        raise RuntimeError(f"rank {r} exited with code {rc}")
            if t_end is not None and time.monotonic() > t_end:
                raise RuntimeError(f"ranks did not finish within {timeout} s")
            time.sleep(0.05)
    except (RuntimeError, KeyboardInterrupt) as e:
        for p in procs:                        # exactly the processes started here
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                p.kill()
        if out0:
            sys.stdout.write(out0)
        print(f"[bench] launch of {n} ranks failed: {e}", file=sys.stderr)
        return 1
    sys.stdout.write(out0 or "")
    sys.stdout.flush()
    lines = [ln for ln in (out0 or "").splitlines() if ln.startswith("{")]
    try:
        got = json.loads(lines[-1])["n_gpus"] if lines else None
    except (ValueError, KeyError):
        got = None
    if got != n:
        print(f"[bench] asked for --gpus {n}, rank 0 reported n_gpus = {got}", file=sys.stderr)
        return 1
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="cfg3", choices=["cfg2", "cfg3", "cfg5"])
    ap.add_argument("--variant", default="all", choices=["all", "estep"])
    ap.add_argument("--dtype", default="f64", choices=["f64", "mixed"],
                    help="f64: the reference's arithmetic (headline).  mixed: BASELINE config 5's mode, fp32 spike pass and "
                         "quadrature ELL with fp64 Kzz / Cholesky / whitening / KL")
    ap.add_argument("--cpu-sample-trials", type=int, default=None,
                    help="trials of the workload the CPU oracle is timed on (default: 200 for cfg3, all 50 of cfg2, 12 for "
                         "cfg5: about 10-30 s of CPU work for three repetitions)")
    ap.add_argument("--cpu-reps", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--time-every", type=int, default=8,
                    help="bracket k_item_bwd (and, half a period later, the spike pass) with HIP events in every N-th step of "
                         "the timed region; 1 = in every step, which lengthens a 0.52-ms step by 12 us (tests/tools/timing_cost.py)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="collective backend; gloo + --single-device rehearses the N > 1 path on a one-GPU box")
    ap.add_argument("--single-device", action="store_true", help="every rank uses cuda:0 (rehearsal only)")
    ap.add_argument("--theta-spread", action="store_true",
                    help="kernel parameters as a kernel M-step leaves them (synthetic.spread_theta: every periodic latent "
                         "its own period, RBF lengthscales 0.2 .. 0.9) instead of SURVEY 8(d)'s one value per type: the "
                         "worst case of the spike pass's class path (not the headline)")
    ap.add_argument("--estep-cache", action="store_true",
                    help="E-step variant only: let value-only evaluations reuse the spike sums while C, Z, theta are "
                         "unchanged (svgpfa_set_embedding_token), as the closure of a real E-step does")
    args = ap.parse_args()
    if args.cpu_sample_trials is None:
        args.cpu_sample_trials = {"cfg2": 50, "cfg3": 200, "cfg5": 12}[args.config]

    # --gpus N > 1 without a launcher's environment: this process becomes the launcher (launch_ranks) -- before torch is
    # imported, so the parent never touches a GPU -- and relays rank 0's line.  Under torch.distributed.run (WORLD_SIZE
    # set) the rank count must BE --gpus: a line whose n_gpus differs from what was asked for is refused, not annotated.
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit(f"[bench] --gpus {args.gpus} but WORLD_SIZE={world}: refusing to report a line for a rank count that was "
                 f"not asked for (launch with --nproc-per-node {args.gpus}, or run `python bench.py --gpus {args.gpus}` "
                 f"without a launcher and let it start its own ranks)")

    import torch
    import torch.distributed as dist
    from svgpfa_amd.distributed import allreduce_shared, allreduce_shared_async
    from svgpfa_amd.engine import engine_from_problem
    from svgpfa_amd.utils import synthetic

    if args.single_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or "RANK" in os.environ      # launched by torch.distributed.run (also with 1 rank)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")

    full = synthetic.make_config(args.config, theta_spread=args.theta_spread)
    prob = synthetic.shard_trials(full, rank, world) if world > 1 else full
    eng, P = engine_from_problem(prob, device=local_rank, precision="mixed" if args.dtype == "mixed" else "fp64")
    mask = 15 if args.variant == "all" else 1
    shared = torch.empty(eng.shared_len, dtype=torch.float64, device=dev)
    grads = eng.alloc_grads(mask)

    if args.estep_cache and args.variant == "estep":
        eng.set_embedding_token(1)

    def evaluate(out):
        if args.variant == "all":
            eng.build_kernels(P["Z"], P["theta"])
        eng.elbo(P["mean"], P["cholVecs"], P["C"], P["d"], mask, out=out, grads=grads)

    # Headline step.  N > 1: the evaluation, then the BLOCKING all-reduce of its shared vector -- the compute stream waits
    # for the reduced values, so evaluation i + 1 starts behind the exchange of evaluation i, as in an optimiser loop whose
    # next closure depends on the reduced bound and gradient (svgpfa_amd/stats: the EM closures call the same function).
    def step():
        evaluate(shared)
        if use_dist:
            allreduce_shared(shared)

    def drain():
        pass

    # Supplementary N > 1 loops (not the headline).  host-synced: the host waits for the reduced vector as well.
    def step_host_synced():
        evaluate(shared)
        allreduce_shared(shared)
        torch.cuda.current_stream(dev).synchronize()

    # pipelined: consecutive evaluations write alternating buffers and their all-reduce is enqueued without blocking the
    # compute stream, so the exchange of evaluation i (17.7 KB, latency-bound) runs beside the kernels of evaluation i + 1;
    # a buffer is waited for before it is written again, and all of them before the clock stops.
    bufs = [shared, torch.empty_like(shared)]
    works = [None, None]
    nstep = [0]

    def step_pipelined():
        b = nstep[0] % 2
        nstep[0] += 1
        if works[b] is not None:
            works[b].wait()
            works[b] = None
        evaluate(bufs[b])
        works[b] = allreduce_shared_async(bufs[b])

    def drain_pipelined():
        for b in range(2):
            if works[b] is not None:
                works[b].wait()
                works[b] = None

    def timed_loop(fn, fin):
        for _ in range(max(args.warmup, 3)):
            fn()
        fin()
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            fn()
        fin()
        torch.cuda.synchronize()
        dist.barrier()
        el = time.perf_counter() - t0
        t = torch.tensor([el], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t[0])

    # Python's cyclic collector, left alone, runs a full (generation 2) pass over everything torch's import left on the heap
    # some tens of evaluations into the loop -- 40 ms, 400 steps' worth at cfg2 (tests/tools/step_jitter.py).  The heap as
    # it is now is frozen (never scanned again); what the loop itself allocates stays collectable.
    import gc
    gc.collect()
    gc.freeze()
    # one-off initialisation outside the W + K protocol: first launches load the code objects and create the RCCL
    # communicator channels lazily, and the GPU leaves its idle clocks only after some tens of milliseconds of work
    # (a 20-step run right behind engine set-up measured 0.88 ms per step against 0.84 ms once the clocks had settled);
    # W warm-up steps then run at steady state even when W is small
    for _ in range(30):
        step()
    drain()
    eng.sync()
    for _ in range(args.warmup):
        step()
    drain()
    eng.sync()
    # the dominant kernels' durations are measured live, by HIP events on their stream, in a SAMPLE of the timed steps: an event
    # record is a packet of its own between two kernels, and two pairs in every step cost 12.5 us of a 0.524-ms step
    time_every = max(1, min(args.time_every, args.steps // 4))
    eng.enable_timing(time_every if time_every > 1 else True)
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    drain()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    spike_ms, launches = eng.kernel_ms(0)
    item_ms, item_launches = eng.kernel_ms(1)
    eng.enable_timing(False)
    eng.sync()
    elbo = float(shared[0])
    extra = {}
    if world > 1:
        extra["value_host_synced"] = args.steps / timed_loop(step_host_synced, drain)
        extra["value_pipelined"] = args.steps / timed_loop(step_pipelined, drain_pipelined)

    if use_dist:
        t = torch.tensor([elapsed, spike_ms / max(launches, 1), item_ms / max(item_launches, 1)], dtype=torch.float64,
                         device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, spike_avg_ms, item_avg_ms = float(t[0]), float(t[1]), float(t[2])
    else:
        spike_avg_ms = spike_ms / max(launches, 1)
        item_avg_ms = item_ms / max(item_launches, 1)

    if rank == 0:
        work = algorithmic_work(prob, args.variant)      # this rank's shard (what one launch processes)
        work_full = algorithmic_work(full, args.variant)
        ms_per_step = 1e3 * elapsed / args.steps
        full_pass = bool(mask & 14)
        mixed = args.dtype == "mixed"
        harmonics = eng.series_harmonics()
        cls_flops, pair_flops = spike_pass_flops(prob, full_pass, bool(mask & 2), PAIR_FLOPS_F32 if mixed else PAIR_FLOPS,
                                                 harmonics)
        # the class path is fp64 in both precision modes; only pair-loop items are fp32 in "mixed".  One bound per entry:
        # the side that has the larger share of the pass's operations names it, each side is priced at its own peak.
        secs = spike_avg_ms * 1e-3
        pair_peak = FP32_PEAK_TFLOPS if mixed else FP64_PEAK_TFLOPS
        exec_flops = cls_flops + pair_flops
        ideal = cls_flops / (FP64_PEAK_TFLOPS * 1e12) + pair_flops / (pair_peak * 1e12)      # seconds at the peaks
        spike_bound = "valu_fp32" if (mixed and pair_flops > cls_flops) else "valu_fp64"
        peak = FP32_PEAK_TFLOPS if spike_bound == "valu_fp32" else FP64_PEAK_TFLOPS
        spike_frac = ideal / secs if spike_avg_ms > 0 else None
        achieved = spike_frac * peak if spike_frac is not None else None
        conv = work["flops_spike"] / secs / 1e12 if spike_avg_ms > 0 else None
        traffic = measured_traffic(args.config, args.variant, full_pass) if (world == 1 and not mixed) else None
        item_flops = item_bwd_flops(prob, mask)
        item_ach = item_flops / (item_avg_ms * 1e-3) / 1e12 if item_avg_ms > 0 else None
        tr_item = measured_traffic(args.config, args.variant, bool(mask & 12), "k_item_bwd<") if world == 1 else None
        roof_item = {"bound": "mfma", "achieved": item_ach, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": (item_ach / FP64_PEAK_TFLOPS) if item_ach else None,
                     "traffic": tr_item[0] if tr_item else None,
                     "traffic_source": ("profiles/%s @ csrc %s" % (tr_item[1], tr_item[3])) if tr_item
                     else "none: no committed PMC digest was taken on this device code (csrc %s)" % kernel_source_sha(),
                     "kernel": "k_item_bwd<%d, %s>" % ((max(prob["Ms"]) + 15) // 16, "true" if mask & 12 else "false"),
                     "kernel_avg_ms": item_avg_ms, "kernel_launches": item_launches,
                     "kernel_sampling": "HIP events around the kernel in every %d%s step of the timed region" %
                                        (time_every, {1: "st", 2: "nd", 3: "rd"}.get(time_every, "th")),
                     "executed_fp64_flops_per_launch": item_flops,
                     "note": "per-item adjoints + backward + dC / dd sums: v_mfma_f64_16x16x4_f64 (2 048 flops each, counted "
                             "from the tile structure) plus the kernel evaluations / adjoint accumulations of its "
                             "quadrature phase on the VALU (ITEM_VALU_FLOPS per (node, inducing point)); fp64 MFMA and fp64 "
                             "VALU share one datapath on CDNA4 (their times add), so both are priced against the 78.6 "
                             "TFLOP/s fp64 peak"}
        out = {
            "metric": "ELBO+grad evals/sec at (trials,neurons,latents,ind)=(%d,%d,%d,%d)" % (
                full["R"], full["N"], full["K"], full["Ms"][0]),
            "value": args.steps / elapsed, "unit": "evals/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None,
            "dtype": "f32 (spike pass, quadrature ELL) + f64 (Kzz, Cholesky, whitening, KL, sums)" if mixed else "f64",
            "data": "synthetic",
            "config": {"workload": f"{args.config}: {full['R']} trials x {full['N']} neurons x {full['K']} latents x "
                                   f"{full['Ms'][0]} inducing, {int(full['spike_offsets'][-1])} spikes, Q={full['Q']}",
                       "variant": "ALL (rebuild kernels + ELBO + 4 gradient groups)" if args.variant == "all"
                       else ("E-step (ELBO + variational gradients)"
                                         + (", spike sums reused across evaluations" if args.estep_cache else "")),
                       "kernel_parameters": "spread (every periodic latent its own period, RBF lengthscales 0.2 .. 0.9)"
                       if args.theta_spread else "SURVEY 8(d): one lengthscale / period per kernel type",
                       "parallelism": f"trials sharded {world}-way by spike count, 1 BLOCKING all-reduce ({args.backend}) per "
                                      f"evaluation: the next evaluation starts behind the exchange (value); value_host_synced: the "
                                      f"host waits for it too; value_pipelined: exchange beside the next evaluation's kernels"
                       if world > 1 else "single GPU"},
            "elbo": elbo,
            **extra,
            "roofline": None,
            "roofline_spike": {"bound": spike_bound, "achieved": achieved, "peak": peak,
                         "unit": "TFLOP/s", "frac": spike_frac,
                         "traffic": traffic[0] if traffic else None,
                         "traffic_unit": "HBM bytes per spike pass (FETCH_SIZE + WRITE_SIZE of its k_spike* launches, separate PMC "
                                         "passes; FETCH_SIZE under-counts what the Infinity Cache serves, MI355X_MICROARCH.md)",
                         "traffic_source": ("profiles/%s @ csrc %s" % (traffic[1], traffic[3])) if traffic
                         else "none: no committed PMC digest was taken on this device code (csrc %s)" % kernel_source_sha(),
                         "algorithmic_bytes_per_launch": work.get("bytes_spike"),
                         "kernel": "spike pass = k_spike_cls<4> + k_spike_cls<2> (class path, fp64; its weights W ride in k_front, "
                                   "the moments' combine in the ELL kernel's launch at this size: neither is in this time) "
                                   "+ the pair-by-pair pass (%s)" %
                                   ("k_spike, fp32" if mixed else "a rider of k_spike_cls<2>'s workers"),
                         "kernel_avg_ms": spike_avg_ms, "kernel_launches": launches,
                         "executed_fp64_flops_per_launch": exec_flops,
                         "class_path_flops_per_launch": cls_flops, "pair_loop_flops_per_launch": pair_flops,
                         "series_harmonics": harmonics,
                         "pair_flops": {"%s/%s" % (t[:3], "full" if f else "value"): v
                                        for (t, f), v in (PAIR_FLOPS_F32 if mixed else PAIR_FLOPS).items()},
                         "note": "class path: per (spike, harmonic, CLASS of latents sharing an angular frequency) a rotation + "
                                 "2 adds on the fp64 VALU (8 flops; 12 with time-weighted moments), then per row and member latent "
                                 "the A-moment product on the f64 MFMA and the dC sum; pair-by-pair items (series_harmonics[k] = -1, "
                                 "or trials whose window is too long for the lengthscale): one software 2^y + 3-7 FMAs per (spike, "
                                 "inducing point).  `frac` = time of those operations at the peak of their arithmetic (fp64 78.6, "
                                 "fp32 157.3 TFLOP/s for pair items in --dtype mixed) / hipEvent time of the whole pass; per-spike "
                                 "set-up not counted.  survey_convention_* prices a kernel evaluation at 24 flop-EQUIVALENTS (x3 "
                                 "with gradients) as SURVEY 8(d) does: the class path does not evaluate kernels pair by pair, so "
                                 "that figure is not a hardware fraction and exceeds the peak.",
                         "survey_convention_flops_eq_per_launch": work["flops_spike"],
                         "survey_convention_tflops_eq": conv,
                         "issued_valu_insts_per_launch": traffic[2] if traffic else None,
                         "issue_slot_frac": (2.0 * 64.0 * traffic[2] / secs / 1e12 / FP64_PEAK_TFLOPS)
                         if traffic and spike_avg_ms > 0 else None,
                         "eval_survey_convention_frac_fp64": work_full["flops_eval"] / world / (ms_per_step * 1e-3) / 1e12 / FP64_PEAK_TFLOPS,
                         "eval_frac_hbm": work_full["bytes_eval"] / world / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS},
        }
        # `roofline` = the larger of the two in THIS run (k_item_bwd since the spike pass took the class path); the other
        # one follows as `roofline_second`
        spike_first = spike_avg_ms >= item_avg_ms
        out["roofline"] = out.pop("roofline_spike") if spike_first else roof_item
        out["roofline_second"] = roof_item if spike_first else out.pop("roofline_spike")
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.config, args.variant, args.cpu_sample_trials, args.cpu_reps)
            if args.variant == "all":      # the E-step closure of the same oracle beside it (both variants)
                out["cpu_baseline_estep"] = cpu_baseline(args.config, "estep", args.cpu_sample_trials, args.cpu_reps)
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
