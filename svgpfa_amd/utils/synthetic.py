"""Seeded synthetic svGPFA workloads shared by bench.py and tests.

Host-side numpy only: returns plain numpy arrays so that the same inputs can be fed
to the HIP engine (through torch device tensors) and to the CPU oracle.

Spikes are packed in the layout the engine consumes (CSR over trials, neuron-major
inside a trial -- the stacking order of the reference,
/root/reference/src/svGPFA/stats/expectedLogLikelihood.py:157-173).
"""
import math

import numpy as np

RBF = "exponentialQuadratic"
PERIODIC = "periodic"

# BASELINE.json configs 2, 3, 5 (config 4 = config 3 sharded by trial)
CONFIGS = {
    "cfg2": dict(index=2, R=50, N=50, K=5, M=16, kernels="rbf"),
    "cfg3": dict(index=3, R=500, N=200, K=10, M=32, kernels="mix"),
    "cfg5": dict(index=5, R=2000, N=500, K=20, M=64, kernels="mix"),
}


def leg_quad(n_quad, starts, ends):
    """Gauss-Legendre nodes/weights mapped to [start_r, end_r] per trial: R x Q each.

    Mirrors utils/miscUtils.py:226-236 (whose gcnu_common.leggaussVarLimits is
    numpy.polynomial.legendre.leggauss mapped affinely)."""
    x, w = np.polynomial.legendre.leggauss(n_quad)
    starts = np.asarray(starts, dtype=np.float64)[:, None]
    ends = np.asarray(ends, dtype=np.float64)[:, None]
    pts = (starts * (1 - x[None, :]) + ends * (1 + x[None, :])) / 2
    wts = (ends - starts) / 2 * w[None, :]
    return pts, wts


def tril_vec(L):
    """Row-major lower-triangle vectorisation (torch.tril_indices order)."""
    M = L.shape[-1]
    i, j = np.tril_indices(M)
    return np.ascontiguousarray(L[..., i, j])


def make_problem(R, N, K, M, kernels="rbf", seed=0, n_quad=200, rate=20.0, reg=1e-3,
                 jitter=0.0, trial_len=1.0):
    """Build one synthetic problem.  `M` may be an int or a list of K ints.

    jitter = 0 gives exactly the SURVEY 8(d) benchmark inputs (equidistant Z,
    S = 1e-2 I); jitter > 0 additionally perturbs Z, fills the strict lower triangle
    of the q(u) Cholesky factors and perturbs kernel params, so that gradient parity
    tests are not run at a symmetric point.
    """
    rng = np.random.default_rng(seed)
    Ms = [M] * K if np.isscalar(M) else list(M)
    assert len(Ms) == K
    if kernels == "rbf":
        ktypes = [RBF] * K
    elif kernels == "periodic":
        ktypes = [PERIODIC] * K
    elif kernels == "mix":
        ktypes = [RBF if k % 2 == 0 else PERIODIC for k in range(K)]
    else:
        ktypes = list(kernels)

    counts = rng.poisson(rate * trial_len, size=(R, N)).astype(np.int64)
    total = int(counts.sum())
    seg = np.repeat(np.arange(R * N), counts.reshape(-1))
    times = rng.random(total) * trial_len
    order = np.lexsort((times, seg))
    times = times[order]
    neuron = (seg % N).astype(np.int32)
    per_trial = counts.sum(axis=1)
    offsets = np.zeros(R + 1, dtype=np.int64)
    np.cumsum(per_trial, out=offsets[1:])

    pts, wts = leg_quad(n_quad, np.zeros(R), np.full(R, trial_len))

    Z, mean, chol, theta = [], [], [], []
    for k in range(K):
        Mk = Ms[k]
        z = np.tile(np.linspace(0.0, trial_len, Mk)[None, :], (R, 1))
        if jitter:
            z = z + jitter * (trial_len / max(Mk - 1, 1)) * rng.uniform(-0.4, 0.4, size=z.shape)
        Z.append(z)
        mean.append(0.1 * rng.standard_normal((R, Mk)))
        L = np.tile(0.1 * np.eye(Mk)[None], (R, 1, 1))
        if jitter:
            L = L + jitter * 0.03 * np.tril(rng.standard_normal((R, Mk, Mk)))
        chol.append(tril_vec(L))
        if ktypes[k] == RBF:
            th = np.array([0.3])
        else:
            th = np.array([1.5, 0.4])
        if jitter:
            th = th * (1 + jitter * rng.uniform(-0.1, 0.1, size=th.shape))
        theta.append(th)
    C = 0.3 * rng.standard_normal((N, K))
    d = np.full(N, math.log(rate))
    if jitter:
        d = d + jitter * 0.1 * rng.standard_normal(N)

    return dict(R=R, N=N, K=K, Ms=Ms, Q=n_quad, kernel_types=ktypes, reg=reg,
                spike_offsets=offsets, spike_times=times, spike_neuron=neuron,
                quad_points=pts, quad_weights=wts,
                Z=Z, mean=mean, cholVecs=chol, theta=theta, C=C, d=d)


def spread_theta(prob):
    """Kernel parameters as they look AFTER a kernel M-step of the reference's EM loop, which optimises one [lengthscale,
    period] per latent independently (stats/svEM.py:234-254, stats/kernels.py:73-85): no two periodic latents share a
    period any more (0.40 (1 + 0.03 k)) and the RBF lengthscales spread over 0.2 .. 0.9.  SURVEY 8(d)'s benchmark inputs
    give every latent of a type the same parameters -- the best case of the spike pass's class path, whose classes are
    latents with EQUAL angular frequency; this is the matching worst case (bench.py --theta-spread)."""
    out = dict(prob)
    rbf = [k for k, t in enumerate(prob["kernel_types"]) if t == RBF]
    theta = []
    for k, t in enumerate(prob["kernel_types"]):
        if t == RBF:
            i = rbf.index(k)
            theta.append(np.array([0.2 + 0.7 * i / max(len(rbf) - 1, 1)]))
        else:
            theta.append(np.array([1.5 * (1.0 + 0.02 * k), 0.40 * (1.0 + 0.03 * k)]))
    out["theta"] = theta
    return out


def make_config(name, n_trials=None, theta_spread=False, **over):
    """Problem for a BASELINE config name ("cfg2" | "cfg3" | "cfg5"); seed per SURVEY 8(d)."""
    c = dict(CONFIGS[name])
    idx = c.pop("index")
    if n_trials is not None:
        c["R"] = n_trials
    c.update(over)
    prob = make_problem(seed=20260103 + idx, **c)
    return spread_theta(prob) if theta_spread else prob


def slice_trials(prob, lo, hi):
    """Trials [lo, hi) of a problem as a problem of its own (CSR arrays re-based)."""
    off = prob["spike_offsets"]
    s0, s1 = off[lo], off[hi]
    out = dict(prob)
    out["R"] = hi - lo
    out["spike_offsets"] = off[lo:hi + 1] - s0
    out["spike_times"] = prob["spike_times"][s0:s1]
    out["spike_neuron"] = prob["spike_neuron"][s0:s1]
    for key in ("quad_points", "quad_weights"):
        out[key] = prob[key][lo:hi]
    for key in ("Z", "mean", "cholVecs"):
        out[key] = [a[lo:hi] for a in prob[key]]
    out["trial_range"] = (lo, hi)
    return out


def shard_bounds(prob, world):
    """Trial boundaries of a `world`-way contiguous partition balanced by spike count.
    Every rank gets at least one trial; fewer trials than ranks is an error on EVERY rank (the same
    arguments give the same exception everywhere, so no rank is left waiting in a collective)."""
    R = prob["R"]
    if world < 1:
        raise ValueError("world must be positive")
    if R < world:
        raise ValueError(f"cannot shard {R} trials over {world} ranks: every rank needs at least one trial")
    off = prob["spike_offsets"]
    total = off[-1]
    # boundaries at equal spike-count quantiles, snapped to trial edges
    targets = [total * i / world for i in range(world + 1)]
    bounds = [int(np.searchsorted(off, t, side="left")) for t in targets]
    bounds[0], bounds[-1] = 0, R
    for i in range(1, world + 1):          # keep boundaries monotone and shards non-empty
        bounds[i] = max(bounds[i], bounds[i - 1] + 1)
    bounds[-1] = R
    for i in range(world - 1, 0, -1):
        bounds[i] = min(bounds[i], bounds[i + 1] - 1)
    return bounds


def shard_trials(prob, rank, world):
    """Contiguous block of trials for `rank`, balanced by spike count."""
    bounds = shard_bounds(prob, world)
    return slice_trials(prob, bounds[rank], bounds[rank + 1])
