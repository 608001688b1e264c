"""L-BFGS over parameters that are sharded by trial across ranks.

The reference optimises every conditional-maximisation step with ``torch.optim.LBFGS`` on one process
(/root/reference/src/svGPFA/stats/svEM.py:266-294: ``max_iter``, ``lr``, ``tolerance_grad``, ``tolerance_change``,
``line_search_fn="strong_wolfe"`` from the ini file; ``func_evals`` / ``n_iter`` read back from the optimiser state).
With trials sharded over GPUs the per-trial parameters (variational means, cholVecs, inducing-point locations) and
their gradients live on their owner; only the replicated parameters (C, d, kernel parameters) exist everywhere.
This optimiser keeps that placement: every rank updates its own slice and the only thing exchanged is a handful of
inner products.

Design for a latency-bound interconnect (xGMI collectives cost ~tens of microseconds each, far more than the bytes):
the two-loop recursion is carried out in COEFFICIENT space over the basis {s_i, y_i, g} using the Gram entries
s_i.y_j, y_i.y_j, s_i.g, y_i.g, g.g ("vector-free" L-BFGS).  All inner products an iteration needs are local
mat-vec products against the stacked history, packed into ONE small all-gather per iteration (plus one per
line-search evaluation for g_new.d); partial sums are added in rank order, so every rank takes bit-identical
decisions.  Semantics (termination tests, step-length initialisation, history update rule y.s > 1e-10, strong-Wolfe
bracketing / zoom with cubic interpolation, evaluation counting) follow torch.optim.LBFGS, so that with one rank
it is a drop-in for it; tests/test_sharded_lbfgs.py checks both the one-rank equivalence and a 2-rank gloo run.
"""
import math
import os

import numpy as np
import torch
import torch.distributed as dist


def _cubic_min(x1, f1, g1, x2, f2, g2, bounds=None):
    """Minimiser of the cubic through (x1, f1, g1), (x2, f2, g2), clipped to bounds (bisection if no real root)."""
    lo, hi = bounds if bounds is not None else ((x1, x2) if x1 <= x2 else (x2, x1))
    d1 = g1 + g2 - 3.0 * (f1 - f2) / (x1 - x2)
    disc = d1 * d1 - g1 * g2
    if disc < 0:
        return 0.5 * (lo + hi)
    d2 = math.sqrt(disc)
    if x1 <= x2:
        pos = x2 - (x2 - x1) * ((g2 + d2 - d1) / (g2 - g1 + 2.0 * d2))
    else:
        pos = x1 - (x1 - x2) * ((g1 + d2 - d1) / (g1 - g2 + 2.0 * d2))
    return min(max(pos, lo), hi)


def _two_loop(SY, YY, S_g, Y_g, H, gg, m):
    """Two-loop recursion on coefficients over the basis {s_i, y_i, g}: q = -g - sum_j al_j y_j,
    r = H q + sum_j (al_j - be_j) s_j.  -> (al, be, g . d)."""
    al, be = np.zeros(m), np.zeros(m)
    dg = np.diag(SY).copy() if m else np.zeros(0)
    for i in range(m - 1, -1, -1):
        al[i] = (-S_g[i] - al[i + 1:].dot(SY[i, i + 1:])) / dg[i]
    yq = -Y_g - YY.dot(al) if m else np.zeros(0)
    for i in range(m):
        be[i] = (H * yq[i] + (al[:i] - be[:i]).dot(SY[:i, i])) / dg[i]
    gtd = -H * gg
    if m:
        gtd += float(-H * al.dot(Y_g) + (al - be).dot(S_g))
    return al, be, gtd


class _TorchVec:
    """The three fused vector primitives of flat mode in plain tensor operations (CPU tensors: tests, gloo rehearsals).
    Same argument meaning and output layout as the svgpfa_lbfgs_* entry points (include/svgpfa.h)."""

    def __init__(self, x, ns, w0, history_size):
        self._w = None
        if ns and w0 != 1.0:
            self._w = torch.ones_like(x)
            self._w[:ns] = w0

    def grad_stats(self, g_raw, sign, d, value, carry):
        g = g_raw.detach().reshape(-1) * sign
        gw = g if self._w is None else g * self._w
        zero = g.new_zeros(())
        out = torch.stack([gw.dot(d) if d is not None else zero, gw.dot(g), gw.abs().sum(),
                           g.abs().max() if g.numel() else zero,
                           value.detach().reshape(()).to(g.dtype) * sign if value is not None else zero,
                           carry.reshape(()) if carry is not None else zero])
        return g, out

    def pair_dots(self, g, prev_g, d, t, S, Y, m):
        torch.sub(g, prev_g, out=Y[m])
        torch.mul(d, t, out=S[m])
        y, s = Y[m], S[m]
        w = self._w
        yw, sw, gw = (y, s, g) if w is None else (y * w, s * w, g * w)
        head = torch.stack([yw.dot(s), yw.dot(y), gw.dot(g), sw.dot(g), yw.dot(g)])
        if not m:
            return head
        Sm, Ym = S[:m], Y[:m]
        return torch.cat([head, Sm @ yw, Ym @ sw, Ym @ yw, Sm @ gw, Ym @ gw])

    def direction(self, g, S, Y, m, coef, d, x, x0):
        c = torch.from_numpy(np.ascontiguousarray(coef)).to(device=g.device, dtype=g.dtype)
        torch.mul(g, -float(coef[2 * m]), out=d)
        if m:
            d.addmv_(Y[:m].t(), c[:m])
            d.addmv_(S[:m].t(), c[m:2 * m])
        if x0 is not None:
            x0.copy_(x)
        return d.abs().max().reshape(1)


class _HipVec:
    """The same primitives through the fused HIP kernels (svgpfa_amd/csrc/lbfgs_api.hip): fp64 vectors on one GPU."""

    def __init__(self, x, ns, w0, history_size):
        from .. import _lib
        self._lib = _lib.load()
        if x.dtype != torch.float64:
            raise TypeError("the fused optimiser kernels work on float64 vectors")
        self._n, self._ns, self._w0 = x.numel(), int(ns), float(w0)
        self._dev = x.device
        self._hist = 8
        self._work = None
        self._grow(8)
        self._dmax = torch.empty(1, dtype=torch.float64, device=x.device)

    def _grow(self, m):
        if self._work is None or m > self._hist:
            self._hist = max(self._hist, 2 * m)
            nw = int(self._lib.svgpfa_lbfgs_workspace_doubles(self._n, self._hist))
            self._work = torch.empty(nw, dtype=torch.float64, device=self._dev)

    def _stream(self):
        return torch.cuda.current_stream(self._dev).cuda_stream

    @staticmethod
    def _ptr(t):
        if t is None:
            return None
        if not (t.is_cuda and t.dtype == torch.float64 and t.is_contiguous()):
            raise TypeError("the fused optimiser kernels need contiguous float64 device tensors")
        return t.data_ptr()

    def _check(self, rc, what):
        if rc:
            raise RuntimeError(f"{what} failed (status {rc})")

    def grad_stats(self, g_raw, sign, d, value, carry):
        g_raw = g_raw.detach().reshape(-1)
        g = torch.empty_like(g_raw)
        out = torch.empty(6, dtype=torch.float64, device=self._dev)
        if value is None or carry is None:
            out.zero_()
        value = None if value is None else value.detach().reshape(1)
        self._check(self._lib.svgpfa_lbfgs_grad_stats(self._stream(), self._n, self._ptr(g_raw), float(sign), self._ptr(d),
                                                      self._ptr(g), self._ns, self._w0, self._ptr(value),
                                                      self._ptr(carry), self._ptr(self._work), self._ptr(out)),
                    "svgpfa_lbfgs_grad_stats")
        return g, out

    def pair_dots(self, g, prev_g, d, t, S, Y, m):
        self._grow(m)
        out = torch.empty(5 * m + 5, dtype=torch.float64, device=self._dev)
        self._check(self._lib.svgpfa_lbfgs_pair_dots(self._stream(), self._n, self._ptr(g), self._ptr(prev_g), self._ptr(d),
                                                     float(t), S.data_ptr(), Y.data_ptr(), S.stride(0), int(m), self._ns,
                                                     self._w0, self._ptr(self._work), self._ptr(out)),
                    "svgpfa_lbfgs_pair_dots")
        return out

    # ---- the coefficient step on the device (svgpfa_lbfgs_coef): Gram matrices, H and the coefficients never visit the host
    HS1_MAX = 128

    def coef_ready(self, history_size, m):
        """The device-side coefficient step can serve a history of m pairs + the candidate (no pair has to be dropped)."""
        return m + 1 < min(history_size + 1, self.HS1_MAX)

    def coef(self, m, part, history_size):
        """-> (coef [2 (m + 1) + 1] on the device, laid out for a direction over m + 1 rows, out4 = [g.d, pairs, accepted, H])."""
        hs1 = min(history_size + 1, self.HS1_MAX)
        if getattr(self, "_gram", None) is None:
            self._hs1 = hs1
            self._gram = torch.zeros(2 * hs1 * hs1 + 8, dtype=torch.float64, device=self._dev)
            self._coefd = torch.empty(2 * hs1 + 1, dtype=torch.float64, device=self._dev)
            self._out4 = torch.empty(4, dtype=torch.float64, device=self._dev)
        self._check(self._lib.svgpfa_lbfgs_coef(self._stream(), int(m), int(self._hs1), self._ptr(part), self._ptr(self._gram),
                                                self._ptr(self._coefd), self._ptr(self._out4)), "svgpfa_lbfgs_coef")
        return self._coefd, self._out4

    def gram_to_host(self, m):
        """(SY, YY, H) of the m accepted pairs as the host path keeps them (when the device path hands over)."""
        g = self._gram.cpu().numpy()
        hs1 = self._hs1
        SY = g[:hs1 * hs1].reshape(hs1, hs1)[:m, :m].copy()
        YY = g[hs1 * hs1:2 * hs1 * hs1].reshape(hs1, hs1)[:m, :m].copy()
        return SY, YY, float(g[2 * hs1 * hs1])

    def gram_from_host(self, SY, YY, H, history_size):
        """Seed the device state from the host's matrices (the device path takes over an optimiser the host path started)."""
        hs1 = min(history_size + 1, self.HS1_MAX)
        g = np.zeros(2 * hs1 * hs1 + 8)
        m = SY.shape[0]
        a = g[:hs1 * hs1].reshape(hs1, hs1)
        b = g[hs1 * hs1:2 * hs1 * hs1].reshape(hs1, hs1)
        a[:m, :m], b[:m, :m] = SY, YY
        g[2 * hs1 * hs1] = H
        self._hs1 = hs1
        self._gram = torch.from_numpy(g).to(self._dev)
        self._coefd = torch.empty(2 * hs1 + 1, dtype=torch.float64, device=self._dev)
        self._out4 = torch.empty(4, dtype=torch.float64, device=self._dev)

    def direction(self, g, S, Y, m, coef, d, x, x0):
        c = coef if torch.is_tensor(coef) else torch.from_numpy(np.ascontiguousarray(coef, dtype=np.float64)).to(self._dev)
        self._check(self._lib.svgpfa_lbfgs_direction(self._stream(), self._n, self._ptr(g),
                                                     S.data_ptr() if m else None, Y.data_ptr() if m else None,
                                                     S.stride(0) if m else self._n, int(m), self._ptr(c), self._ptr(d),
                                                     self._ptr(x), self._ptr(x0), self._ptr(self._work),
                                                     self._ptr(self._dmax)),
                    "svgpfa_lbfgs_direction")
        return self._dmax


class ShardedLBFGS:
    """``local_params``: tensors this rank owns exclusively.  ``shared_params``: tensors replicated on every rank,
    whose gradients the closure has already summed over ranks (svgpfa_amd.distributed.attach does that).

    Flat mode (``flat=x``): the whole parameter group is ONE contiguous tensor ``x`` (the model keeps each group in one
    buffer and hands out views, SVLowerBound.flatParams) and ``closure()`` returns ``(loss, flat_gradient)`` directly
    (SVLowerBound.evalFlat: no autograd graph, gradients written by the engine straight into a flat buffer);
    ``shared=True`` says the group is replicated on every rank.  ``maximize=True``: the closure returns (value,
    gradient) of an objective to MAXIMISE and the optimiser minimises its negative (the sign change rides on the copy
    of the gradient it has to take anyway).  Arithmetic and decisions are those of list mode; two execution paths,
    chosen per group:

    * vectors on the device (per-trial groups): an iteration's vector work is three fused passes -- gradient copy +
      statistics, history update + all inner products, direction + restart point (``_HipVec``: the
      ``svgpfa_lbfgs_*`` entry points of include/svgpfa.h; ``_TorchVec``: the same primitives in plain tensor
      operations for CPU tensors, i.e. tests and gloo rehearsals) -- and ONE host read-back per evaluation;
    * vectors on the host (``host_max`` doubles or fewer, and the group replicated or the run single-rank: embedding
      and kernel parameters): value and gradient come back in the evaluation's one transfer, the recursion runs in numpy
      and the new parameters go up with one copy.  A replicated group needs no collective at all: every rank sees the
      same all-reduced gradient and takes the same decisions."""

    def __init__(self, local_params=(), shared_params=(), lr=1.0, max_iter=20, max_eval=None, tolerance_grad=1e-7,
                 tolerance_change=1e-9, history_size=100, line_search_fn=None, group=None, flat=None, shared=False,
                 maximize=False, host_max=None, after_sync=None):
        if line_search_fn not in (None, "strong_wolfe"):
            raise RuntimeError("only 'strong_wolfe' is supported")
        if maximize and flat is None:
            raise ValueError("maximize=True needs flat mode")
        self._maximize = bool(maximize)
        # flat mode: called right after every host read-back that follows a closure (the stream is idle then): a closure
        # that defers a device-side status check (SVLowerBound.evalFlat(defer_status=True): the Cholesky of a kernel
        # rebuild) surfaces it here, from inside step(), without a synchronisation of its own
        self._after_sync = after_sync
        self._flat = flat
        if flat is not None:
            if not (flat.dim() == 1 and flat.is_contiguous()):
                raise ValueError("flat parameter group must be a contiguous 1-D tensor")
            self._shared = [flat] if shared else []
            self._local = [] if shared else [flat]
        else:
            self._shared = list(shared_params)
            self._local = list(local_params)
        self._params = self._shared + self._local
        if not self._params:
            raise ValueError("optimizer got an empty parameter list")
        self.lr, self.max_iter = lr, max_iter
        self.max_eval = max_eval if max_eval is not None else max_iter * 5 // 4
        self.tolerance_grad, self.tolerance_change = tolerance_grad, tolerance_change
        self.history_size, self.line_search_fn = history_size, line_search_fn
        self._group = group
        on = dist.is_available() and dist.is_initialized()
        self._world = dist.get_world_size(group) if on else 1
        self._rank = dist.get_rank(group) if on else 0
        self._ns = sum(p.numel() for p in self._shared)
        self._n = sum(p.numel() for p in self._params)
        p0 = self._params[0]
        # weight of each entry in an inner product: replicated entries count once (on rank 0)
        self._w = torch.ones(self._n, dtype=p0.dtype, device=p0.device)
        if self._rank != 0:
            self._w[:self._ns] = 0.0
        self.state = {p0: {"func_evals": 0, "n_iter": 0}}
        self.collectives = 0          # number of all-gathers issued (reported by the tests / bench)
        if host_max is None:
            host_max = int(os.environ.get("SVGPFA_LBFGS_HOST_MAX", "32768"))
        self._host = flat is not None and flat.numel() <= host_max and (shared or self._world == 1)
        self._vec = None

    # ------------------------------------------------------------------ flat views
    def zero_grad(self, set_to_none=True):
        for p in self._params:
            if set_to_none:
                p.grad = None
            elif p.grad is not None:
                p.grad.zero_()

    def _flat_grad(self):
        return torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in self._params])

    def _eval(self, closure):
        """-> (loss tensor, flat gradient as a tensor of its own)."""
        loss = closure()
        return loss, self._flat_grad()

    @torch.no_grad()
    def _add(self, t, d):
        o = 0
        for p in self._params:
            n = p.numel()
            p.add_(d[o:o + n].view_as(p), alpha=t)
            o += n

    def _clone(self):
        return [p.detach().clone() for p in self._params]

    @torch.no_grad()
    def _restore(self, saved):
        for p, s in zip(self._params, saved):
            p.copy_(s)

    def _scalars(self, sums=None, maxes=None, extra=()):
        """Global sums / maxima of local partials plus already-global 0-dim tensors `extra`, as Python floats, with ONE
        device-to-host transfer (and at most one all-gather)."""
        rs, rm = self._reduce(sums, maxes)
        parts = [t.reshape(-1) for t in (rs, rm) if t is not None and t.numel()]
        dev = parts[0].device if parts else None
        for x in extra:
            x = x.detach().reshape(1)
            parts.append(x if dev is None else x.to(dev, parts[0].dtype))
            dev = parts[0].device
        vals = torch.cat(parts).double().cpu().tolist()
        ns = 0 if sums is None else sums.numel()
        nm = 0 if maxes is None else maxes.numel()
        return vals[:ns], vals[ns:ns + nm], vals[ns + nm:]

    # ------------------------------------------------------------------ the only communication
    def _reduce(self, sums=None, maxes=None):
        """Global sums of `sums` and maxima of `maxes` (1-D tensors of local partials) in ONE all-gather;
        partials are combined in rank order on every rank -> identical results everywhere."""
        ns = 0 if sums is None else sums.numel()
        parts = [t for t in (sums, maxes) if t is not None and t.numel()]
        if self._world == 1 or not parts:
            return sums, maxes
        buf = torch.cat(parts)
        if buf.is_cuda and dist.get_backend(self._group) == "gloo":   # CPU rehearsal of the exchange (tests)
            buf = buf.cpu()
        out = [torch.empty_like(buf) for _ in range(self._world)]
        dist.all_gather(out, buf, group=self._group)
        self.collectives += 1
        allp = torch.stack(out)
        rs = allp[:, :ns].sum(dim=0) if ns else sums
        rm = allp[:, ns:].max(dim=0).values if maxes is not None and maxes.numel() else maxes
        return rs, rm

    # ------------------------------------------------------------------ optimisation
    def step(self, closure):
        if self._flat is not None:
            with torch.no_grad():
                return self._step_host(closure) if self._host else self._step_flat(closure)
        return self._step_list(closure)

    def _step_list(self, closure):
        st = self.state[self._params[0]]
        closure = torch.enable_grad()(closure)
        tg, tc = self.tolerance_grad, self.tolerance_change
        w = self._w

        orig_loss, g = self._eval(closure)
        if not torch.is_tensor(orig_loss):
            orig_loss = torch.as_tensor(float(orig_loss), dtype=g.dtype, device=g.device)
        _, (gmax0,), (loss,) = self._scalars(None, g.abs().max().reshape(1), (orig_loss,))
        evals = 1
        st["func_evals"] += 1
        if gmax0 <= tg:
            return orig_loss

        d, t = st.get("d"), st.get("t")
        Sb, Yb, m = st.get("Sb"), st.get("Yb"), st.get("m", 0)      # history rows 0..m-1 (oldest first)
        SY, YY = st.get("SY"), st.get("YY")
        H = st.get("H", 1.0)
        prev_g, prev_loss = st.get("prev_g"), st.get("prev_loss")
        n_iter = 0
        while n_iter < self.max_iter:
            n_iter += 1
            st["n_iter"] += 1
            if st["n_iter"] == 1:
                part = torch.stack([(g * w).dot(g), (g.abs() * w).sum()])
                (gg, gabs), _, _ = self._scalars(part, None)
                d = g.neg()
                m = 0
                SY = YY = np.zeros((0, 0))
                H = 1.0
                gtd = -gg
            else:
                y = g - prev_g
                s = d * t
                gw, yw, sw = g * w, y * w, s * w
                head = torch.stack([yw.dot(s), yw.dot(y), gw.dot(g), sw.dot(g), yw.dot(g)])
                if m:
                    Sm, Ym = Sb[:m], Yb[:m]
                    part = torch.cat([head, Sm @ yw, Ym @ sw, Ym @ yw, Sm @ gw, Ym @ gw])
                else:
                    part = head
                tot, _ = self._reduce(part, None)
                tot = tot.double().cpu().numpy()                      # the one host sync of the iteration
                ys, yy, gg, sg, yg = (float(v) for v in tot[:5])
                S_y, s_Y, Y_y, S_g, Y_g = (tot[5 + i * m:5 + (i + 1) * m] for i in range(5))
                if ys > 1e-10:
                    if m == self.history_size:       # drop the oldest pair
                        Sb[:m - 1] = Sb[1:m].clone()
                        Yb[:m - 1] = Yb[1:m].clone()
                        SY, YY = SY[1:, 1:], YY[1:, 1:]
                        S_y, s_Y, Y_y, S_g, Y_g = S_y[1:], s_Y[1:], Y_y[1:], S_g[1:], Y_g[1:]
                        m -= 1
                    if Sb is None or m == Sb.shape[0]:               # grow the history buffers geometrically
                        cap = min(self.history_size, max(8, 2 * m))
                        nS, nY = g.new_empty(cap, g.numel()), g.new_empty(cap, g.numel())
                        if m:
                            nS[:m], nY[:m] = Sb[:m], Yb[:m]
                        Sb, Yb = nS, nY
                    Sb[m], Yb[m] = s, y
                    nSY, nYY = np.zeros((m + 1, m + 1)), np.zeros((m + 1, m + 1))
                    nSY[:m, :m], nYY[:m, :m] = SY, YY
                    nSY[:m, m], nSY[m, :m], nSY[m, m] = S_y, s_Y, ys       # SY[i, j] = s_i . y_j
                    nYY[:m, m], nYY[m, :m], nYY[m, m] = Y_y, Y_y, yy
                    SY, YY = nSY, nYY
                    S_g, Y_g = np.append(S_g, sg), np.append(Y_g, yg)
                    H = ys / yy
                    m += 1
                al, be, gtd = _two_loop(SY, YY, S_g, Y_g, H, gg, m)
                d = g.mul(-H)
                if m:
                    coef = torch.from_numpy(np.concatenate([-H * al, al - be])).to(device=g.device, dtype=g.dtype)
                    d.addmv_(Yb[:m].t(), coef[:m])
                    d.addmv_(Sb[:m].t(), coef[m:])
            prev_g = g.clone()
            prev_loss = loss
            t = min(1.0, 1.0 / gabs) * self.lr if st["n_iter"] == 1 else self.lr
            if gtd > -tc:
                break

            ls_evals = 0
            if self.line_search_fn == "strong_wolfe":
                x0 = self._clone()

                dabs = d.abs().max()

                def phi(step):    # -> (loss, flat gradient, g . d, max |g|, max |d|) at x0 + step d; ONE host sync
                    self._add(step, d)
                    f, gn = self._eval(closure)
                    self._restore(x0)
                    (gd,), (gm, dm), (fv,) = self._scalars(((gn * w).dot(d)).reshape(1),
                                                           torch.stack([gn.abs().max(), dabs]), (f,))
                    return fv, gn, gd, gm, dm
                loss, g, t, ls_evals, gmax, dmax = self._strong_wolfe(phi, t, d, loss, g, gtd)
                self._add(t, d)
            else:
                self._add(t, d)
                gmax = math.inf
                if n_iter != self.max_iter:
                    f, g = self._eval(closure)
                    ls_evals = 1
                    _, (gmax, dmax), (loss,) = self._scalars(None, torch.stack([g.abs().max(), d.abs().max()]), (f,))
                else:
                    _, (dmax,), _ = self._scalars(None, d.abs().max().reshape(1))
            evals += ls_evals
            st["func_evals"] += ls_evals

            if n_iter == self.max_iter or evals >= self.max_eval:
                break
            if gmax <= tg:
                break
            if dmax * abs(t) <= tc:
                break
            if abs(loss - prev_loss) < tc:
                break

        st.update(d=d, t=t, Sb=Sb, Yb=Yb, m=m, SY=SY, YY=YY, H=H, prev_g=prev_g, prev_loss=prev_loss)
        return orig_loss

    # ------------------------------------------------------------------ flat mode, vectors on the device
    def _read(self, out, nsum, maxes=(), extra=()):
        """Host copy of a small device vector `out` whose first nsum entries are local partial sums, entries `maxes`
        local partial maxima and entries `extra` already global: ONE transfer (and at most one all-gather)."""
        if self._world == 1:
            vals = out.double().cpu().numpy()
            if self._after_sync is not None:
                self._after_sync()
            return vals
        idx = list(range(nsum)) + list(maxes)
        loc = out[idx] if len(idx) != out.numel() else out
        rs, rm = self._reduce(loc[:nsum], loc[nsum:] if maxes else None)
        vals = out.double().cpu().numpy().copy()
        if nsum:
            vals[:nsum] = rs.double().cpu().numpy()
        if maxes:
            vals[list(maxes)] = rm.double().cpu().numpy()
        if self._after_sync is not None:
            self._after_sync()
        return vals

    def _vectors(self):
        if self._vec is None:
            x = self._flat
            ns = x.numel() if self._shared else 0
            w0 = 1.0 if self._rank == 0 else 0.0
            cls = _HipVec if x.is_cuda else _TorchVec
            self._vec = cls(x, ns, w0, self.history_size)
        return self._vec

    def _step_flat(self, closure):
        """One optimiser step over the flat group with the vectors on the device; same control flow as _step_list."""
        st = self.state[self._params[0]]
        tg, tc = self.tolerance_grad, self.tolerance_change
        vec, x = self._vectors(), self._flat
        sign = -1.0 if self._maximize else 1.0

        val, graw = closure()
        g, out = vec.grad_stats(graw, sign, None, val, None)
        v = self._read(out[:5], 3, maxes=(3,))
        gg, gabs, gmax, loss = float(v[1]), float(v[2]), float(v[3]), float(v[4])
        orig_loss = loss
        evals = 1
        st["func_evals"] += 1
        if gmax <= tg:
            return torch.as_tensor(orig_loss)

        d, t = st.get("d"), st.get("t")
        # History: a window of m rows (oldest first) that SLIDES through buffers of up to 2 history_size + 1 rows, the
        # candidate pair behind it.  Dropping the oldest pair moves the window's start by one row; only when the window
        # reaches the end of the buffer is it copied back to the front -- once per history_size iterations instead of a
        # shift of the whole history (2 m n doubles moved plus as many of temporary) every iteration.
        Sb, Yb, m, w0 = st.get("Sb"), st.get("Yb"), st.get("m", 0), st.get("w0", 0)
        SY, YY = st.get("SY"), st.get("YY")
        H = st.get("H", 1.0)
        prev_g, prev_loss = st.get("prev_g"), st.get("prev_loss")
        x0 = st.get("x0")
        if d is None:
            d, x0 = torch.empty_like(x), torch.empty_like(x)
        # The coefficient step on the device (one host read-back per iteration instead of two): fp64 vectors on one GPU, one
        # rank (across ranks the inner products are summed over the ranks on the way: the host path), SVGPFA_LBFGS_DEVCOEF=0: off
        devcoef = isinstance(vec, _HipVec) and self._world == 1 and os.environ.get("SVGPFA_LBFGS_DEVCOEF", "1") != "0"
        n_iter = 0
        while n_iter < self.max_iter:
            n_iter += 1
            st["n_iter"] += 1
            if st["n_iter"] == 1:
                m = w0 = 0
                SY = YY = np.zeros((0, 0))
                H = 1.0
                gtd = -gg
                coef = np.array([H])
                st["gram_on_device"] = False
            else:
                if Sb is None or w0 + m + 1 > Sb.shape[0]:
                    full = 2 * self.history_size + 1
                    if Sb is None or Sb.shape[0] < full:              # grow the history buffers geometrically
                        cap = min(full, max(8, 2 * (m + 1)))
                        nS, nY = x.new_empty(cap, x.numel()), x.new_empty(cap, x.numel())
                        if m:
                            nS[:m], nY[:m] = Sb[w0:w0 + m], Yb[w0:w0 + m]
                        Sb, Yb = nS, nY
                    else:                                             # window at the end: back to the front (w0 >= m here)
                        Sb[:m], Yb[:m] = Sb[w0:w0 + m], Yb[w0:w0 + m]
                    w0 = 0
                Sw, Yw = Sb[w0:], Yb[w0:]
                part = vec.pair_dots(g, prev_g, d, t, Sw, Yw, m)      # writes the candidate pair into the window's row m
                dev_it = (devcoef and self.line_search_fn == "strong_wolfe" and vec.coef_ready(self.history_size, m))
                if devcoef and not dev_it and st.get("gram_on_device"):      # the device path hands over (history full)
                    SY, YY, H = vec.gram_to_host(m)
                    st["gram_on_device"] = False
                if dev_it:
                    # The coefficient step on the device: nothing is read back here.  g.d, the number of pairs and H come
                    # with the read-back of the first trial evaluation below.
                    if not st.get("gram_on_device"):
                        vec.gram_from_host(SY if SY is not None else np.zeros((0, 0)), YY if YY is not None else np.zeros((0, 0)),
                                           H, self.history_size)
                        st["gram_on_device"] = True
                    coef, pend = vec.coef(m, part, self.history_size)
                    dmax_t = vec.direction(g, Sw, Yw, m + 1, coef, d, x, x0)
                    prev_g, prev_loss = g, loss
                    t = self.lr
                    torch.add(x0, d, alpha=t, out=x)
                    f, graw = closure()
                    gn, o6 = vec.grad_stats(graw, sign, d, f, dmax_t)
                    v = self._read(torch.cat([o6, pend]), 3, maxes=(3, 5))
                    gtd, m, H = float(v[6]), int(round(float(v[7]))), float(v[9])
                    if gtd > -tc:                  # torch stops in front of the line search: the trial evaluation is dropped
                        x.copy_(x0)
                        break
                    first = (float(v[4]), gn, float(v[0]), float(v[3]), float(v[5]))
                    step0, used, last = t, [False], [t]

                    def phi(step):    # the first trial has been evaluated already; further ones as in the host path
                        if not used[0] and step == step0:
                            used[0] = True
                            return first
                        torch.add(x0, d, alpha=step, out=x)
                        last[0] = step
                        fv, gr = closure()
                        gnew, o = vec.grad_stats(gr, sign, d, fv, dmax_t)
                        w = self._read(o, 3, maxes=(3, 5))
                        return float(w[4]), gnew, float(w[0]), float(w[3]), float(w[5])
                    loss, g, t, ls_evals, gmax, dmax = self._strong_wolfe(phi, t, d, loss, g, gtd, gm0=gmax)
                    if last[0] != t:
                        torch.add(x0, d, alpha=t, out=x)
                    evals += ls_evals
                    st["func_evals"] += ls_evals
                    if n_iter == self.max_iter or evals >= self.max_eval:
                        break
                    if gmax <= tg:
                        break
                    if dmax * abs(t) <= tc:
                        break
                    if abs(loss - prev_loss) < tc:
                        break
                    continue
                tot = self._read(part, part.numel())                  # the one host sync of the iteration
                ys, yy, gg, sg, yg = (float(v) for v in tot[:5])
                S_y, s_Y, Y_y, S_g, Y_g = (tot[5 + i * m:5 + (i + 1) * m] for i in range(5))
                if ys > 1e-10:
                    if m == self.history_size:       # drop the oldest pair: the window starts one row later
                        w0 += 1
                        SY, YY = SY[1:, 1:], YY[1:, 1:]
                        S_y, s_Y, Y_y, S_g, Y_g = S_y[1:], s_Y[1:], Y_y[1:], S_g[1:], Y_g[1:]
                        m -= 1
                    nSY, nYY = np.zeros((m + 1, m + 1)), np.zeros((m + 1, m + 1))
                    nSY[:m, :m], nYY[:m, :m] = SY, YY
                    nSY[:m, m], nSY[m, :m], nSY[m, m] = S_y, s_Y, ys       # SY[i, j] = s_i . y_j
                    nYY[:m, m], nYY[m, :m], nYY[m, m] = Y_y, Y_y, yy
                    SY, YY = nSY, nYY
                    S_g, Y_g = np.append(S_g, sg), np.append(Y_g, yg)
                    H = ys / yy
                    m += 1
                al, be, gtd = _two_loop(SY, YY, S_g, Y_g, H, gg, m)
                coef = np.concatenate([-H * al, al - be, [H]])
            dmax_t = vec.direction(g, Sb[w0:] if Sb is not None else None, Yb[w0:] if Yb is not None else None, m, coef, d,
                                   x, x0)                                  # d, x0 = x, max |d| (stays on the device)
            prev_g = g
            prev_loss = loss
            t = min(1.0, 1.0 / gabs) * self.lr if st["n_iter"] == 1 else self.lr
            if gtd > -tc:
                break

            ls_evals = 0
            if self.line_search_fn == "strong_wolfe":
                last = [None]

                def phi(step):    # -> (loss, gradient, g . d, max |g|, max |d|) at x0 + step d; ONE host sync
                    torch.add(x0, d, alpha=step, out=x)
                    last[0] = step
                    f, graw = closure()
                    gn, o6 = vec.grad_stats(graw, sign, d, f, dmax_t)
                    v = self._read(o6, 3, maxes=(3, 5))
                    return float(v[4]), gn, float(v[0]), float(v[3]), float(v[5])
                loss, g, t, ls_evals, gmax, dmax = self._strong_wolfe(phi, t, d, loss, g, gtd, gm0=gmax)
                if last[0] != t:
                    torch.add(x0, d, alpha=t, out=x)
            else:
                x.add_(d, alpha=t)
                if n_iter != self.max_iter:
                    f, graw = closure()
                    g, o6 = vec.grad_stats(graw, sign, None, f, dmax_t)
                    ls_evals = 1
                    v = self._read(o6, 3, maxes=(3, 5))
                    loss, gmax, dmax = float(v[4]), float(v[3]), float(v[5])
                else:
                    gmax = math.inf
                    dmax = float(self._read(dmax_t, 0, maxes=(0,))[0])
            evals += ls_evals
            st["func_evals"] += ls_evals

            if n_iter == self.max_iter or evals >= self.max_eval:
                break
            if gmax <= tg:
                break
            if dmax * abs(t) <= tc:
                break
            if abs(loss - prev_loss) < tc:
                break

        st.update(d=d, t=t, Sb=Sb, Yb=Yb, m=m, w0=w0, SY=SY, YY=YY, H=H, prev_g=prev_g, prev_loss=prev_loss, x0=x0)
        return torch.as_tensor(orig_loss)

    # ------------------------------------------------------------------ flat mode, vectors on the host
    def _step_host(self, closure):
        """One optimiser step over a small flat group with the vectors in host memory (numpy): the evaluation's one
        transfer brings value and gradient, the new parameters go up with one copy.  torch.optim.LBFGS's own two-loop
        recursion, operation for operation."""
        st = self.state[self._params[0]]
        tg, tc = self.tolerance_grad, self.tolerance_change
        xdev = self._flat
        n = xdev.numel()
        sign = -1.0 if self._maximize else 1.0
        cuda = xdev.is_cuda
        if "host" not in st:
            pin = dict(pin_memory=True) if cuda else {}
            st["host"] = (torch.empty(n + 1, dtype=xdev.dtype, **pin), torch.empty(n, dtype=xdev.dtype, **pin))
        rbuf, xbuf = st["host"]
        pending = [False]

        def evaluate():          # -> (loss, gradient) as (float, fresh array): the ONE synchronisation of an evaluation
            val, graw = closure()
            pack = torch.cat([val.detach().reshape(1).to(graw.dtype), graw.detach().reshape(-1)])
            if cuda:
                rbuf.copy_(pack, non_blocking=True)
                torch.cuda.current_stream(xdev.device).synchronize()
            else:
                rbuf.copy_(pack)
            pending[0] = False
            if self._after_sync is not None:
                self._after_sync()
            a = rbuf.numpy()
            return sign * float(a[0]), sign * a[1:]

        def upload(xnew):
            if pending[0] and cuda:      # the pinned buffer may still be travelling
                torch.cuda.current_stream(xdev.device).synchronize()
            xbuf.numpy()[:] = xnew
            xdev.copy_(xbuf, non_blocking=True)
            pending[0] = True

        xcur = xdev.detach().cpu().numpy().astype(np.float64, copy=True)
        loss, g = evaluate()
        orig_loss = loss
        evals = 1
        st["func_evals"] += 1
        gmax = float(np.abs(g).max()) if n else 0.0
        if gmax <= tg:
            return torch.as_tensor(orig_loss)

        d, t = st.get("d"), st.get("t")
        old_dirs, old_stps, ro = st.get("old_dirs", []), st.get("old_stps", []), st.get("ro", [])
        H = st.get("H", 1.0)
        prev_g, prev_loss = st.get("prev_g"), st.get("prev_loss")
        n_iter = 0
        while n_iter < self.max_iter:
            n_iter += 1
            st["n_iter"] += 1
            if st["n_iter"] == 1:
                d = -g
                old_dirs, old_stps, ro = [], [], []
                H = 1.0
            else:
                y = g - prev_g
                s = d * t
                ys = float(y.dot(s))
                if ys > 1e-10:
                    if len(old_dirs) == self.history_size:
                        old_dirs.pop(0)
                        old_stps.pop(0)
                        ro.pop(0)
                    old_dirs.append(y)
                    old_stps.append(s)
                    ro.append(1.0 / ys)
                    H = ys / float(y.dot(y))
                k = len(old_dirs)
                al = [0.0] * k
                q = -g
                for i in range(k - 1, -1, -1):
                    al[i] = float(old_stps[i].dot(q)) * ro[i]
                    q = q - al[i] * old_dirs[i]
                d = q * H
                for i in range(k):
                    be_i = float(old_dirs[i].dot(d)) * ro[i]
                    d = d + old_stps[i] * (al[i] - be_i)
            prev_g = g
            prev_loss = loss
            t = min(1.0, 1.0 / float(np.abs(g).sum())) * self.lr if st["n_iter"] == 1 else self.lr
            gtd = float(g.dot(d))
            if gtd > -tc:
                break

            ls_evals = 0
            dmax = float(np.abs(d).max())
            if self.line_search_fn == "strong_wolfe":
                x0 = xcur
                last = [None]

                def phi(step):
                    upload(x0 + step * d)
                    last[0] = step
                    f, gn = evaluate()
                    return f, gn, float(gn.dot(d)), float(np.abs(gn).max()), dmax
                loss, g, t, ls_evals, gmax, dmax = self._strong_wolfe(phi, t, d, loss, g, gtd, gm0=gmax)
                xcur = x0 + t * d
                if last[0] != t:
                    upload(xcur)
            else:
                xcur = xcur + t * d
                upload(xcur)
                gmax = math.inf
                if n_iter != self.max_iter:
                    loss, g = evaluate()
                    ls_evals = 1
                    gmax = float(np.abs(g).max())
            evals += ls_evals
            st["func_evals"] += ls_evals

            if n_iter == self.max_iter or evals >= self.max_eval:
                break
            if gmax <= tg:
                break
            if dmax * abs(t) <= tc:
                break
            if abs(loss - prev_loss) < tc:
                break

        if pending[0] and cuda:          # the pinned buffer belongs to this optimiser: leave nothing in flight
            torch.cuda.current_stream(xdev.device).synchronize()
        st.update(d=d, t=t, old_dirs=old_dirs, old_stps=old_stps, ro=ro, H=H, prev_g=prev_g, prev_loss=prev_loss)
        return torch.as_tensor(orig_loss)

    def _strong_wolfe(self, phi, t, d, f, g, gtd, c1=1e-4, c2=0.9, max_ls=25, gm0=None):
        """Bracketing + zoom line search for the strong Wolfe conditions; phi(t) -> (loss, flat grad, grad . d,
        max |grad|, max |d|); gm0 = max |g| at the starting point when the caller knows it.  Returns (loss, grad, step,
        evaluations, max |grad| at the step, max |d|)."""
        tc = self.tolerance_change
        f_new, g_new, gtd_new, gm_new, d_norm = phi(t)
        n_eval = 1
        t_prev, f_prev, g_prev, gtd_prev, gm_prev = 0.0, f, g, gtd, gm0
        done = False
        it = 0
        br = None
        while it < max_ls:
            if f_new > f + c1 * t * gtd or (it > 1 and f_new >= f_prev) or (gtd_new >= 0 and abs(gtd_new) > -c2 * gtd):
                br = [[t_prev, f_prev, g_prev, gtd_prev, gm_prev], [t, f_new, g_new, gtd_new, gm_new]]
                break
            if abs(gtd_new) <= -c2 * gtd:
                br = [[t, f_new, g_new, gtd_new, gm_new]]
                done = True
                break
            lo_step, hi_step = t + 0.01 * (t - t_prev), t * 10.0
            t_next = _cubic_min(t_prev, f_prev, gtd_prev, t, f_new, gtd_new, bounds=(lo_step, hi_step))
            t_prev, f_prev, g_prev, gtd_prev, gm_prev = t, f_new, g_new, gtd_new, gm_new
            t = t_next
            f_new, g_new, gtd_new, gm_new, _ = phi(t)
            n_eval += 1
            it += 1
        if br is None:                      # ran out of bracketing iterations
            br = [[0.0, f, g, gtd, gm0], [t, f_new, g_new, gtd_new, gm_new]]

        insufficient = False
        lo, hi = (0, 1) if br[0][1] <= br[-1][1] else (1, 0)
        while not done and it < max_ls:
            if abs(br[1][0] - br[0][0]) * d_norm < tc:
                break
            t = _cubic_min(br[0][0], br[0][1], br[0][3], br[1][0], br[1][1], br[1][3])
            bmax, bmin = max(br[0][0], br[1][0]), min(br[0][0], br[1][0])
            eps = 0.1 * (bmax - bmin)
            if min(bmax - t, t - bmin) < eps:
                if insufficient or t >= bmax or t <= bmin:
                    t = bmax - eps if abs(t - bmax) < abs(t - bmin) else bmin + eps
                    insufficient = False
                else:
                    insufficient = True
            else:
                insufficient = False
            f_new, g_new, gtd_new, gm_new, _ = phi(t)
            n_eval += 1
            it += 1
            if f_new > f + c1 * t * gtd or f_new >= br[lo][1]:
                br[hi] = [t, f_new, g_new, gtd_new, gm_new]
                lo, hi = (0, 1) if br[0][1] <= br[1][1] else (1, 0)
            else:
                if abs(gtd_new) <= -c2 * gtd:
                    done = True
                elif gtd_new * (br[hi][0] - br[lo][0]) >= 0:
                    br[hi] = list(br[lo])
                br[lo] = [t, f_new, g_new, gtd_new, gm_new]
        if len(br) == 1:
            lo = 0
        t, f_new, g_new, gm = br[lo][0], br[lo][1], br[lo][2], br[lo][4]
        if gm is None:                      # the search fell back to the starting point: its gradient is the caller's g
            _, (gm,), _ = self._scalars(None, torch.as_tensor(g_new).abs().max().reshape(1))
        return f_new, g_new, t, n_eval, gm, d_norm
