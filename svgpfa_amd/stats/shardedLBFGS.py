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


class ShardedLBFGS:
    """``local_params``: tensors this rank owns exclusively.  ``shared_params``: tensors replicated on every rank,
    whose gradients the closure has already summed over ranks (svgpfa_amd.distributed.attach does that).

    Flat mode (``flat=x``): the whole parameter group is ONE contiguous tensor ``x`` (the model keeps each group in one
    buffer and hands out views, SVLowerBound.flatParams) and ``closure()`` returns ``(loss, flat_gradient)`` directly
    (SVLowerBound.evalFlat: no autograd graph, gradients written by the engine straight into a flat buffer).  A trial
    step is then one axpy and a line-search rollback one copy, instead of one small kernel per parameter tensor (4 K + 2
    tensors at the north-star size: ~100 launches per closure); ``shared=True`` says the group is replicated on every
    rank.  Arithmetic and decisions are the same as in list mode."""

    def __init__(self, local_params=(), shared_params=(), lr=1.0, max_iter=20, max_eval=None, tolerance_grad=1e-7,
                 tolerance_change=1e-9, history_size=100, line_search_fn=None, group=None, flat=None, shared=False):
        if line_search_fn not in (None, "strong_wolfe"):
            raise RuntimeError("only 'strong_wolfe' is supported")
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
        if self._flat is not None:
            loss, g = closure()
            return loss, g
        loss = closure()
        return loss, self._flat_grad()

    @torch.no_grad()
    def _add(self, t, d):
        if self._flat is not None:
            self._flat.add_(d, alpha=t)
            return
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
                # two-loop recursion on coefficients: q = -g - sum_j al_j y_j ; r = H q + sum_j (al_j - be_j) s_j
                al, be = np.zeros(m), np.zeros(m)
                dg = np.diag(SY).copy() if m else np.zeros(0)
                for i in range(m - 1, -1, -1):
                    al[i] = (-S_g[i] - al[i + 1:].dot(SY[i, i + 1:])) / dg[i]
                yq = -Y_g - YY.dot(al) if m else np.zeros(0)
                for i in range(m):
                    be[i] = (H * yq[i] + (al[:i] - be[:i]).dot(SY[:i, i])) / dg[i]
                d = g.mul(-H)
                gtd = -H * gg
                if m:
                    coef = torch.from_numpy(np.concatenate([-H * al, al - be])).to(device=g.device, dtype=g.dtype)
                    d.addmv_(Yb[:m].t(), coef[:m])
                    d.addmv_(Sb[:m].t(), coef[m:])
                    gtd += float(-H * al.dot(Y_g) + (al - be).dot(S_g))
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

    def _strong_wolfe(self, phi, t, d, f, g, gtd, c1=1e-4, c2=0.9, max_ls=25):
        """Bracketing + zoom line search for the strong Wolfe conditions; phi(t) -> (loss, flat grad, grad . d,
        max |grad|, max |d|).  Returns (loss, grad, step, evaluations, max |grad| at the step, max |d|)."""
        tc = self.tolerance_change
        f_new, g_new, gtd_new, gm_new, d_norm = phi(t)
        n_eval = 1
        t_prev, f_prev, g_prev, gtd_prev, gm_prev = 0.0, f, g, gtd, None
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
            br = [[0.0, f, g, gtd, None], [t, f_new, g_new, gtd_new, gm_new]]

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
            _, (gm,), _ = self._scalars(None, g_new.abs().max().reshape(1))
        return f_new, g_new, t, n_eval, gm, d_norm
