"""Thin Python handle over the C ABI: owns nothing numerical, only marshals torch ROCm tensors.

All tensors handed in must live on the engine's GPU, be fp64 and contiguous; pointers are taken
with data_ptr() (zero copy).  torch is used for device memory and streams only.
"""
import ctypes

import numpy as np
import torch

from . import _lib


class SVGPFAEngineError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"svgpfa error {code}: {msg}")
        self.code = code


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


class Engine:
    """One engine = one GPU = one contiguous block of trials (see include/svgpfa.h)."""

    def __init__(self, n_trials, n_neurons, n_quad, n_ind, kernel_types, scales=None,
                 lengthscale_scales=None, period_scales=None, device=None, precision="fp64"):
        self._h = None
        self._tables = {}
        self._stream = None
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise RuntimeError("svgpfa_amd needs a ROCm GPU (MI355X); no CPU fallback exists")
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None else device)
        K = len(n_ind)
        self.R, self.N, self.K, self.Q = int(n_trials), int(n_neurons), K, int(n_quad)
        self.Ms = [int(m) for m in n_ind]
        self.Ps = [m * (m + 1) // 2 for m in self.Ms]
        self.ktypes = [int(t) for t in kernel_types]
        self.nth = [1 if t == _lib.KERNEL_RBF else 2 for t in self.ktypes]
        self._szM = [self.R * m for m in self.Ms]
        self._szP = [self.R * p for p in self.Ps]
        arr_i = lambda xs: (ctypes.c_int32 * K)(*[int(x) for x in xs])
        arr_d = lambda xs: (ctypes.c_double * K)(*[float(x) for x in xs]) if xs is not None else None
        self._keep = (arr_i(self.Ms), arr_i(self.ktypes), arr_d(scales), arr_d(lengthscale_scales),
                      arr_d(period_scales))
        if precision not in ("fp64", "mixed"):
            raise ValueError("precision must be 'fp64' or 'mixed'")
        self.precision = precision
        dims = _lib.Dims(self.R, self.N, K, self.Q, self._keep[0], self._keep[1],
                         self._keep[2], self._keep[3], self._keep[4],
                         _lib.PRECISION_MIXED if precision == "mixed" else _lib.PRECISION_FP64)
        h = ctypes.c_void_p()
        rc = self.lib.svgpfa_create(ctypes.byref(dims), self.device.index, ctypes.byref(h))
        self._h = h
        self._check(rc)
        self.shared_len = int(self.lib.svgpfa_shared_len(self._h))
        self.S = 0
        self._use_current_stream()

    # ------------------------------------------------------------------ plumbing
    def _check(self, rc):
        if rc != _lib.OK:
            msg = self.lib.svgpfa_last_error(self._h).decode() if self._h else "engine creation failed"
            raise SVGPFAEngineError(rc, msg)

    def _use_current_stream(self):
        s = torch.cuda.current_stream(self.device).cuda_stream
        if s != self._stream:
            self._check(self.lib.svgpfa_set_stream(self._h, ctypes.c_void_p(s)))
            self._stream = s

    def _dev(self, t, dtype=torch.float64):
        if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == dtype and t.is_contiguous()
                and t.device == self.device):
            raise ValueError(f"expected a contiguous {dtype} tensor on {self.device}, got "
                             f"{type(t).__name__} {getattr(t, 'dtype', None)} {getattr(t, 'device', None)}")
        return t

    def _table(self, tensors, sizes=None, slot=None):
        """Host array of K device pointers.  `slot` names the argument: the array built for the same tensors (same
        addresses) on the previous call is reused without re-validating them -- an EM closure passes the same parameter
        tensors thousands of times, and on small problems the marshalling was longer than the kernels."""
        if tensors is None:
            return None
        if len(tensors) != self.K:
            raise ValueError("expected one tensor per latent")
        ptrs = tuple(t.data_ptr() for t in tensors)
        # A hit needs the SAME tensor objects at the same addresses.  The cache keeps the validated tensors alive, so neither
        # their addresses nor their ids can be recycled for a tensor of another size or type (the caching allocator does
        # recycle addresses: an address-only key would let such a tensor ride on the previous one's validation).  Identity
        # and address cost ~2 us per table; re-reading size / dtype / device of every tensor cost 5 (x 6 tables per call).
        if slot is not None:
            hit = self._tables.get(slot)
            if hit is not None and hit[0] == ptrs and len(hit[2]) == len(tensors) and \
                    all(a is b for a, b in zip(hit[2], tensors)):
                return hit[1]
        for k, t in enumerate(tensors):
            self._dev(t)
            if sizes is not None and t.numel() != sizes[k]:
                raise ValueError(f"latent {k}: expected {sizes[k]} elements, got {t.numel()}")
        arr = (ctypes.c_void_p * self.K)(*ptrs)
        if slot is not None:
            self._tables[slot] = (ptrs, arr, list(tensors))
        return arr

    def close(self):
        if self._h:
            self.lib.svgpfa_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---------------------------------------------------------------------- data
    def set_reg(self, eps):
        self._check(self.lib.svgpfa_set_reg(self._h, float(eps)))

    def set_quad(self, points, weights):
        self._use_current_stream()
        p, w = self._dev(points), self._dev(weights)
        if p.numel() != self.R * self.Q or w.numel() != self.R * self.Q:
            raise ValueError("quadrature arrays must hold n_trials x n_quad values")
        self._check(self.lib.svgpfa_set_quad(self._h, _ptr(p), _ptr(w)))

    def set_spikes(self, offsets, times, neuron):
        """offsets: host int64 array of R+1; times (S,) fp64 and neuron (S,) int32 on the device."""
        self._use_current_stream()
        off = np.ascontiguousarray(np.asarray(offsets, dtype=np.int64))
        if off.shape != (self.R + 1,):
            raise ValueError("offsets must have n_trials + 1 entries")
        S = int(off[-1])
        if S:
            self._dev(times)
            self._dev(neuron, torch.int32)
            if times.numel() != S or neuron.numel() != S:
                raise ValueError("times / neuron length must equal offsets[-1]")
        torch.cuda.current_stream(self.device).synchronize()
        self._check(self.lib.svgpfa_set_spikes(
            self._h, off.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)),
            _ptr(times if S else None), _ptr(neuron if S else None)))
        self.S = S

    # ------------------------------------------------------------------- compute
    def build_kernels(self, Z, theta):
        self._use_current_stream()
        zt = self._table(Z, self._szM, "Z")
        tt = self._table(theta, self.nth, "theta")
        self._check(self.lib.svgpfa_build_kernels(self._h, zt, tt))

    def alloc_grads(self, grad_mask):
        """Per-trial gradient buffers (g_mean, g_cholvec, g_Z) for `grad_mask` (None where not requested)."""
        mk = lambda r, c: torch.empty(r, c, dtype=torch.float64, device=self.device)
        g_mean = g_chol = g_Z = None
        if grad_mask & _lib.GRAD_VARIATIONAL:
            g_mean = [mk(self.R, m) for m in self.Ms]
            g_chol = [mk(self.R, p) for p in self.Ps]
        if grad_mask & _lib.GRAD_INDPOINTS:
            g_Z = [mk(self.R, m) for m in self.Ms]
        return g_mean, g_chol, g_Z

    def elbo(self, mean, cholvec, C, d, grad_mask=0, out=None, grads=None):
        """Enqueue one ELBO(+grad) evaluation.  Returns (shared, g_mean, g_cholvec, g_Z); the
        lists are None for groups not in grad_mask.  Call sync() before trusting the numbers."""
        self._use_current_stream()
        mt = self._table(mean, self._szM, "mean")
        ct = self._table(cholvec, self._szP, "chol")
        cd = (C.data_ptr(), d.data_ptr())
        hit = self._tables.get("Cd")
        if hit is None or hit[0] != cd or hit[1] is not C or hit[2] is not d:      # same objects at the same addresses
            self._dev(C), self._dev(d)
            if C.numel() != self.N * self.K or d.numel() != self.N:
                raise ValueError("C must be N x K and d must have N entries")
            self._tables["Cd"] = (cd, C, d)
        shared = out if out is not None else torch.empty(self.shared_len, dtype=torch.float64, device=self.device)
        g_mean, g_chol, g_Z = grads if grads is not None else self.alloc_grads(grad_mask)
        self._check(self.lib.svgpfa_elbo(self._h, mt, ct, cd[0], cd[1], int(grad_mask), _ptr(shared),
                                         self._table(g_mean, self._szM, "g_mean"),
                                         self._table(g_chol, self._szP, "g_chol"),
                                         self._table(g_Z, self._szM, "g_Z")))
        return shared, g_mean, g_chol, g_Z

    def set_embedding_token(self, token):
        """Content id of C for the reuse of the spike sums across value-only evaluations (include/svgpfa.h); 0 disables."""
        self._check(self.lib.svgpfa_set_embedding_token(self._h, int(token) & 0xFFFFFFFFFFFFFFFF))

    def split_shared(self, shared):
        """Views into the shared vector: (elbo, ell, kl, gC N x K, gd N, [gtheta_k])."""
        NK = self.N * self.K
        gC = shared[3:3 + NK].view(self.N, self.K)
        gd = shared[3 + NK:3 + NK + self.N]
        gth, o = [], 3 + NK + self.N
        for n in self.nth:
            gth.append(shared[o:o + n])
            o += n
        return shared[0], shared[1], shared[2], gC, gd, gth

    def latent_stats(self, mean, cholvec, want_outputs=True):
        self._use_current_stream()
        mt = self._table(mean, [self.R * m for m in self.Ms])
        ct = self._table(cholvec, [self.R * p for p in self.Ps])
        outs = [None] * 4
        if want_outputs:
            mk = lambda *s: torch.empty(*s, dtype=torch.float64, device=self.device)
            outs = [mk(self.R, self.Q, self.K), mk(self.R, self.Q, self.K), mk(self.S, self.K), mk(self.S, self.K)]
        self._check(self.lib.svgpfa_latent_stats(self._h, mt, ct, *[_ptr(o) for o in outs]))
        return outs

    def freeze_stats(self, mean, cholvec, C):
        """The statistics ell_given_stats needs, kept resident, without the mean / variance at every spike."""
        self._use_current_stream()
        self._dev(C)
        mt = self._table(mean, [self.R * m for m in self.Ms])
        ct = self._table(cholvec, [self.R * p for p in self.Ps])
        self._check(self.lib.svgpfa_freeze_stats(self._h, mt, ct, _ptr(C)))

    def series_harmonics(self):
        """Per latent: H >= 0 if the spike pass evaluates the (periodic) kernel through H + 1 terms of its cosine series,
        -1 if pair by pair (include/svgpfa.h).  Waits for the stream."""
        out = (ctypes.c_int32 * self.K)()
        self._check(self.lib.svgpfa_get_series_harmonics(self._h, out))
        return list(out)

    def kl(self):
        """KL term of the parameters of the latest elbo / latent_stats / predict call (0-dim tensor)."""
        self._use_current_stream()
        out = torch.empty(1, dtype=torch.float64, device=self.device)
        self._check(self.lib.svgpfa_kl(self._h, _ptr(out)))
        return out[0]

    def latent_stats_backward(self, mean, cholvec, g_mu_quad, g_var_quad, g_mu_spk, g_var_spk, grad_mask):
        """Gradients of (caller's ELL) - KL given d ELL / d statistics (include/svgpfa.h).  Returns
        (shared, g_mean, g_cholvec, g_Z): shared[2] = KL, shared's trailing entries = dtheta."""
        self._use_current_stream()
        mt = self._table(mean, [self.R * m for m in self.Ms])
        ct = self._table(cholvec, [self.R * p for p in self.Ps])
        for t, n in ((g_mu_quad, self.R * self.Q * self.K), (g_var_quad, self.R * self.Q * self.K),
                     (g_mu_spk, self.S * self.K), (g_var_spk, self.S * self.K)):
            if n:
                self._dev(t)
                if t.numel() != n:
                    raise ValueError("adjoint tensor of the wrong size")
        grad_mask = int(grad_mask) & (_lib.GRAD_VARIATIONAL | _lib.GRAD_KERNELS | _lib.GRAD_INDPOINTS)
        shared = torch.zeros(self.shared_len, dtype=torch.float64, device=self.device)
        g_mean, g_chol, g_Z = self.alloc_grads(grad_mask)
        self._check(self.lib.svgpfa_latent_stats_backward(
            self._h, mt, ct, _ptr(g_mu_quad), _ptr(g_var_quad), _ptr(g_mu_spk if self.S else None),
            _ptr(g_var_spk if self.S else None), grad_mask, _ptr(shared),
            self._table(g_mean, [self.R * m for m in self.Ms]), self._table(g_chol, [self.R * p for p in self.Ps]),
            self._table(g_Z, [self.R * m for m in self.Ms])))
        return shared, g_mean, g_chol, g_Z

    def ell_given_stats(self, C, d, want_grad=True):
        self._use_current_stream()
        self._dev(C), self._dev(d)
        out = torch.empty(1 + self.N * self.K + self.N, dtype=torch.float64, device=self.device)
        self._check(self.lib.svgpfa_ell_given_stats(self._h, _ptr(C), _ptr(d), int(bool(want_grad)), _ptr(out)))
        return out

    def predict(self, mean, cholvec, times, want_var=True):
        self._use_current_stream()
        mt = self._table(mean, [self.R * m for m in self.Ms])
        ct = self._table(cholvec, [self.R * p for p in self.Ps])
        self._dev(times)
        if times.numel() % self.R:
            raise ValueError("times must be n_trials x T")
        T = times.numel() // self.R
        mu = torch.empty(self.R, T, self.K, dtype=torch.float64, device=self.device)
        var = torch.empty_like(mu) if want_var else None
        self._check(self.lib.svgpfa_predict(self._h, mt, ct, _ptr(times), T, _ptr(mu), _ptr(var)))
        return mu, var

    def get_kzz(self, k):
        self._use_current_stream()
        out = torch.empty(self.R, self.Ms[k], self.Ms[k], dtype=torch.float64, device=self.device)
        self._check(self.lib.svgpfa_get_kzz(self._h, k, _ptr(out)))
        return out

    def get_kzz_chol(self, k):
        self._use_current_stream()
        out = torch.empty(self.R, self.Ms[k], self.Ms[k], dtype=torch.float64, device=self.device)
        self._check(self.lib.svgpfa_get_kzz_chol(self._h, k, _ptr(out)))
        return out

    def get_ktz_quad(self, k):
        self._use_current_stream()
        out = torch.empty(self.R, self.Q, self.Ms[k], dtype=torch.float64, device=self.device)
        self._check(self.lib.svgpfa_get_ktz_quad(self._h, k, _ptr(out)))
        return out

    def get_ktz_spikes(self, k):
        self._use_current_stream()
        out = torch.empty(self.S, self.Ms[k], dtype=torch.float64, device=self.device)
        self._check(self.lib.svgpfa_get_ktz_spikes(self._h, k, _ptr(out)))
        return out

    def sync(self):
        """Wait for the stream; raises SVGPFAEngineError(ERR_NOT_PD) if a Cholesky failed."""
        self._check(self.lib.svgpfa_sync(self._h))

    def enable_timing(self, on=True):
        """False / True: off / every evaluation's spike pass and k_item_bwd bracketed with hipEvent pairs; an int P > 1: every
        P-th evaluation's k_item_bwd and, half a period later, spike pass (one pair per bracketed evaluation)."""
        self._check(self.lib.svgpfa_enable_timing(self._h, int(on)))

    def kernel_ms(self, which):
        """(total device ms, launches) of the spike pass (which = 0) or k_item_bwd (which = 1) since enable_timing(True)
        (syncs the streams)."""
        tot, n = ctypes.c_double(0.0), ctypes.c_int32(0)
        self._check(self.lib.svgpfa_kernel_ms(self._h, int(which), ctypes.byref(tot), ctypes.byref(n)))
        return tot.value, n.value

    def spike_kernel_ms(self):
        return self.kernel_ms(0)


def engine_from_problem(prob, device=None, precision="fp64"):
    """Build an Engine + device parameter tensors from a numpy problem (svgpfa_amd.utils.synthetic
    layout).  Returns (engine, params) with params = dict(mean, cholVecs, Z, theta, C, d)."""
    from .utils import synthetic
    dev = torch.device("cuda", torch.cuda.current_device() if device is None else device)
    kt = [_lib.KERNEL_RBF if t == synthetic.RBF else _lib.KERNEL_PERIODIC for t in prob["kernel_types"]]
    eng = Engine(prob["R"], prob["N"], prob["Q"], prob["Ms"], kt, prob.get("scales"),
                 prob.get("lengthscale_scales"), prob.get("period_scales"), device=dev.index, precision=precision)
    td = lambda a, dt=torch.float64: torch.as_tensor(np.ascontiguousarray(a), dtype=dt).to(dev).contiguous()
    eng.set_reg(prob["reg"])
    eng.set_quad(td(prob["quad_points"]), td(prob["quad_weights"]))
    eng.set_spikes(prob["spike_offsets"], td(prob["spike_times"]), td(prob["spike_neuron"], torch.int32))
    params = dict(mean=[td(a) for a in prob["mean"]], cholVecs=[td(a) for a in prob["cholVecs"]],
                  Z=[td(a) for a in prob["Z"]], theta=[td(a) for a in prob["theta"]],
                  C=td(prob["C"]), d=td(prob["d"]))
    eng.build_kernels(params["Z"], params["theta"])
    return eng, params
