// Kernels over time points (quadrature nodes / arbitrary grids):
//   k_times_fwd : posterior mean / variance of latent k at a CSR list of times
//                 (reference svPosteriorOnLatents.py:185-216 and :265-300, fused: Ktz never leaves LDS)
//   k_quad_ell  : expected log-likelihood quadrature term and its adjoints
//                 (reference svEmbedding.py:80-84 + expectedLogLikelihood.py:107-135,205-208)
//   k_quad_bwd  : adjoints wrt a, W and the direct kappa(t_q, Z) dependence
//   k_ktz       : materialise a Ktz block for the KernelsMatricesStore getters (tests only)
#pragma once
#include "engine_common.h"

// ------------------------------------------------------------------------------ k_times_fwd
// grid = R*K blocks of 256 threads.  Output element (time gi, latent k) at gi*os_t + k*os_k.
// Per batch of 64 times: Kt = kappa(t, Z) (64 x M), Kw = Kt Linv^T (whitened), then
//   mu = Kw.ut,  var = s2 - |Kw|^2 + |Vt^T Kw|^2.
__global__ __launch_bounds__(256) void k_times_fwd(EngineView ev, const double* times,
                                                   const long long* toff, int nper, double* mu_out,
                                                   double* var_out, long long os_t, long long os_k,
                                                   int want_var) {
  extern __shared__ double sm[];
  const int item = blockIdx.x;
  const int r = item / ev.K, k = item % ev.K;
  const LatentMeta m = ev.meta[k];
  const int M = m.M, LD = M + 1, tid = threadIdx.x;
  double* Li = sm;               // M*LD
  double* Vl = Li + M * LD;      // M*LD
  double* Kt = Vl + M * LD;      // 64*LD
  double* Kw = Kt + 64 * LD;     // 64*LD
  double* ul = Kw + 64 * LD;     // 64
  double* zl = ul + 64;          // 64
  double* red = zl + 64;         // 3*256
  const KPar p = make_kpar(m, ev.theta + 2 * k);
  const long long t0 = toff ? toff[r] : (long long)r * nper;
  const int nt = toff ? (int)(toff[r + 1] - toff[r]) : nper;
  const long long base = m.offMM + (long long)r * M * M;
  const long long vbase = m.offM + (long long)r * M;

  for (int e = tid; e < M * M; e += 256) {
    Li[(e / M) * LD + (e % M)] = ev.Linv[base + e];
    if (want_var) Vl[(e / M) * LD + (e % M)] = ev.Vt[base + e];
  }
  if (tid < M) {
    ul[tid] = ev.ut[vbase + tid];
    zl[tid] = ev.Zs[vbase + tid];
  }
  __syncthreads();

  for (int b0 = 0; b0 < nt; b0 += 64) {
    const int nb = (nt - b0 < 64) ? (nt - b0) : 64;
    for (int e = tid; e < 64 * M; e += 256) {
      int q = e & 63, j = e >> 6;
      double tq = times[t0 + b0 + (q < nb ? q : 0)];
      double d = tq - zl[j];
      Kt[q * LD + j] =
          p.s2 * (m.ktype == SV_RBF ? kern_val<SV_RBF>(d, p) : kern_val<SV_PERIODIC>(d, p));
    }
    __syncthreads();
    for (int e = tid; e < 64 * M; e += 256) {
      int q = e & 63, i = e >> 6;
      double s = 0.0;
      for (int j = 0; j <= i; ++j) s += Li[i * LD + j] * Kt[q * LD + j];
      Kw[q * LD + i] = s;
    }
    __syncthreads();
    const int q = tid & 63, part = tid >> 6;
    double pm = 0.0, n2 = 0.0, tt = 0.0;
    for (int i = part; i < M; i += 4) {
      double kw = Kw[q * LD + i];
      pm += kw * ul[i];
      n2 += kw * kw;
    }
    if (want_var) {
      for (int j = part; j < M; j += 4) {
        double t = 0.0;
        for (int i = j; i < M; ++i) t += Vl[i * LD + j] * Kw[q * LD + i];
        tt += t * t;
      }
    }
    red[part * 64 + q] = pm;
    red[256 + part * 64 + q] = n2;
    red[512 + part * 64 + q] = tt;
    __syncthreads();
    if (part == 0 && q < nb) {
      long long gi = t0 + b0 + q;
      double s = ((red[q] + red[64 + q]) + red[128 + q]) + red[192 + q];
      mu_out[gi * os_t + k * os_k] = s;
      if (want_var) {
        double a2 = ((red[256 + q] + red[320 + q]) + red[384 + q]) + red[448 + q];
        double b2 = ((red[512 + q] + red[576 + q]) + red[640 + q]) + red[704 + q];
        var_out[gi * os_t + k * os_k] = (p.s2 - a2) + b2;
      }
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------- k_quad_ell
// grid = R*nqt blocks of 256 threads; KT = K rounded up to a multiple of 4 (zero padded).
// pass 1 (thread = neuron): E = w exp(h), ELL partial, dC / dd partials (column sums of E)
// pass 2 (thread = node)  : adjoints of the latent mean / variance (row sums of E)
template <int KT>
__global__ __launch_bounds__(256) void k_quad_ell(EngineView ev, const double* C, const double* d,
                                                  int want_cd, int want_lat) {
  extern __shared__ double sm[];
  const int blk = blockIdx.x, tid = threadIdx.x;
  const int r = blk / ev.nqt, qt = blk % ev.nqt;
  const int q0 = qt * ev.qtile;
  const int nq = (ev.Q - q0 < ev.qtile) ? (ev.Q - q0) : ev.qtile;
  const int K = ev.K, N = ev.N;
  double* mul = sm;                    // qtile*KT
  double* varl = mul + ev.qtile * KT;  // qtile*KT
  double* wl = varl + ev.qtile * KT;   // qtile
  double* Cl = wl + ev.qtile;          // N*KT
  double* dl = Cl + (long long)N * KT; // N
  double* red = dl + N;                // 4

  for (int e = tid; e < nq * KT; e += 256) {
    int q = e / KT, k = e % KT;
    long long gi = ((long long)r * ev.Q + q0 + q) * K + k;
    mul[e] = (k < K) ? ev.muq[gi] : 0.0;
    varl[e] = (k < K) ? ev.varq[gi] : 0.0;
  }
  for (int q = tid; q < nq; q += 256) wl[q] = ev.wq[(long long)r * ev.Q + q0 + q];
  for (int e = tid; e < N * KT; e += 256) {
    int n = e / KT, k = e % KT;
    Cl[e] = (k < K) ? C[(long long)n * K + k] : 0.0;
  }
  for (int n = tid; n < N; n += 256) dl[n] = d[n];
  __syncthreads();

  double esum = 0.0;
  for (int n = tid; n < N; n += 256) {
    double Cn[KT], hC[KT], gA[KT], gB[KT];
#pragma unroll
    for (int k = 0; k < KT; ++k) {
      Cn[k] = Cl[n * KT + k];
      hC[k] = 0.5 * Cn[k];
      gA[k] = 0.0;
      gB[k] = 0.0;
    }
    const double dn = dl[n];
    double gd = 0.0;
    for (int q = 0; q < nq; ++q) {
      double h = dn;
#pragma unroll
      for (int k = 0; k < KT; ++k) h = fma(Cn[k], fma(hC[k], varl[q * KT + k], mul[q * KT + k]), h);
      double E = wl[q] * sv_exp2(h * 1.4426950408889634074);
      gd += E;
      if (want_cd) {
#pragma unroll
        for (int k = 0; k < KT; ++k) {
          gA[k] = fma(E, mul[q * KT + k], gA[k]);
          gB[k] = fma(E, varl[q * KT + k], gB[k]);
        }
      }
    }
    esum += gd;
    if (want_cd) {
#pragma unroll
      for (int k = 0; k < KT; ++k)
        if (k < K) {
          ev.gCpart[((long long)blk * N + n) * 2 * K + k] = gA[k];
          ev.gCpart[((long long)blk * N + n) * 2 * K + K + k] = gB[k];
        }
      ev.gdpart[(long long)blk * N + n] = -gd;
    }
  }
  double tot = block_sum(esum, red);
  if (tid == 0) ev.ellq[blk] = -tot;

  if (want_lat) {
    for (int q = tid; q < nq; q += 256) {
      double mq[KT], vq[KT], gm[KT], gv[KT];
#pragma unroll
      for (int k = 0; k < KT; ++k) {
        mq[k] = mul[q * KT + k];
        vq[k] = varl[q * KT + k];
        gm[k] = 0.0;
        gv[k] = 0.0;
      }
      const double w = wl[q];
      for (int n = 0; n < N; ++n) {
        double h = dl[n];
#pragma unroll
        for (int k = 0; k < KT; ++k) {
          double c = Cl[n * KT + k];
          h = fma(c, fma(0.5 * c, vq[k], mq[k]), h);
        }
        double E = w * sv_exp2(h * 1.4426950408889634074);
#pragma unroll
        for (int k = 0; k < KT; ++k) {
          double c = Cl[n * KT + k];
          double ec = E * c;
          gm[k] += ec;
          gv[k] = fma(ec, c, gv[k]);
        }
      }
#pragma unroll
      for (int k = 0; k < KT; ++k)
        if (k < K) {
          long long gi = ((long long)r * ev.Q + q0 + q) * K + k;
          ev.gmuq[gi] = -gm[k];
          ev.gvarq[gi] = -0.5 * gv[k];
        }
    }
  }
}

// ------------------------------------------------------------------------------- k_quad_bwd
// grid = R*K blocks of 256 threads.  Per batch of QB nodes (whitened):
//   Kw = Kt Linv^T ; gt += gmu Kw ; Gt += gvar Kw Kw^T                       (always)
//   FULL: T = Kw Vt, Y = T Vt^T, adj_Kw = gmu ut + 2 gvar (Y - Kw), gk = adj_Kw Linv  (dELL/dkappa(t_q, z_j))
//         -> direct dZ_j = cz sum_q gk ktil bz, dtheta_t = c_t sum_qj gk ktil b_t
template <int KT, bool FULL>
__device__ void quad_bwd_body(const EngineView& ev, double* sm) {
  const int item = blockIdx.x, tid = threadIdx.x;
  const int r = item / ev.K, k = item % ev.K;
  const LatentMeta m = ev.meta[k];
  const int M = m.M, LD = M + 1, Q = ev.Q, K = ev.K;
  const int QB = (M <= 32) ? 64 : 32, QSH = (M <= 32) ? 6 : 5;
  double* Li = sm;                         // M*LD
  double* Vl = Li + M * LD;                // M*LD
  double* Kt = Vl + M * LD;                // QB*LD  (later adj_Kw)
  double* Kw = Kt + QB * LD;               // QB*LD
  double* Tl = Kw + QB * LD;               // QB*LD
  double* ul = Tl + QB * LD;               // 64
  double* zl = ul + 64;
  double* gmu = zl + 64;
  double* gvar = gmu + 64;
  double* tql = gvar + 64;
  double* red = tql + 64;                  // 4
  const KPar p = make_kpar(m, ev.theta + 2 * k);
  const long long base = m.offMM + (long long)r * M * M;
  const long long vbase = m.offM + (long long)r * M;

  for (int e = tid; e < M * M; e += 256) {
    Li[(e / M) * LD + (e % M)] = ev.Linv[base + e];
    if (FULL) Vl[(e / M) * LD + (e % M)] = ev.Vt[base + e];
  }
  if (tid < M) {
    ul[tid] = ev.ut[vbase + tid];
    zl[tid] = ev.Zs[vbase + tid];
  }
  double gw[16];
#pragma unroll
  for (int n = 0; n < 16; ++n) gw[n] = 0.0;
  double gu_acc = 0.0, gz_acc = 0.0, th0 = 0.0, th1 = 0.0;

  for (int b0 = 0; b0 < Q; b0 += QB) {
    const int nb = (Q - b0 < QB) ? (Q - b0) : QB;
    __syncthreads();
    if (tid < QB) {
      long long gi = ((long long)r * Q + b0 + tid) * K + k;
      bool ok = tid < nb;
      gmu[tid] = ok ? ev.gmuq[gi] : 0.0;
      gvar[tid] = ok ? ev.gvarq[gi] : 0.0;
      tql[tid] = ok ? ev.tq[(long long)r * Q + b0 + tid] : 0.0;
    }
    __syncthreads();
    double kbz[8], kb0[8], kb1[8];
#pragma unroll
    for (int n = 0; n < 8; ++n) {
      int e = tid + 256 * n;
      if (e < QB * M) {
        int q = e & (QB - 1), j = e >> QSH;
        double kt, bz, b0v, b1v;
        kern_eval<KT>(tql[q] - zl[j], p, kt, bz, b0v, b1v);
        Kt[q * LD + j] = p.s2 * kt;
        kbz[n] = kt * bz;
        kb0[n] = kt * b0v;
        kb1[n] = kt * b1v;
      }
    }
    __syncthreads();
#pragma unroll
    for (int n = 0; n < 8; ++n) {
      int e = tid + 256 * n;
      if (e < QB * M) {
        int q = e & (QB - 1), i = e >> QSH;
        double s = 0.0;
        for (int j = 0; j <= i; ++j) s += Li[i * LD + j] * Kt[q * LD + j];
        Kw[q * LD + i] = s;
      }
    }
    __syncthreads();
    if (FULL) {
#pragma unroll
      for (int n = 0; n < 8; ++n) {   // T = Kw Vt
        int e = tid + 256 * n;
        if (e < QB * M) {
          int q = e & (QB - 1), j = e >> QSH;
          double s = 0.0;
          for (int i = j; i < M; ++i) s += Vl[i * LD + j] * Kw[q * LD + i];
          Tl[q * LD + j] = s;
        }
      }
      __syncthreads();
#pragma unroll
      for (int n = 0; n < 8; ++n) {   // adj_Kw = gmu ut + 2 gvar (T Vt^T - Kw)  -> Kt buffer
        int e = tid + 256 * n;
        if (e < QB * M) {
          int q = e & (QB - 1), i = e >> QSH;
          double y = 0.0;
          for (int j = 0; j <= i; ++j) y += Vl[i * LD + j] * Tl[q * LD + j];
          Kt[q * LD + i] = gmu[q] * ul[i] + 2.0 * gvar[q] * (y - Kw[q * LD + i]);
        }
      }
      __syncthreads();
#pragma unroll
      for (int n = 0; n < 8; ++n) {   // gk = adj_Kw Linv ; direct Z / theta integrands
        int e = tid + 256 * n;
        if (e < QB * M) {
          int q = e & (QB - 1), j = e >> QSH;
          double gk = 0.0;
          for (int i = j; i < M; ++i) gk += Li[i * LD + j] * Kt[q * LD + i];
          th0 += gk * kb0[n];
          th1 += gk * kb1[n];
          Tl[q * LD + j] = gk * kbz[n];
        }
      }
      __syncthreads();
    }
    if (tid < M) {
      for (int q = 0; q < QB; ++q) {
        gu_acc += gmu[q] * Kw[q * LD + tid];
        if (FULL) gz_acc += Tl[q * LD + tid];
      }
    }
#pragma unroll
    for (int n = 0; n < 16; ++n) {
      int e = tid + 256 * n;
      if (e < M * M) {
        int i = e / M, j = e % M;
        double s = 0.0;
        for (int q = 0; q < QB; ++q) s += gvar[q] * Kw[q * LD + i] * Kw[q * LD + j];
        gw[n] += s;
      }
    }
  }
  if (tid < M) {
    ev.gaq[vbase + tid] = gu_acc;
    if (FULL) ev.gZq[vbase + tid] = p.cz * gz_acc;
  }
#pragma unroll
  for (int n = 0; n < 16; ++n) {
    int e = tid + 256 * n;
    if (e < M * M) ev.GW[base + e] = gw[n];
  }
  if (FULL) {
    double s0 = block_sum(th0, red);
    double s1 = block_sum(th1, red);
    if (tid == 0) {
      ev.gthq[item * 2 + 0] = p.c0 * s0;
      ev.gthq[item * 2 + 1] = p.c1 * s1;
    }
  }
}

template <bool FULL>
__global__ __launch_bounds__(256) void k_quad_bwd(EngineView ev) {
  extern __shared__ double sm[];
  const int k = blockIdx.x % ev.K;
  if (ev.meta[k].ktype == SV_RBF)
    quad_bwd_body<SV_RBF, FULL>(ev, sm);
  else
    quad_bwd_body<SV_PERIODIC, FULL>(ev, sm);
}

// ------------------------------------------------------------------------------------ k_ktz
__global__ void k_ktz(EngineView ev, int k, const double* times, const long long* toff, int nper,
                      long long ntot, double* out) {
  const LatentMeta m = ev.meta[k];
  const KPar p = make_kpar(m, ev.theta + 2 * k);
  long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= ntot * m.M) return;
  long long gi = id / m.M;
  int j = (int)(id % m.M);
  int r;
  if (toff) {
    int lo = 0, hi = ev.R;  // find trial with toff[r] <= gi < toff[r+1]
    while (hi - lo > 1) {
      int mid = (lo + hi) >> 1;
      if (toff[mid] <= gi) lo = mid; else hi = mid;
    }
    r = lo;
  } else {
    r = (int)(gi / nper);
  }
  double d = times[gi] - ev.Zs[m.offM + (long long)r * m.M + j];
  out[id] = p.s2 * (m.ktype == SV_RBF ? kern_val<SV_RBF>(d, p) : kern_val<SV_PERIODIC>(d, p));
}
