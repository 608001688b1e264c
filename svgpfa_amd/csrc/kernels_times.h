// VALU kernels over time points:
//   k_quad_ell  : expected log-likelihood quadrature term and its adjoints, scalar form
//                 (reference svEmbedding.py:80-84 + expectedLogLikelihood.py:107-135,205-208).
//                 FALLBACK only: used when the GEMM-form MFMA kernels (kernels_mfma.h) do not fit LDS
//                 (large N * K, e.g. BASELINE config 5).
//   k_ktz       : materialise a Ktz block for the KernelsMatricesStore getters (tests / plots only)
// The posterior, quadrature-backward and per-item backward kernels live in kernels_mfma.h.
#pragma once
#include "engine_common.h"

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
