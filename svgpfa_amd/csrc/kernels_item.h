// Per-(trial, latent) kernels: one wavefront owns one M x M problem held in LDS.
//   k_build : Kzz + eps I, Cholesky L, Linv = L^-1   (reference kernelsMatricesStore.py:107-117, miscUtils.py:209-216)
//   k_prep  : ut = Linv m, a = Linv^T ut (= Kzz^-1 m), Vt = Linv L_S, KL
//             (reference svPosteriorOnLatents.py:185-216 algebra in whitened form, klDivergence.py:31-44)
//   k_bwd   : adjoints of (ut, Vt Vt^T, KL, direct kappa terms) -> gradients wrt m, cholVecs, Z, theta
// Derivation (per trial/latent; Kzz = L L^T, S = L_S L_S^T, kt = Linv k(t)):
//   mu(t) = kt.ut,  var(t) = s2 - |kt|^2 + |Vt^T kt|^2,  KL = (|Vt|_F^2 + |ut|^2 + logdet Kzz - logdet S - M)/2
//   with gt = dELL/d ut and Gt = sum_t dELL/dvar(t) kt kt^T (adjoint of Vt Vt^T):
//   dELBO/dm    = Linv^T (gt - ut)
//   dELBO/dL_S  = tril(Linv^T (2 Gt - I) Vt) + diag(1 / L_S,ii)
//   dELBO/dKzz  = Linv^T At Linv,  At = -(gt ut^T + ut gt^T)/2 + ut ut^T/2 + Gt + (Vt Vt^T - I)/2 - C - C^T,  C = Vt Vt^T Gt
// LDS rows are padded to LD = M + 1 doubles so that column accesses (stride LD) are bank-conflict free.
#pragma once
#include "engine_common.h"

// Lane -> (column j, row group rg) map for M x M outputs computed by one wavefront.
struct WaveMap {
  int j, rg, G;
  bool act;
};
__device__ __forceinline__ WaveMap wave_map(int M) {
  int mp2 = 1;
  while (mp2 < M) mp2 <<= 1;
  WaveMap w;
  int lane = threadIdx.x & 63;
  w.G = 64 / mp2;
  w.j = lane % mp2;
  w.rg = lane / mp2;
  w.act = w.j < M;
  return w;
}

// -------------------------------------------------------------------------------- k_build
__global__ __launch_bounds__(64) void k_build(EngineView ev) {
  extern __shared__ double sm[];
  const int item = blockIdx.x;
  const int r = item / ev.K, k = item % ev.K;
  const LatentMeta m = ev.meta[k];
  const int M = m.M, LD = M + 1, lane = threadIdx.x;
  double* A = sm;
  double* B = sm + M * LD;
  const double* z = ev.Zs + m.offM + (long long)r * M;
  const KPar p = make_kpar(m, ev.theta + 2 * k);
  const long long base = m.offMM + (long long)r * M * M;

  for (int e = lane; e < M * M; e += 64) {
    int i = e / M, j = e % M;
    double d = z[i] - z[j];
    double v = p.s2 * (m.ktype == SV_RBF ? kern_val<SV_RBF>(d, p) : kern_val<SV_PERIODIC>(d, p));
    if (i == j) v += ev.reg;
    A[i * LD + j] = v;
    B[i * LD + j] = 0.0;
    ev.Kzz[base + e] = v;
  }
  __syncthreads();

  // left-looking Cholesky, lane = row
  bool bad = false;
  for (int j = 0; j < M; ++j) {
    double s = 0.0;
    if (lane >= j && lane < M) {
      s = A[lane * LD + j];
      for (int q = 0; q < j; ++q) s -= A[lane * LD + q] * A[j * LD + q];
    }
    double d = __shfl(s, j, 64);
    if (!(d > 0.0)) bad = true;
    double piv = sqrt(d);
    if (lane >= j && lane < M) A[lane * LD + j] = (lane == j) ? piv : s / piv;
    __syncthreads();
  }
  if (bad && lane == 0) atomicCAS(ev.status, 0, 1 + item);

  // Linv = L^-1 by forward substitution, lane = column
  if (lane < M) {
    const int c = lane;
    for (int i = c; i < M; ++i) {
      double s = (i == c) ? 1.0 : 0.0;
      for (int q = c; q < i; ++q) s -= A[i * LD + q] * B[q * LD + c];
      B[i * LD + c] = s / A[i * LD + i];
    }
  }
  __syncthreads();

  for (int e = lane; e < M * M; e += 64) {
    int i = e / M, j = e % M;
    ev.L[base + e] = (j <= i) ? A[i * LD + j] : 0.0;
    ev.Linv[base + e] = B[i * LD + j];
  }
}

// --------------------------------------------------------------------------------- k_prep
__device__ void k_prep_body(const EngineView& ev, const LatentPtrs pt, double* sm) {
  const int item = blockIdx.x;
  const int r = item / ev.K, k = item % ev.K;
  const LatentMeta m = ev.meta[k];
  const int M = m.M, LD = M + 1, lane = threadIdx.x;
  double* Ls = sm;             // M*LD
  double* mv = Ls + M * LD;    // 64
  double* uv = mv + 64;        // 64
  const long long base = m.offMM + (long long)r * M * M;
  const long long vbase = m.offM + (long long)r * M;
  const double* __restrict__ Li = ev.Linv + base;   // global; accesses below are wave-uniform or row-contiguous
  const double* cv = pt.cholvec + (long long)r * m.P;

  for (int e = lane; e < M * M; e += 64) {
    int i = e / M, j = e % M;
    Ls[i * LD + j] = (j <= i) ? cv[i * (i + 1) / 2 + j] : 0.0;
  }
  if (lane < M) mv[lane] = pt.mean[(long long)r * M + lane];
  __syncthreads();

  double u = 0.0;
  if (lane < M) {
    for (int q = 0; q <= lane; ++q) u += Li[lane * M + q] * mv[q];
    uv[lane] = u;
    ev.ut[vbase + lane] = u;
  }
  const double quad = wave_sum(u * u);
  __syncthreads();
  if (lane < M) {
    double s = 0.0;
    for (int q = lane; q < M; ++q) s += Li[q * M + lane] * uv[q];
    ev.a[vbase + lane] = s;
  }

  const WaveMap w = wave_map(M);
  double fro = 0.0;
  if (w.act) {
    const int j = w.j;
    for (int i = w.rg; i < M; i += w.G) {
      double s = 0.0;
      for (int q = j; q <= i; ++q) s += Li[i * M + q] * Ls[q * LD + j];   // empty above the diagonal
      ev.Vt[base + i * M + j] = s;
      fro += s * s;
    }
  }
  fro = wave_sum(fro);
  double ld = 0.0;
  if (lane < M) ld = -2.0 * log(Li[lane * M + lane]) - 2.0 * log(fabs(Ls[lane * LD + lane]));
  ld = wave_sum(ld);
  if (lane == 0) ev.klpart[item] = 0.5 * (fro + quad + ld - (double)M);
}

// ---------------------------------------------------------------------------------- k_bwd
template <int KT>
__device__ void bwd_kzz_path(const LatentMeta& m, const KPar& p, const double* A, int LD, const double* zv,
                             double* gz_out, double* th_acc) {
  // dZ_i = sum_j (A_ij + A_ji) dKzz_ij/dZ_i with A symmetric here; d/dx1 = -d/dx2
  const int M = m.M, lane = threadIdx.x;
  double t0 = 0.0, t1 = 0.0, gz = 0.0;
  if (lane < M) {
    const int i = lane;
    for (int j = 0; j < M; ++j) {
      double kt, bz, b0, b1;
      kern_eval<KT>(zv[i] - zv[j], p, kt, bz, b0, b1);
      double ak = A[i * LD + j] * kt;
      gz -= ak * bz;
      t0 += ak * b0;
      t1 += ak * b1;
    }
    gz *= 2.0 * p.cz;
  }
  *gz_out = gz;
  th_acc[0] = wave_sum(t0) * p.c0;
  th_acc[1] = wave_sum(t1) * p.c1;
}

__device__ void k_bwd_body(const EngineView& ev, const LatentPtrs pt, unsigned mask, double* sm) {
  const int item = blockIdx.x;
  const int r = item / ev.K, k = item % ev.K;
  const LatentMeta m = ev.meta[k];
  const int M = m.M, LD = M + 1, lane = threadIdx.x;
  double* Vt = sm;             // M*LD
  double* G = Vt + M * LD;     // M*LD  (Gt, later A)
  double* B4 = G + M * LD;     // M*LD
  double* B5 = B4 + M * LD;    // M*LD
  double* gt = B5 + M * LD;    // 64
  double* ul = gt + 64;        // 64
  double* gs = ul + 64;        // 64
  double* zv = gs + 64;        // 64
  const long long base = m.offMM + (long long)r * M * M;
  const long long vbase = m.offM + (long long)r * M;
  const double* __restrict__ Li = ev.Linv + base;
  const KPar p = make_kpar(m, ev.theta + 2 * k);
  const bool want_zt = (mask & (4u | 8u)) != 0;

  // ---- gather: spike partial sums over the chunks of this trial (fixed order)
  double p1sum = 0.0, espk = 0.0, a_l = 0.0;
  if (lane < M) {
    double p0 = 0.0;
    for (int c = ev.chunk_first[r]; c < ev.chunk_first[r + 1]; ++c) {
      p0 += ev.P0[(long long)c * ev.sumM + m.colM + lane];
      if (want_zt) p1sum += ev.P1[(long long)c * ev.sumM + m.colM + lane];
    }
    a_l = ev.a[vbase + lane];
    double gsp = p.s2 * p0;       // dELL/da from the spike term
    gs[lane] = gsp;
    ul[lane] = ev.ut[vbase + lane];
    zv[lane] = ev.Zs[vbase + lane];
    espk = a_l * gsp;
  }
  // spike term of the ELL for this (trial, latent): sum_s c_s mu(t_s) = a . (s2 P0)
  espk = wave_sum(espk);
  if (lane == 0) ev.ells[item] = espk;
  if ((mask & (1u | 4u | 8u)) == 0) return;
  __syncthreads();

  // gt = dELL/d ut = quad part + Linv (dELL/da)_spikes      (a = Linv^T ut)
  if (lane < M) {
    double s = ev.gaq[vbase + lane];
    for (int q = 0; q <= lane; ++q) s += Li[lane * M + q] * gs[q];
    gt[lane] = s;
  }
  for (int e = lane; e < M * M; e += 64) {
    int i = e / M, j = e % M;
    Vt[i * LD + j] = ev.Vt[base + e];
    G[i * LD + j] = ev.GW[base + e];
  }
  __syncthreads();

  const WaveMap w = wave_map(M);
  if (mask & 1u) {
    if (lane < M) {
      double s = 0.0;
      for (int q = lane; q < M; ++q) s += Li[q * M + lane] * (gt[q] - ul[q]);
      pt.g_mean[(long long)r * M + lane] = s;
    }
    // B4 = (2 Gt - I) Vt   (Vt lower triangular: q >= j)
    if (w.act) {
      const int j = w.j;
      for (int i = w.rg; i < M; i += w.G) {
        double s = 0.0;
        for (int q = j; q < M; ++q) s += (2.0 * G[i * LD + q] - (i == q ? 1.0 : 0.0)) * Vt[q * LD + j];
        B4[i * LD + j] = s;
      }
    }
    __syncthreads();
    if (w.act) {
      const int j = w.j;
      const double* cv = pt.cholvec + (long long)r * m.P;
      double* gcv = pt.g_cholvec + (long long)r * m.P;
      for (int i = w.rg; i < M; i += w.G) {
        if (i >= j) {
          double s = 0.0;
          for (int q = i; q < M; ++q) s += Li[q * M + i] * B4[q * LD + j];
          if (i == j) s += 1.0 / cv[i * (i + 1) / 2 + i];
          gcv[i * (i + 1) / 2 + j] = s;
        }
      }
    }
    __syncthreads();
  }
  if (!want_zt) return;

  // B4 = Vt^T Gt
  if (w.act) {
    const int j = w.j;
    for (int i = w.rg; i < M; i += w.G) {
      double s = 0.0;
      for (int q = i; q < M; ++q) s += Vt[q * LD + i] * G[q * LD + j];
      B4[i * LD + j] = s;
    }
  }
  __syncthreads();
  // B5 = C = Vt B4 = Vt Vt^T Gt
  if (w.act) {
    const int j = w.j;
    for (int i = w.rg; i < M; i += w.G) {
      double s = 0.0;
      for (int q = 0; q <= i; ++q) s += Vt[i * LD + q] * B4[q * LD + j];
      B5[i * LD + j] = s;
    }
  }
  __syncthreads();
  // B4 = At
  if (w.act) {
    const int j = w.j;
    for (int i = w.rg; i < M; i += w.G) {
      double vv = 0.0;
      const int qm = (i < j) ? i : j;
      for (int q = 0; q <= qm; ++q) vv += Vt[i * LD + q] * Vt[j * LD + q];
      double v = -0.5 * (gt[i] * ul[j] + ul[i] * gt[j]) + 0.5 * ul[i] * ul[j] + G[i * LD + j] +
                 0.5 * (vv - (i == j ? 1.0 : 0.0)) - B5[i * LD + j] - B5[j * LD + i];
      B4[i * LD + j] = v;
    }
  }
  __syncthreads();
  // B5 = At Linv   (Linv lower: q >= j)
  if (w.act) {
    const int j = w.j;
    for (int i = w.rg; i < M; i += w.G) {
      double s = 0.0;
      for (int q = j; q < M; ++q) s += B4[i * LD + q] * Li[q * M + j];
      B5[i * LD + j] = s;
    }
  }
  __syncthreads();
  // G <- A = Linv^T B5
  if (w.act) {
    const int j = w.j;
    for (int i = w.rg; i < M; i += w.G) {
      double s = 0.0;
      for (int q = i; q < M; ++q) s += Li[q * M + i] * B5[q * LD + j];
      G[i * LD + j] = s;
    }
  }
  __syncthreads();

  double gz, th[2];
  if (m.ktype == SV_RBF)
    bwd_kzz_path<SV_RBF>(m, p, G, LD, zv, &gz, th);
  else
    bwd_kzz_path<SV_PERIODIC>(m, p, G, LD, zv, &gz, th);

  if ((mask & 8u) && lane < M) {
    double direct_spk = a_l * p.cz * p1sum;
    pt.g_Z[(long long)r * M + lane] = gz + ev.gZq[vbase + lane] + direct_spk;
  }
  if (lane == 0) {
    double t0 = 0.0, t1 = 0.0;
    for (int c = ev.chunk_first[r]; c < ev.chunk_first[r + 1]; ++c) {
      t0 += ev.Tpart[((long long)c * ev.K + k) * 2 + 0];
      t1 += ev.Tpart[((long long)c * ev.K + k) * 2 + 1];
    }
    ev.gthitem[item * 2 + 0] = th[0] + ev.gthq[item * 2 + 0] + p.c0 * t0;
    ev.gthitem[item * 2 + 1] = th[1] + ev.gthq[item * 2 + 1] + p.c1 * t1;
  }
}
