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

// PACK items per wavefront (PACK = 2 when every M_k <= 32): each item owns LPI = 64 / PACK lanes and its own
// LDS slice; the two items of a wave are consecutive trials of the SAME latent, so control flow is uniform.
template <int PACK>
struct ItemMap {
  int k, r, l, sub;
  bool active;
};
template <int PACK>
__device__ __forceinline__ ItemMap<PACK> item_map(const EngineView& ev) {
  constexpr int LPI = 64 / PACK;
  const int rp_count = (ev.R + PACK - 1) / PACK;
  ItemMap<PACK> im;
  im.k = blockIdx.x / rp_count;
  im.sub = threadIdx.x / LPI;
  im.l = threadIdx.x % LPI;
  int r = (blockIdx.x % rp_count) * PACK + im.sub;
  im.active = r < ev.R;
  im.r = im.active ? r : ev.R - 1;
  return im;
}
template <int PACK>
__device__ __forceinline__ double item_sum(double x) {   // sum over the lanes of one item
#pragma unroll
  for (int o = 32 / PACK; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
  return x;
}
template <int PACK>
__device__ __forceinline__ WaveMap item_wave_map(int M, int l) {
  constexpr int LPI = 64 / PACK;
  int mp2 = 1;
  while (mp2 < M) mp2 <<= 1;
  WaveMap w;
  w.G = LPI / mp2;
  w.j = l % mp2;
  w.rg = l / mp2;
  w.act = w.j < M;
  return w;
}

// -------------------------------------------------------------------------------- k_build
template <int PACK>
__global__ __launch_bounds__(64) void k_build(EngineView ev) {
  extern __shared__ double sm[];
  constexpr int LPI = 64 / PACK;
  const ItemMap<PACK> im = item_map<PACK>(ev);
  const int r = im.r, k = im.k, l = im.l;
  const int item = r * ev.K + k;
  const LatentMeta m = ev.meta[k];
  const int M = m.M, LD = M + 1;
  double* A = sm + im.sub * (2 * M * LD + 64);
  double* B = A + M * LD;
  double* dinv = B + M * LD;     // 1 / L_jj
  const double* z = ev.Zs + m.offM + (long long)r * M;
  const KPar p = make_kpar(m, ev.theta + 2 * k);
  const long long base = m.offMM + (long long)r * M * M;

  for (int e = l; e < M * M; e += LPI) {
    int i = e / M, j = e % M;
    double d = z[i] - z[j];
    double v = p.s2 * (m.ktype == SV_RBF ? kern_val<SV_RBF>(d, p) : kern_val<SV_PERIODIC>(d, p));
    if (i == j) v += ev.reg;
    A[i * LD + j] = v;
    B[i * LD + j] = 0.0;
    if (im.active) ev.Kzz[base + e] = v;
  }
  __syncthreads();

  // left-looking Cholesky, lane = row; pivots through rsqrt (one Newton step keeps ~1 ulp)
  bool bad = false;
  for (int j = 0; j < M; ++j) {
    double s = 0.0;
    if (l >= j && l < M) {
      s = A[l * LD + j];
      for (int q = 0; q < j; ++q) s -= A[l * LD + q] * A[j * LD + q];
    }
    const double d = __shfl(s, im.sub * LPI + j, 64);
    if (!(d > 0.0)) bad = true;
    double ri = __builtin_amdgcn_rsq(d);          // v_rsq_f64 seed + two Newton steps
    ri = ri * fma(-0.5 * d * ri, ri, 1.5);
    ri = ri * fma(-0.5 * d * ri, ri, 1.5);
    if (l >= j && l < M) A[l * LD + j] = (l == j) ? d * ri : s * ri;
    if (l == j) dinv[j] = ri;
    __syncthreads();
  }
  if (bad && l == 0 && im.active) atomicCAS(ev.status, 0, 1 + item);

  // Linv = L^-1 by forward substitution, lane = column
  if (l < M) {
    const int c = l;
    for (int i = c; i < M; ++i) {
      double s = (i == c) ? 1.0 : 0.0;
      for (int q = c; q < i; ++q) s -= A[i * LD + q] * B[q * LD + c];
      B[i * LD + c] = s * dinv[i];
    }
  }
  __syncthreads();
  if (im.active)
    for (int e = l; e < M * M; e += LPI) {
      int i = e / M, j = e % M;
      ev.L[base + e] = (j <= i) ? A[i * LD + j] : 0.0;
      ev.Linv[base + e] = B[i * LD + j];
    }
}

// --------------------------------------------------------------------------------- k_prep
template <int PACK>
__device__ void k_prep_body(const EngineView& ev, const LatentPtrs* ptab, double* sm) {
  constexpr int LPI = 64 / PACK;
  const ItemMap<PACK> im = item_map<PACK>(ev);
  const int r = im.r, k = im.k, l = im.l;
  const int item = r * ev.K + k;
  const LatentMeta m = ev.meta[k];
  const LatentPtrs pt = ptab[k];
  const int M = m.M, LD = M + 1;
  double* Ls = sm + im.sub * (M * LD + 128);   // M*LD
  double* mv = Ls + M * LD;                    // 64
  double* uv = mv + 64;                        // 64
  const long long base = m.offMM + (long long)r * M * M;
  const long long vbase = m.offM + (long long)r * M;
  const double* __restrict__ Li = ev.Linv + base;   // global; accesses below are item-uniform or row-contiguous
  const double* cv = pt.cholvec + (long long)r * m.P;

  for (int e = l; e < M * M; e += LPI) {
    int i = e / M, j = e % M;
    Ls[i * LD + j] = (j <= i) ? cv[i * (i + 1) / 2 + j] : 0.0;
  }
  if (l < M) mv[l] = pt.mean[(long long)r * M + l];
  __syncthreads();

  double u = 0.0;
  if (l < M) {
    for (int q = 0; q <= l; ++q) u += Li[l * M + q] * mv[q];
    uv[l] = u;
    if (im.active) ev.ut[vbase + l] = u;
  }
  const double quad = item_sum<PACK>(u * u);
  __syncthreads();
  if (l < M) {
    double s = 0.0;
    for (int q = l; q < M; ++q) s += Li[q * M + l] * uv[q];
    if (im.active) ev.a[vbase + l] = s;
  }

  const WaveMap w = item_wave_map<PACK>(M, l);
  double fro = 0.0;
  if (w.act) {
    const int j = w.j;
    for (int i = w.rg; i < M; i += w.G) {
      double s = 0.0;
      for (int q = j; q <= i; ++q) s += Li[i * M + q] * Ls[q * LD + j];   // empty above the diagonal
      if (im.active) ev.Vt[base + i * M + j] = s;
      fro += s * s;
    }
  }
  fro = item_sum<PACK>(fro);
  double ld = 0.0;
  if (l < M) ld = -2.0 * log(Li[l * M + l]) - 2.0 * log(fabs(Ls[l * LD + l]));
  ld = item_sum<PACK>(ld);
  if (l == 0 && im.active) ev.klpart[item] = 0.5 * (fro + quad + ld - (double)M);
}

// ------------------------------------------------------------------------------ k_build_reg
// Register-resident build for M <= MP (MP = 16 or 32): 64 / MP (trial, latent) items per wavefront.
// Lane i of an item holds ROW i of Kzz + eps I (then of its Cholesky factor L) and COLUMN i of Linv = L^-1
// in VGPRs.  Factorisation and inverse advance together, one row j per step: both need exactly row j of L
// (L[j][q], q < j), which every lane reads from a small LDS copy with one uniform-address (broadcast)
// ds_read_b64 per value; the pivot is broadcast with a wave shuffle.  One wave per block, so the only
// synchronisation is a wave-level LDS fence; loops are fully unrolled (static register indices).
// Padding rows / columns are the identity.  write_aux = 1 additionally stores Kzz and L (only the
// KernelsMatricesStore getters read them).
// snap != 0: Z / theta come from the caller's tensors (st) and are snapshotted into engine-owned HBM on the way
// (reference semantics: later evaluations keep using the matrices of the last buildKernelsMatrices() call).
struct SnapTable {
  const double* Z[SV_MAX_K];
  const double* theta[SV_MAX_K];
};

template <int MP>
__global__ __launch_bounds__(64) void k_build_reg(EngineView ev, int write_aux, int snap, SnapTable st) {
  constexpr int IPW = 64 / MP;
  __shared__ double Lt[IPW][MP][MP];       // Lt[item][q][i] = L[i][q]  (column-major: conflict-free writes)
  const int lane = threadIdx.x, sub = lane / MP, i = lane % MP, lbase = lane - i;
  const int rp = (ev.R + IPW - 1) / IPW;
  const int k = blockIdx.x / rp;
  int r = (blockIdx.x % rp) * IPW + sub;
  const bool active = r < ev.R;
  if (!active) r = ev.R - 1;
  const LatentMeta m = ev.meta[k];
  const int M = m.M;
  const KPar p = make_kpar(m, snap ? st.theta[k] : ev.theta + 2 * k);
  const long long base = m.offMM + (long long)r * M * M;
  const double* z = (snap ? st.Z[k] : ev.Zs + m.offM) + (long long)r * M;
  const bool rowok = i < M;
  const double zi = rowok ? z[i] : 0.0;
  if (snap) {
    if (active && rowok) ev.Zs[m.offM + (long long)r * M + i] = zi;
    if (blockIdx.x % rp == 0 && lane < m.nth) ev.theta[2 * k + lane] = st.theta[k][lane];
  }

  double a[MP];
  if (m.ktype == SV_RBF) {
#pragma unroll
    for (int q = 0; q < MP; ++q) {
      const double d = zi - __shfl(zi, lbase + q, 64);
      a[q] = p.s2 * sv_exp2_neg(-(d * d) * p.h2);
      if ((q & 3) == 3) __builtin_amdgcn_sched_barrier(0);
    }
  } else {   // cos(2 pi (zi - zq) / p) by angle addition: one sincos per lane
    double si, ci;
    sv_sincos_quarter(zi * p.q4, si, ci);
#pragma unroll
    for (int q = 0; q < MP; ++q) {
      const double cs = fma(ci, __shfl(ci, lbase + q, 64), si * __shfl(si, lbase + q, 64));
      a[q] = p.s2 * sv_exp2_neg(fmin(fma(cs, p.h2, -p.h2), 0.0));
      if ((q & 3) == 3) __builtin_amdgcn_sched_barrier(0);
    }
  }
#pragma unroll
  for (int q = 0; q < MP; ++q) {
    const bool in = rowok && q < M;
    a[q] = (q == i) ? (in ? p.s2 + ev.reg : 1.0) : (in ? a[q] : 0.0);
  }
  if (write_aux && active && rowok) {
#pragma unroll
    for (int q = 0; q < MP; ++q)
      if (q < M) ev.Kzz[base + i * M + q] = a[q];
  }

  // step j: column j of L (all rows i >= j) and row j of Linv (all columns), pivots through v_rsq_f64 + 2 Newton
  double x[MP];
  bool bad = false;
  const double* Lrow = &Lt[sub][0][0];
#pragma unroll
  for (int j = 0; j < MP; ++j) {
    double s = a[j];
    double sx = (i == j) ? 1.0 : 0.0;
#pragma unroll
    for (int q = 0; q < j; ++q) {
      const double l = Lrow[q * MP + j];        // L[j][q]: same address in every lane of the item
      s = fma(-a[q], l, s);
      sx = fma(-l, x[q], sx);                   // lanes c > j: x[q] = 0 for q < c, so X[j][c] stays 0
    }
    const double d = __shfl(s, lbase + j, 64);
    if (!(d > 0.0)) bad = true;
    double ri = __builtin_amdgcn_rsq(d);
    ri = ri * fma(-0.5 * d * ri, ri, 1.5);
    ri = ri * fma(-0.5 * d * ri, ri, 1.5);
    a[j] = (i > j) ? s * ri : ((i == j) ? d * ri : 0.0);
    x[j] = sx * ri;
    Lt[sub][j][i] = a[j];
    lds_wave_fence();
    __builtin_amdgcn_sched_barrier(0);          // keep later rows' LDS reads from being hoisted (register pressure)
  }
  if (bad && i == 0 && active) atomicCAS(ev.status, 0, 1 + r * ev.K + k);

#pragma unroll
  for (int q = 0; q < MP; ++q) asm volatile("" : "+v"(x[q]));   // finish the solve before the (branchy) stores
  if (active && rowok) {
#pragma unroll
    for (int q = 0; q < MP; ++q)
      if (q < M) {
        ev.Linv[base + q * M + i] = x[q];                       // column i: lanes write consecutive addresses
        if (write_aux) ev.L[base + i * M + q] = a[q];            // row i
      }
  }
}
