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

