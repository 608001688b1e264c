// f64 MFMA (v_mfma_f64_16x16x4_f64) versions of the time-point kernels.
//
// Everything is kept TRANSPOSED: a tile holds 16 time points on the lane index (col = lane & 15)
// and 16 inducing-point rows in the 4 accumulator registers (row = (lane >> 4) + 4 r).  In that
// layout an accumulator tile is exactly the B operand of the next MFMA that contracts over its
// row index (probe: tests/tools/mfma_f64_probe.hip), so the whole whitening chain
//     Kt^T -> Kw^T = Linv Kt^T -> T^T = Vt^T Kw^T -> Y^T = Vt T^T -> adj -> gk^T = Linv^T adj
// runs register-to-register; only the A operands (fragments of the M x M factors) come from LDS.
// kappa(t, z) itself is evaluated straight into B-operand layout (lane = time point, registers =
// inducing points), so Ktz never touches LDS or HBM.
//
// MP = 16 NT >= max_k M_k; matrices are zero padded to MP, LDS rows padded to LD = MP + 2 doubles
// (A-fragment reads are then bank-conflict free: lane -> (l & 15) * LD + (l >> 4)).
#pragma once
#include "engine_common.h"

typedef double d4 __attribute__((ext_vector_type(4)));

// D[a] = sum_b sum_r mfma(Afrag(a, b, r), B[b][r]);  Mat is lower triangular (zero padded).
// MODE 0: A = Mat   -> tiles b <= a, Afrag = Mat[16a + (l&15)][16b + 4r + (l>>4)]
// MODE 1: A = Mat^T -> tiles b >= a, Afrag = Mat[16b + 4r + (l>>4)][16a + (l&15)]
template <int NT, int MODE>
__device__ __forceinline__ void tri_chain(d4 (&D)[NT], const double* Mat, int LD, const d4 (&B)[NT],
                                          int li, int lg) {
#pragma unroll
  for (int a = 0; a < NT; ++a) {
    d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int b = 0; b < NT; ++b) {
      if (MODE == 0 ? (b <= a) : (b >= a)) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          double av = (MODE == 0) ? Mat[(16 * a + li) * LD + 16 * b + 4 * r + lg]
                                  : Mat[(16 * b + 4 * r + lg) * LD + 16 * a + li];
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, B[b][r], acc, 0, 0, 0);
        }
      }
    }
    D[a] = acc;
  }
}

__device__ __forceinline__ double row16_sum(double x) {   // sum over the 16 lanes sharing lane >> 4
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
  return x;
}
__device__ __forceinline__ double grp4_sum(double x) {    // sum over the 4 lanes sharing lane & 15
  x += __shfl_xor(x, 16, 64);
  x += __shfl_xor(x, 32, 64);
  return x;
}

template <int NT>
__device__ __forceinline__ void load_factors(const EngineView& ev, const LatentMeta& m, int r, double* Li,
                                             double* Vl, double* ul, double* zl, bool want_v) {
  constexpr int MP = 16 * NT, LD = MP + 2;
  const int M = m.M, tid = threadIdx.x;
  const long long base = m.offMM + (long long)r * M * M;
  const long long vbase = m.offM + (long long)r * M;
  for (int e = tid; e < MP * MP; e += (int)blockDim.x) {
    int i = e / MP, j = e % MP;
    bool in = (i < M) && (j < M);
    Li[i * LD + j] = in ? ev.Linv[base + i * M + j] : 0.0;
    if (want_v) Vl[i * LD + j] = in ? ev.Vt[base + i * M + j] : 0.0;
  }
  if (tid < MP) {
    ul[tid] = (tid < M) ? ev.ut[vbase + tid] : 0.0;
    zl[tid] = (tid < M) ? ev.Zs[vbase + tid] : 0.0;
  }
}

// ------------------------------------------------------------------------ k_times_fwd_mfma
// Posterior mean / variance of latent k at a CSR list of times (reference
// svPosteriorOnLatents.py:185-216, :265-300).  grid = R*K blocks of 128 or 256 threads; each wavefront
// walks tiles of 16 time points.  Output element (time gi, latent k) at gi*os_t + k*os_k.
template <int NT, int KT>
__device__ void times_fwd_body(const EngineView& ev, const int r, const int k, const double* times,
                               const long long* toff, int nper, double* mu_out, double* var_out, long long os_t,
                               long long os_k, int want_var, double* sm) {
  constexpr int MP = 16 * NT, LD = MP + 2;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6, li = lane & 15, lg = lane >> 4;
  const LatentMeta m = ev.meta[k];
  double* Li = sm;
  double* Vl = Li + MP * LD;
  double* ul = Vl + MP * LD;
  double* zl = ul + MP;
  const KPar p = make_kpar(m, ev.theta + 2 * k);
  const long long t0 = toff ? toff[r] : (long long)r * nper;
  const int nt = toff ? (int)(toff[r + 1] - toff[r]) : nper;
  load_factors<NT>(ev, m, r, Li, Vl, ul, zl, want_var != 0);
  __syncthreads();

  // periodic: sin / cos of the inducing points' phases in LDS, so that a (node, inducing point) pair costs two FMAs by
  // angle addition instead of a sincos; read at the use site (jo is an opaque lg: keeps them out of long-lived VGPRs)
  double* szl = zl + MP;
  double* czl = szl + MP;
  if (KT == SV_PERIODIC && tid < MP) sv_sincos_quarter(zl[tid] * p.q4, szl[tid], czl[tid]);
  if (KT == SV_PERIODIC) __syncthreads();
  double uj[NT][4], zj[NT][4];
#pragma unroll
  for (int a = 0; a < NT; ++a)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      uj[a][q] = ul[16 * a + 4 * q + lg];
      zj[a][q] = zl[16 * a + 4 * q + lg];
    }

  const int ntiles = (nt + 15) >> 4;
  const int NWB = blockDim.x >> 6;     // waves per block (launch parameter; 4)
  // the time of the NEXT tile is fetched one iteration ahead (global latency hidden behind the MFMA chain)
  double tq_next = (wave < ntiles) ? times[t0 + ((16 * wave + li < nt) ? 16 * wave + li : 0)] : 0.0;
  for (int tile = wave; tile < ntiles; tile += NWB) {
    const int node = 16 * tile + li;
    const bool ok = node < nt;
    const double tq = tq_next;
    {
      const int nn = node + 16 * NWB;
      if (tile + NWB < ntiles) tq_next = times[t0 + ((nn < nt) ? nn : 0)];
    }
    d4 X0[NT];
    double stq = 0.0, ctq = 0.0;
    if (KT == SV_PERIODIC) sv_sincos_quarter(tq * p.q4, stq, ctq);
    int jo = lg;
    asm volatile("" : "+v"(jo));
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (KT == SV_RBF) {
          X0[a][q] = p.s2 * kern_val<SV_RBF>(tq - zj[a][q], p);
        } else {
          const int j = 16 * a + 4 * q + jo;
          const double cs = fma(stq, szl[j], ctq * czl[j]);            // cos(2 pi (t - z) / p)
          X0[a][q] = p.s2 * sv_exp2_neg(fmin(fma(cs, p.h2, -p.h2), 0.0));
        }
      }
    d4 Kw[NT];
    tri_chain<NT, 0>(Kw, Li, LD, X0, li, lg);
    double pm = 0.0, n2 = 0.0, tt = 0.0;
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        pm = fma(uj[a][q], Kw[a][q], pm);
        n2 = fma(Kw[a][q], Kw[a][q], n2);
      }
    if (want_var) {
      d4 T[NT];
      tri_chain<NT, 1>(T, Vl, LD, Kw, li, lg);
#pragma unroll
      for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int q = 0; q < 4; ++q) tt = fma(T[a][q], T[a][q], tt);
    }
    pm = grp4_sum(pm);
    n2 = grp4_sum(n2);
    tt = grp4_sum(tt);
    if (lg == 0 && ok) {
      long long gi = t0 + node;
      mu_out[gi * os_t + k * os_k] = pm;
      if (want_var) var_out[gi * os_t + k * os_k] = (p.s2 - n2) + tt;
    }
  }
}

// One launch over all (trial, latent) items; the covariance type is resolved per block (wave-uniform branch).
template <int NT>
__global__ __launch_bounds__(256) void k_times_fwd_mfma(EngineView ev, const double* times,
                                                        const long long* toff, int nper, double* mu_out,
                                                        double* var_out, long long os_t, long long os_k,
                                                        int want_var) {
  extern __shared__ double sm[];
  const int r = blockIdx.x / ev.K, k = blockIdx.x % ev.K;
  if (ev.meta[k].ktype == SV_RBF)
    times_fwd_body<NT, SV_RBF>(ev, r, k, times, toff, nper, mu_out, var_out, os_t, os_k, want_var, sm);
  else
    times_fwd_body<NT, SV_PERIODIC>(ev, r, k, times, toff, nper, mu_out, var_out, os_t, os_k, want_var, sm);
}

// ------------------------------------------------------------------------- k_quad_bwd_mfma
// Adjoints of the quadrature term wrt ut (gt), Vt Vt^T (Gt) and -- FULL -- the direct dependence
// on Z / theta through kappa(t_q, Z).  grid = R*K blocks of 256 threads.
// (forcing 3 waves/SIMD with amdgpu_waves_per_eu spills 76 VGPRs and is 36 % slower: the FULL variant stays at 2)
template <int NT, int KT, bool FULL>
__device__ void quad_bwd_mfma_body(const EngineView& ev, const int r, const int k, double* sm) {
  constexpr int MP = 16 * NT, LD = MP + 2, NTRI = NT * (NT + 1) / 2;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6, li = lane & 15, lg = lane >> 4;
  const int item = r * ev.K + k;
  const LatentMeta m = ev.meta[k];
  const int M = m.M, Q = ev.Q, K = ev.K;
  double* Li = sm;                    // MP*LD
  double* Vl = Li + MP * LD;          // MP*LD
  double* ul = Vl + MP * LD;          // MP
  double* zl = ul + MP;               // MP
  const int NTHB = blockDim.x, NWB = NTHB >> 6;   // 2 or 4 waves per block (host: 2 when the grid is large enough)
  double* S = zl + MP;                // NWB * 16*LD  (per-wave transpose scratch; later the MP x MP Gt sum buffer)
  double* gvs = S + NWB * 16 * LD;    // NWB * 16
  double* vred = gvs + NWB * 16;      // NWB * 2 * MP  (gu / dz partials per wave)
  double* tred = vred + NWB * 2 * MP; // NWB * 2
  double* szl = tred + NWB * 2;       // MP  sin of the inducing-point phases (periodic)
  double* czl = szl + MP;             // MP  cos
  const KPar p = make_kpar(m, ev.theta + 2 * k);
  const long long base = m.offMM + (long long)r * M * M;
  const long long vbase = m.offM + (long long)r * M;
  load_factors<NT>(ev, m, r, Li, Vl, ul, zl, FULL);
  __syncthreads();
  if (KT == SV_PERIODIC && tid < MP) sv_sincos_quarter(zl[tid] * p.q4, szl[tid], czl[tid]);
  if (FULL) {
    // The adjoint of kappa(t, Z) at a node is gk = Linv^T [gm ut + 2 gv (Vt Vt^T - I) kt] = gm a + 2 gv D kt with the
    // per-item matrix D = Linv^T (Vt Vt^T - I): one dense M x M x 16 product per tile instead of three triangular ones
    // (T = Vt^T kt, Vt T, Linv^T .).  D is built once per item, one 16-column panel per wavefront, and replaces Vt in LDS.
    d4 Dp[NT];
    const int c = wave;
    if (c < NT) {
      d4 P[NT], VV[NT];
#pragma unroll
      for (int b = 0; b < NT; ++b)
#pragma unroll
        for (int q = 0; q < 4; ++q) P[b][q] = Vl[(16 * c + li) * LD + 16 * b + 4 * q + lg];   // Vt^T panel
      tri_chain<NT, 0>(VV, Vl, LD, P, li, lg);
#pragma unroll
      for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (16 * a + 4 * q + lg == 16 * c + li) VV[a][q] -= 1.0;
      tri_chain<NT, 1>(Dp, Li, LD, VV, li, lg);
    }
    if (tid < MP) ul[tid] = (tid < M) ? ev.a[vbase + tid] : 0.0;     // a = Linv^T ut takes the place of ut
    __syncthreads();                                                   // every wave is done reading Vt
    if (c < NT) {
#pragma unroll
      for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int q = 0; q < 4; ++q) Vl[(16 * a + 4 * q + lg) * LD + 16 * c + li] = Dp[a][q];
    }
  }
  __syncthreads();
  d4 gu[NT], dz[NT], GW[NTRI];
#pragma unroll
  for (int a = 0; a < NT; ++a) {
    gu[a] = (d4){0.0, 0.0, 0.0, 0.0};
    dz[a] = (d4){0.0, 0.0, 0.0, 0.0};
  }
#pragma unroll
  for (int t = 0; t < NTRI; ++t) GW[t] = (d4){0.0, 0.0, 0.0, 0.0};
  double th0 = 0.0, th1 = 0.0;
  double* Sw = S + wave * 16 * LD;
  double* gvw = gvs + wave * 16;
  // periodic: kappa and its derivative bases follow from the node's own sin / cos and the inducing
  // points' (szl, czl) by angle addition (no per-pair sincos, no second exp).  Per-j constants are read
  // from LDS at their use sites (jo is an opaque 0: it keeps them out of long-lived registers).
  const double inv_s2 = 1.0 / p.s2;

  const int ntiles = (Q + 15) >> 4;
  // inputs of the NEXT tile are fetched one iteration ahead (global latency hidden behind the MFMA chain)
  double tq_n = 0.0, gm_n = 0.0, gv_n = 0.0;
  if (wave < ntiles) {
    const int nn = 16 * wave + li;
    const long long gi = (long long)r * Q + ((nn < Q) ? nn : 0);
    tq_n = ev.tq[gi];
    gm_n = (nn < Q) ? ev.gmuq[gi * K + k] : 0.0;
    gv_n = (nn < Q) ? ev.gvarq[gi * K + k] : 0.0;
  }
  for (int tile = wave; tile < ntiles; tile += NWB) {
    const double tq = tq_n, gm = gm_n, gvv = gv_n;
    if (tile + NWB < ntiles) {
      const int nn = 16 * (tile + NWB) + li;
      const long long gi = (long long)r * Q + ((nn < Q) ? nn : 0);
      tq_n = ev.tq[gi];
      gm_n = (nn < Q) ? ev.gmuq[gi * K + k] : 0.0;
      gv_n = (nn < Q) ? ev.gvarq[gi * K + k] : 0.0;
    }
    double stq = 0.0, ctq = 0.0;
    if (KT == SV_PERIODIC) sv_sincos_quarter(tq * p.q4, stq, ctq);
    int jo = lg;
    asm volatile("" : "+v"(jo));
    d4 X0[NT];
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int j = 16 * a + 4 * q + jo;
        if (KT == SV_RBF) {
          const double dd = tq - zl[j];
          X0[a][q] = p.s2 * sv_exp2_neg(-(dd * dd) * p.h2);
        } else {
          const double cs = fma(stq, szl[j], ctq * czl[j]);
          X0[a][q] = p.s2 * sv_exp2_neg(fma(cs, p.h2, -p.h2));
        }
      }
    d4 Kw[NT];
    tri_chain<NT, 0>(Kw, Li, LD, X0, li, lg);
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
      for (int q = 0; q < 4; ++q) gu[a][q] = fma(gm, Kw[a][q], gu[a][q]);
    if (FULL) {
      d4 Gk[NT];
#pragma unroll
      for (int a = 0; a < NT; ++a) {
        d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int b = 0; b < NT; ++b)
#pragma unroll
          for (int q = 0; q < 4; ++q)
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Vl[(16 * a + li) * LD + 16 * b + 4 * q + lg], Kw[b][q], acc, 0, 0, 0);
        Gk[a] = acc;                                            // D kt
      }
#pragma unroll
      for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int q = 0; q < 4; ++q) Gk[a][q] = fma(2.0 * gvv, Gk[a][q], gm * ul[16 * a + 4 * q + jo]);
      asm volatile("" : "+v"(jo));
#pragma unroll
      for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int j = 16 * a + 4 * q + jo;
          const double gkk = Gk[a][q] * (X0[a][q] * inv_s2);    // gk * ktil
          const double dd = tq - zl[j];
          if (KT == SV_RBF) {
            const double gd = gkk * dd;
            dz[a][q] += gd;                                    // bz = delta
            th0 = fma(gd, dd, th0);                            // b0 = delta^2
          } else {
            const double cs = fma(stq, szl[j], ctq * czl[j]);
            const double sn = fma(stq, czl[j], -(ctq * szl[j]));
            const double gs = gkk * sn;
            dz[a][q] += gs;                                    // bz = sin(2 pi delta / p)
            th0 = fma(gkk, 0.5 - 0.5 * cs, th0);               // b0 = sin^2(pi delta / p)
            th1 = fma(gs, dd, th1);                            // b1 = delta sin(2 pi delta / p)
          }
        }
    }
    // Gt += Kw^T diag(gvar) Kw : contraction over the time index -> transpose through LDS
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
      for (int q = 0; q < 4; ++q) Sw[li * LD + 16 * a + 4 * q + lg] = Kw[a][q];
    if (lg == 0) gvw[li] = gvv;
    lds_wave_fence();
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int nc = 4 * c + lg;
      const double g = gvw[nc];
      double aop[NT];
#pragma unroll
      for (int a = 0; a < NT; ++a) aop[a] = Sw[nc * LD + 16 * a + li];
      int t = 0;
#pragma unroll
      for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int b = 0; b <= a; ++b) {
          GW[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(aop[a], g * aop[b], GW[t], 0, 0, 0);
          ++t;
        }
    }
    lds_wave_fence();
  }

  // ---- reductions: over the 16 time lanes, then over the wavefronts (fixed order)
#pragma unroll
  for (int a = 0; a < NT; ++a)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      double g = row16_sum(gu[a][q]);
      double z = FULL ? row16_sum(dz[a][q]) : 0.0;
      if (li == 0) {
        vred[(wave * 2 + 0) * MP + 16 * a + 4 * q + lg] = g;
        vred[(wave * 2 + 1) * MP + 16 * a + 4 * q + lg] = z;
      }
    }
  th0 = wave_sum(th0);
  th1 = wave_sum(th1);
  if (lane == 0) {
    tred[wave * 2 + 0] = th0;
    tred[wave * 2 + 1] = th1;
  }
  __syncthreads();   // all waves done with their scratch: S becomes the MP x MP sum buffer
  for (int w = 0; w < NWB; ++w) {
    if (wave == w) {
      int t = 0;
#pragma unroll
      for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int b = 0; b <= a; ++b) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            int row = 16 * a + lg + 4 * q, col = 16 * b + li;
            if (w == 0)
              S[row * MP + col] = GW[t][q];
            else
              S[row * MP + col] += GW[t][q];
          }
          ++t;
        }
    }
    __syncthreads();
  }
  if (tid < M) {
    double g = vred[tid];
    for (int w = 1; w < NWB; ++w) g += vred[(2 * w) * MP + tid];
    ev.gaq[vbase + tid] = g;
    if (FULL) {
      double z = vred[MP + tid];
      for (int w = 1; w < NWB; ++w) z += vred[(2 * w + 1) * MP + tid];
      ev.gZq[vbase + tid] = p.cz * z;
    }
  }
  for (int e = tid; e < M * M; e += NTHB) {
    int i = e / M, j = e % M;
    ev.GW[base + e] = ((i >> 4) >= (j >> 4)) ? S[i * MP + j] : S[j * MP + i];
  }
  if (FULL && tid == 0) {
    double t0s = tred[0], t1s = tred[1];
    for (int w = 1; w < NWB; ++w) {
      t0s += tred[2 * w];
      t1s += tred[2 * w + 1];
    }
    ev.gthq[item * 2 + 0] = p.c0 * t0s;
    ev.gthq[item * 2 + 1] = p.c1 * t1s;
  }
}

template <int NT, bool FULL>
__global__ __launch_bounds__(256, (NT <= 2 ? 2 : 1)) void k_quad_bwd_mfma(EngineView ev) {   // M <= 32: keep 2 waves / SIMD
  extern __shared__ double sm[];
  const int r = blockIdx.x / ev.K, k = blockIdx.x % ev.K;
  if (ev.meta[k].ktype == SV_RBF)
    quad_bwd_mfma_body<NT, SV_RBF, FULL>(ev, r, k, sm);
  else
    quad_bwd_mfma_body<NT, SV_PERIODIC, FULL>(ev, r, k, sm);
}

// ------------------------------------------------------------------------------ k_bwd_mfma
// Per-(trial, latent) backward through the whitened algebra (derivation: kernels_item.h header).
// NT wavefronts per item (block = 64 NT threads): wave c owns the 16-column panel c of every M x M product, all on
// the f64 MFMA with the left operand read from LDS as A fragments and the right operand held as register tiles.
// The panels only meet at the LDS hand-offs (C, At) and in the fixed-order sums of the dZ / dtheta partials.
template <int NT, int KT>
__device__ void bwd_mfma_body(const EngineView& ev, const int r, const int k, const LatentPtrs pt, unsigned mask,
                              double* sm) {
  constexpr int MP = 16 * NT, LD = MP + 2, NTH = 64 * NT;
  const int tid = threadIdx.x, lane = tid & 63, c = tid >> 6, li = lane & 15, lg = lane >> 4;
  const int item = r * ev.K + k;
  const LatentMeta m = ev.meta[k];
  const int M = m.M;
  double* Li = sm;               // MP*LD
  double* Vl = Li + MP * LD;     // MP*LD  Vt; once C and Vt Vt^T are in registers it is reused for C
  double* Gl = Vl + MP * LD;     // MP*LD  Gt, later At
  double* Cs = Vl;
  double* gt = Gl + MP * LD;     // MP
  double* ul = gt + MP;          // MP
  double* gs = ul + MP;          // MP
  double* zl = gs + MP;          // MP
  double* redz = zl + MP;        // NT*MP  per-panel row sums of the dZ contraction
  double* redt = redz + NT * MP; // NT*2   per-panel dtheta partials
  const long long base = m.offMM + (long long)r * M * M;
  const long long vbase = m.offM + (long long)r * M;
  const KPar p = make_kpar(m, ev.theta + 2 * k);
  const bool want_zt = (mask & (4u | 8u)) != 0;

  // ---- gather the spike partial sums of this trial (fixed chunk order); thread j < M owns inducing point j
  double p1sum = 0.0, espk = 0.0, a_l = 0.0;
  for (int i = tid; i < MP; i += NTH) {
    gs[i] = 0.0; ul[i] = 0.0; zl[i] = 0.0; gt[i] = 0.0;
  }
  const int c0 = ev.chunk_first[r], c1 = ev.chunk_first[r + 1];
  if (tid < M) {
    double p0 = 0.0;
    for (int ch = c0; ch < c1; ch += 4) {          // 4 chunks per step: independent loads in flight, fixed order
      double v[4], w[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const bool in = ch + u < c1;
        const long long o = (long long)(in ? ch + u : ch) * ev.sumM + m.colM + tid;
        v[u] = ev.P0[o];
        w[u] = want_zt ? ev.P1[o] : 0.0;
        if (!in) { v[u] = 0.0; w[u] = 0.0; }
      }
      p0 += (v[0] + v[1]) + (v[2] + v[3]);
      p1sum += (w[0] + w[1]) + (w[2] + w[3]);
    }
    a_l = ev.a[vbase + tid];
    double gsp = p.s2 * p0;
    espk = a_l * gsp;
    gs[tid] = gsp;
    ul[tid] = ev.ut[vbase + tid];
    zl[tid] = ev.Zs[vbase + tid];
  }
  // spike-direct theta partials: one chunk per lane of wave 0, then a fixed-order wave tree
  double tp0 = 0.0, tp1 = 0.0;
  if (c == 0) {
    espk = wave_sum(espk);
    if (want_zt) {
      for (int ch = c0 + lane; ch < c1; ch += 64) {
        tp0 += ev.Tpart[((long long)ch * ev.K + k) * 2 + 0];
        tp1 += ev.Tpart[((long long)ch * ev.K + k) * 2 + 1];
      }
      tp0 = wave_sum(tp0);
      tp1 = wave_sum(tp1);
    }
    if (lane == 0) ev.ells[item] = espk;
  }
  if ((mask & (1u | 4u | 8u)) == 0) return;

  for (int e = tid; e < MP * MP; e += NTH) {
    int i = e / MP, j = e % MP;
    bool in = (i < M) && (j < M);
    Li[i * LD + j] = in ? ev.Linv[base + i * M + j] : 0.0;
    Vl[i * LD + j] = in ? ev.Vt[base + i * M + j] : 0.0;
    Gl[i * LD + j] = in ? ev.GW[base + i * M + j] : 0.0;
  }
  __syncthreads();
  if (tid < M) {
    double s = ev.gaq[vbase + tid];
    for (int q = 0; q <= tid; ++q) s += Li[tid * LD + q] * gs[q];
    gt[tid] = s;
  }
  __syncthreads();

  if (mask & 1u) {
    if (tid < M) {
      double s = 0.0;
      for (int q = tid; q < M; ++q) s += Li[q * LD + tid] * (gt[q] - ul[q]);
      pt.g_mean[(long long)r * M + tid] = s;
    }
    // dL_S = tril(Linv^T (2 Gt - I) Vt) + diag(1 / L_S,ii): this wave's 16-column panel
    const double* cv = pt.cholvec + (long long)r * m.P;
    double* gcv = pt.g_cholvec + (long long)r * m.P;
    d4 P[NT], X[NT];
#pragma unroll
    for (int b = 0; b < NT; ++b)
#pragma unroll
      for (int q = 0; q < 4; ++q) P[b][q] = Vl[(16 * b + 4 * q + lg) * LD + 16 * c + li];
#pragma unroll
    for (int a = 0; a < NT; ++a) {
      d4 acc = {0.0, 0.0, 0.0, 0.0};
      if (a >= c) {
#pragma unroll
        for (int b = 0; b < NT; ++b) {
          if (b >= c) {   // Vt lower triangular: panel c has zero tiles above the diagonal
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              int row = 16 * a + li, col = 16 * b + 4 * q + lg;
              double av = 2.0 * Gl[row * LD + col] - (row == col ? 1.0 : 0.0);
              acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, P[b][q], acc, 0, 0, 0);
            }
          }
        }
      }
      X[a] = acc;
    }
#pragma unroll
    for (int a = 0; a < NT; ++a) {
      if (a >= c) {
        d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int b = a; b < NT; ++b)
#pragma unroll
          for (int q = 0; q < 4; ++q)
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Li[(16 * b + 4 * q + lg) * LD + 16 * a + li], X[b][q], acc, 0, 0, 0);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          int i = 16 * a + 4 * q + lg, j = 16 * c + li;
          if (i < M && j <= i) {
            double g = acc[q];
            if (i == j) g += 1.0 / cv[i * (i + 1) / 2 + i];
            gcv[i * (i + 1) / 2 + j] = g;
          }
        }
      }
    }
  }
  if (!want_zt) return;

  // ---- panel c of C = Vt Vt^T Gt and of VV = Vt Vt^T, kept in registers until Vt is dead in every wave
  d4 VV[NT], Cm[NT];
  {
    d4 P[NT], Bm[NT];
#pragma unroll
    for (int b = 0; b < NT; ++b)
#pragma unroll
      for (int q = 0; q < 4; ++q) P[b][q] = Gl[(16 * b + 4 * q + lg) * LD + 16 * c + li];
    tri_chain<NT, 1>(Bm, Vl, LD, P, li, lg);
    tri_chain<NT, 0>(Cm, Vl, LD, Bm, li, lg);
#pragma unroll
    for (int b = 0; b < NT; ++b)
#pragma unroll
      for (int q = 0; q < 4; ++q) P[b][q] = Vl[(16 * c + li) * LD + 16 * b + 4 * q + lg];   // Vt^T panel
    tri_chain<NT, 0>(VV, Vl, LD, P, li, lg);
  }
  __syncthreads();
#pragma unroll
  for (int a = 0; a < NT; ++a)
#pragma unroll
    for (int q = 0; q < 4; ++q) Cs[(16 * a + 4 * q + lg) * LD + 16 * c + li] = Cm[a][q];
  __syncthreads();
  // ---- At (in place over Gt): each element is read and written by exactly one lane; the transposed reads of Gt
  //      itself do not occur (only C is read transposed), so no wave sees a half-updated panel
#pragma unroll
  for (int a = 0; a < NT; ++a)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      int i = 16 * a + 4 * q + lg, j = 16 * c + li;
      double v = -0.5 * (gt[i] * ul[j] + ul[i] * gt[j]) + 0.5 * ul[i] * ul[j] + Gl[i * LD + j] +
                 0.5 * (VV[a][q] - (i == j ? 1.0 : 0.0)) - Cs[i * LD + j] - Cs[j * LD + i];
      Gl[i * LD + j] = v;
    }
  __syncthreads();
  // ---- panel c of A = Linv^T At Linv, contracted on the fly with dKzz/dZ and dKzz/dtheta
  {
    d4 P[NT], Dm[NT], Am[NT];
#pragma unroll
    for (int b = 0; b < NT; ++b)
#pragma unroll
      for (int q = 0; q < 4; ++q) P[b][q] = Li[(16 * b + 4 * q + lg) * LD + 16 * c + li];
#pragma unroll
    for (int a = 0; a < NT; ++a) {
      d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int b = 0; b < NT; ++b)
        if (b >= c) {   // Linv lower triangular
#pragma unroll
          for (int q = 0; q < 4; ++q)
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Gl[(16 * a + li) * LD + 16 * b + 4 * q + lg], P[b][q], acc, 0, 0, 0);
        }
      Dm[a] = acc;
    }
    tri_chain<NT, 1>(Am, Li, LD, Dm, li, lg);
    const double zc = zl[16 * c + li];
    double t0 = 0.0, t1 = 0.0;
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        int i = 16 * a + 4 * q + lg, j = 16 * c + li;
        double gz = 0.0;
        if (i < M && j < M) {
          double kt, bz, b0, b1;
          kern_eval<KT>(zl[i] - zc, p, kt, bz, b0, b1);
          double ak = Am[a][q] * kt;
          gz = -(ak * bz);
          t0 = fma(ak, b0, t0);
          t1 = fma(ak, b1, t1);
        }
        gz = row16_sum(gz);
        if (li == 0) redz[c * MP + i] = gz;
      }
    t0 = wave_sum(t0);
    t1 = wave_sum(t1);
    if (lane == 0) {
      redt[2 * c + 0] = t0;
      redt[2 * c + 1] = t1;
    }
  }
  __syncthreads();
  if ((mask & 8u) && tid < M) {
    double g = 0.0;
#pragma unroll
    for (int w = 0; w < NT; ++w) g += redz[w * MP + tid];     // fixed panel order
    // direct spike part a_j cz P1_j is already on this thread
    pt.g_Z[(long long)r * M + tid] = g * 2.0 * p.cz + ev.gZq[vbase + tid] + a_l * p.cz * p1sum;
  }
  if (tid == 0) {
    double t0 = 0.0, t1 = 0.0;
#pragma unroll
    for (int w = 0; w < NT; ++w) {
      t0 += redt[2 * w + 0];
      t1 += redt[2 * w + 1];
    }
    ev.gthitem[item * 2 + 0] = t0 * p.c0 + ev.gthq[item * 2 + 0] + p.c0 * tp0;
    ev.gthitem[item * 2 + 1] = t1 * p.c1 + ev.gthq[item * 2 + 1] + p.c1 * tp1;
  }
}

// ----------------------------------------------------------------------------- k_prep_mfma
// ut = Linv m, a = Linv^T ut, Vt = Linv L_S and the KL term per (trial, latent), on the f64 MFMA.
// NT wavefronts per item: wave c produces the 16-column panel c of Vt; wave 0 also carries the vector m as an
// extra panel (column 0) so that ut and a come out of the same tri_chain.  Linv and L_S are staged in LDS with
// coalesced loads by the whole block.
template <int NT>
__global__ __launch_bounds__(64 * NT) void k_prep_mfma(EngineView ev, LatentPtrTable ptab) {
  extern __shared__ double sm[];
  constexpr int MP = 16 * NT, LD = MP + 2, NTH = 64 * NT;
  const int item = blockIdx.x, tid = threadIdx.x, lane = tid & 63, c = tid >> 6, li = lane & 15, lg = lane >> 4;
  const int r = item / ev.K, k = item % ev.K;
  const LatentMeta m = ev.meta[k];
  const LatentPtrs pt = ptab.p[k];
  const int M = m.M;
  double* Li = sm;               // MP*LD
  double* Ls = Li + MP * LD;     // MP*LD
  double* red = Ls + MP * LD;    // NT
  const long long base = m.offMM + (long long)r * M * M;
  const long long vbase = m.offM + (long long)r * M;
  const double* cv = pt.cholvec + (long long)r * m.P;
  for (int e = tid; e < MP * MP; e += NTH) {
    int i = e / MP, j = e % MP;
    bool in = (i < M) && (j < M);
    Li[i * LD + j] = in ? ev.Linv[base + i * M + j] : 0.0;
    Ls[i * LD + j] = (in && j <= i) ? cv[i * (i + 1) / 2 + j] : 0.0;
  }
  __syncthreads();

  // panel c of L_S -> Vt ; |Vt|_F^2
  double part = 0.0;
  {
    d4 P[NT], V[NT];
#pragma unroll
    for (int b = 0; b < NT; ++b)
#pragma unroll
      for (int q = 0; q < 4; ++q) P[b][q] = Ls[(16 * b + 4 * q + lg) * LD + 16 * c + li];
    tri_chain<NT, 0>(V, Li, LD, P, li, lg);
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        int i = 16 * a + 4 * q + lg, j = 16 * c + li;
        if (i < M && j < M) ev.Vt[base + i * M + j] = V[a][q];
        part = fma(V[a][q], V[a][q], part);
      }
  }
  if (c == 0) {
    // the mean as column 0 of an extra panel: ut = Linv m, a = Linv^T ut
    d4 Pm[NT], U[NT], A[NT];
#pragma unroll
    for (int b = 0; b < NT; ++b)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        int i = 16 * b + 4 * q + lg;
        Pm[b][q] = (li == 0 && i < M) ? pt.mean[(long long)r * M + i] : 0.0;
      }
    tri_chain<NT, 0>(U, Li, LD, Pm, li, lg);
    tri_chain<NT, 1>(A, Li, LD, U, li, lg);
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        int i = 16 * a + 4 * q + lg;
        if (li == 0 && i < M) {
          ev.ut[vbase + i] = U[a][q];
          ev.a[vbase + i] = A[a][q];
          part = fma(U[a][q], U[a][q], part);
        }
      }
    if (lane < M) part += -2.0 * log(Li[lane * LD + lane]) - 2.0 * log(fabs(Ls[lane * LD + lane]));
  }
  part = wave_sum(part);
  if (lane == 0) red[c] = part;
  __syncthreads();
  if (tid == 0) {
    double tot = 0.0;
#pragma unroll
    for (int w = 0; w < NT; ++w) tot += red[w];     // fixed panel order
    ev.klpart[item] = 0.5 * (tot - (double)M);
  }
}

// ------------------------------------------------------------------------- k_quad_ell_mfma
// Quadrature term of the expected log-likelihood in GEMM form (reference svEmbedding.py:80-84,
// expectedLogLikelihood.py:107-135,205-208), per trial:
//   h = [mu | var] (Q x 2K) . [C | C^2/2]^T (2K x N) + d        E = w_q exp(h)          ELL_quad = -sum E
//   ROLE 0 (n-tile per wave, loop q-tiles): [A1 | A2]^T = [mu | var]^T E -> dC = -(A1 + C A2), dd = -colsum E
//   ROLE 1 (q-tile per wave, loop n-tiles): [gmu | gvar]^T = -[C | C^2/2]^T E^T   (latent adjoints)
// A 16x16 tile of E lives in accumulator layout and is the B operand of the sum products as is; role 1
// recomputes the transposed tile instead of transposing through LDS.  log2(e) is folded into the LDS copy
// of [C | C^2/2] and d, so that E = w 2^h.  The LOOPED operand matrix of the whole trial sits in LDS
// (rows padded to LDK = 4 KS + 2 doubles: conflict-free fragment reads); the wave's own 16 rows of the
// other matrix sit in a small per-wave LDS slice.  When the looped matrix exceeds the LDS budget (large N K or
// Q K) it is streamed through LDS in chunks of lc tiles.  grid = R * parts blocks of 64 NW threads (NW = 4 waves; the
// count is a launch parameter, larger blocks measured slower because fewer of them fit a CU).
template <int KS, int ROLE>
__device__ __forceinline__ void quad_ell_mfma_body(const EngineView& ev, const double* C, const double* d,
                                                   int want_sums, int write_ell, int parts, int lc, int blk,
                                                   double* sm) {
  constexpr int KKP = 4 * KS, LDK = KKP + 2, KKT = (KKP + 15) / 16;
  const double L2E = 1.4426950408889634074, LN2 = 0.69314718055994530942;
  const int K = ev.K, N = ev.N, Q = ev.Q, K2 = 2 * K;
  const int QT = (Q + 15) >> 4, NTl = (N + 15) >> 4;
  const int LT = (ROLE == 0) ? QT : NTl;      // tiles of the looped dimension
  const int OT = (ROLE == 0) ? NTl : QT;      // tiles of the wave-owned dimension
  const int Lp = LT * 16;
  const int LC = (lc < LT) ? lc : LT;         // looped tiles resident in LDS at a time (LC < LT: large N K or Q K)
  const int Lrows = LC * 16;
  const int r = blk / parts, part = blk % parts;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lg = lane >> 4;
  const int NTH = blockDim.x, NW = NTH >> 6;   // waves per block (host launch parameter)
  double* LM = sm;                          // (Lrows + 1) * LDK  looped matrix: ROLE 0 [mu | var], ROLE 1 log2e [C | C^2/2]
  double* lv = LM + (Lrows + 1) * LDK;      // Lp   looped vector: ROLE 0 w_q, ROLE 1 log2e d_n
  double* OW = lv + Lp;                     // NW * 16 * LDK  per-wave rows of the other matrix
  double* red = OW + NW * 16 * LDK;         // 8
  const int lrows_tot = (ROLE == 0) ? Q : N;

  // fill rows [row0, row0 + Lrows) of the looped matrix: zero the padding, then batched independent loads
  // (8 in flight per thread) over the contiguous sources mu/var [Q x K] (ROLE 0) or C [N x K] (ROLE 1)
  auto fill_lm = [&](int row0) {
    for (int e = tid; e < (Lrows + 1) * LDK; e += NTH) LM[e] = 0.0;
    __syncthreads();
    const int nrow = (lrows_tot - row0 < Lrows) ? (lrows_tot - row0) : Lrows;
    const int tot = nrow * K;
    const double* s0 = ((ROLE == 0) ? ev.muq + (long long)r * Q * K : C) + (long long)row0 * K;
    const double* s1 = ((ROLE == 0) ? ev.varq + (long long)r * Q * K : C) + (long long)row0 * K;
    for (int b0 = 0; b0 < tot; b0 += NTH * 8) {
      double v0[8], v1[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int idx = b0 + u * NTH + tid;
        const bool ok = idx < tot;
        v0[u] = ok ? s0[idx] : 0.0;
        v1[u] = (ROLE == 0) ? (ok ? s1[idx] : 0.0) : v0[u];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int idx = b0 + u * NTH + tid;
        if (idx < tot) {
          const int row = idx / K, c = idx - row * K;
          if (ROLE == 0) {
            LM[row * LDK + c] = v0[u];
            LM[row * LDK + K + c] = v1[u];
          } else {
            LM[row * LDK + c] = v0[u] * L2E;
            LM[row * LDK + K + c] = 0.5 * v0[u] * v0[u] * L2E;
          }
        }
      }
    }
    __syncthreads();
  };

  for (int i = tid; i < Lp; i += NTH)
    lv[i] = (ROLE == 0) ? ((i < Q) ? ev.wq[(long long)r * Q + i] : 0.0) : ((i < N) ? d[i] * L2E : 0.0);
  if (LC == LT) fill_lm(0);     // common case: the whole looped matrix stays resident
  else __syncthreads();

  double esum = 0.0;
  double* ow = OW + wave * 16 * LDK;
  for (int ot0 = part * NW; ot0 < OT; ot0 += parts * NW) {     // block-uniform rounds; wave `wave` owns tile ot0 + wave
    const int ot = ot0 + wave;
    const bool active = ot < OT;
    // own 16 rows -> per-wave LDS slice (wave-local: no block barrier needed); the source rows are
    // contiguous (16 K doubles): batched loads again
    for (int e = lane; e < 16 * LDK; e += 64) ow[e] = 0.0;
    lds_wave_fence();
    if (active) {
      const int orows = (ROLE == 0) ? N : Q;
      const int nrow = (orows - ot * 16 < 16) ? (orows - ot * 16) : 16;
      const int tot = nrow * K;
      const double* s0 = (ROLE == 0) ? C + (long long)ot * 16 * K : ev.muq + ((long long)r * Q + ot * 16) * K;
      const double* s1 = (ROLE == 0) ? s0 : ev.varq + ((long long)r * Q + ot * 16) * K;
      double v0[8], v1[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int idx = u * 64 + lane;
        const bool ok = idx < tot;
        v0[u] = ok ? s0[idx] : 0.0;
        v1[u] = (ROLE == 0) ? v0[u] : (ok ? s1[idx] : 0.0);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int idx = u * 64 + lane;
        if (idx < tot) {
          const int row = idx / K, c = idx - row * K;
          if (ROLE == 0) {
            ow[row * LDK + c] = v0[u] * L2E;
            ow[row * LDK + K + c] = 0.5 * v0[u] * v0[u] * L2E;
          } else {
            ow[row * LDK + c] = v0[u];
            ow[row * LDK + K + c] = v1[u];
          }
        }
      }
    }
    lds_wave_fence();
    const int oi = ot * 16 + li;              // own index of this lane's column
    const bool ook = active && oi < ((ROLE == 0) ? N : Q);
    const double ov = !ook ? 0.0 : ((ROLE == 0) ? d[oi] * L2E : ev.wq[(long long)r * Q + oi]);
    double bfrag[KS];                         // own rows as the B operand of the h product
#pragma unroll
    for (int c = 0; c < KS; ++c) bfrag[c] = ow[li * LDK + 4 * c + lg];
    d4 acc[KKT];
#pragma unroll
    for (int t = 0; t < KKT; ++t) acc[t] = (d4){0.0, 0.0, 0.0, 0.0};
    double colsum = 0.0;
    for (int lt0 = 0; lt0 < LT; lt0 += LC) {
      if (LC < LT) fill_lm(lt0 * 16);         // block-uniform: every wave takes part, active or not
      const int lt1 = (lt0 + LC < LT) ? lt0 + LC : LT;
      if (active)
        for (int lt = lt0; lt < lt1; ++lt) {
          const int lr = (lt - lt0) * 16;     // row of this tile inside the resident chunk
          d4 h = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
          for (int c = 0; c < KS; ++c)
            h = __builtin_amdgcn_mfma_f64_16x16x4f64(LM[(lr + li) * LDK + 4 * c + lg], bfrag[c], h, 0, 0, 0);
          double afrag[KKT][4];               // looped rows, transposed, for the sum product (independent of E)
          if (want_sums) {
#pragma unroll
            for (int t = 0; t < KKT; ++t)
#pragma unroll
              for (int c = 0; c < 4; ++c) afrag[t][c] = LM[(lr + 4 * c + lg) * LDK + 16 * t + li];
          }
          d4 E;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int l = lt * 16 + lg + 4 * q;   // looped index of this element's row
            // ROLE 0: E = w_q 2^(h + d_n) ; ROLE 1: E^T = w_q 2^(h^T + d_n)
            const double x = (ROLE == 0) ? lv[l] * sv_exp2(h[q] + ov) : ov * sv_exp2(h[q] + lv[l]);
            const bool lok = l < lrows_tot;
            E[q] = (ook && lok) ? x : 0.0;
            colsum += E[q];
          }
          if (want_sums) {
#pragma unroll
            for (int t = 0; t < KKT; ++t)
#pragma unroll
              for (int c = 0; c < 4; ++c)
                acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(afrag[t][c], E[c], acc[t], 0, 0, 0);
          }
        }
      if (LC < LT) __syncthreads();           // all waves done with this chunk before it is overwritten
    }
    esum += colsum;
    if (want_sums && ook) {
      if (ROLE == 0) {
#pragma unroll
        for (int t = 0; t < KKT; ++t)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int kk = 16 * t + lg + 4 * q;
            if (kk < K2) ev.gCpart[((long long)r * N + oi) * K2 + kk] = acc[t][q];
          }
      } else {
        const long long gi = ((long long)r * Q + oi) * K;
#pragma unroll
        for (int t = 0; t < KKT; ++t)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int kk = 16 * t + lg + 4 * q;
            if (kk < K) ev.gmuq[gi + kk] = -acc[t][q] * LN2;
            else if (kk < K2) ev.gvarq[gi + kk - K] = -acc[t][q] * LN2;
          }
      }
    }
    if (ROLE == 0 && want_sums) {
      colsum = grp4_sum(colsum);
      if (lg == 0 && ook) ev.gdpart[(long long)r * N + oi] = -colsum;
    }
    lds_wave_fence();
  }
  if (write_ell) {
    const double tot = block_sum(esum, red);
    if (tid == 0) ev.ellq[blk] = -tot;
  }
}

// ROLE 0: dC / dd / ELL only (no latent-side gradient wanted).  ROLE 2: both roles in one launch (blocks [0, nb) role 0,
// [nb, 2 nb) role 1): measured faster than two launches at every size.
template <int KS, int ROLE>
__global__ __launch_bounds__(512) void k_quad_ell_mfma(EngineView ev, const double* C, const double* d,
                                                       int want_sums, int write_ell, int parts, int lc0, int lc1) {
  extern __shared__ double sm[];
  if (ROLE == 2) {
    const int nb = ev.R * parts;
    if ((int)blockIdx.x < nb)
      quad_ell_mfma_body<KS, 0>(ev, C, d, want_sums, 1, parts, lc0, blockIdx.x, sm);
    else
      quad_ell_mfma_body<KS, 1>(ev, C, d, 1, 0, parts, lc1, blockIdx.x - nb, sm);
  } else {
    quad_ell_mfma_body<KS, (ROLE == 2 ? 0 : ROLE)>(ev, C, d, want_sums, write_ell, parts, ROLE == 0 ? lc0 : lc1,
                                                    blockIdx.x, sm);
  }
}
