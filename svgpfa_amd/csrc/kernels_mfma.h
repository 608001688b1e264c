// f64 MFMA (v_mfma_f64_16x16x4_f64) versions of the time-point kernels.
//
// Everything is kept TRANSPOSED: a tile holds 16 time points on the lane index (col = lane & 15)
// and 16 inducing-point rows in the 4 accumulator registers (row = (lane >> 4) + 4 r).  In that
// layout an accumulator tile is exactly the B operand of the next MFMA that contracts over its
// row index (probe: tests/tools/mfma_f64_probe.hip), so the whole whitening chain
//     Kt^T -> Kw^T = Linv Kt^T -> T^T = Vt^T Kw^T -> Y^T = Vt T^T -> adj -> gk^T = Linv^T adj
// runs register-to-register (the backward half lives in kernels_item_bwd.h); only the A operands (fragments of the M x M factors) come from LDS.
// kappa(t, z) itself is evaluated straight into B-operand layout (lane = time point, registers =
// inducing points), so Ktz never touches LDS or HBM.
//
// MP = 16 NT >= max_k M_k; matrices are zero padded to MP, LDS rows padded to LD = MP + 2 doubles
// (A-fragment reads are then bank-conflict free: lane -> (l & 15) * LD + (l >> 4)).
#pragma once
// Waves per block of k_times_fwd_mfma (SVGPFA_TIMES_NW up to SV_TIMES_MAXTH / 64; experiment builds raise the bound).
// Measured at cfg3 (us): 2 waves 93.3, 3 95.9, 4 92.2; with a 1024-thread build 5 waves 125, 7 118, 13 (one tile each)
// 149.  The kernel runs at ~85 % of its MFMA + VALU issue time: how the 13 tiles are dealt over the waves is not its limit.
#ifndef SV_TIMES_MAXTH
#define SV_TIMES_MAXTH 256
#endif
#include "engine_common.h"
#include "spike_cls_common.h"


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
          double av = (MODE == 0) ? lds_frag(&Mat[(16 * a + li) * LD + 16 * b + 4 * r + lg])
                                  : lds_frag(&Mat[(16 * b + 4 * r + lg) * LD + 16 * a + li]);
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, B[b][r], acc, 0, 0, 0);
        }
      }
    }
    D[a] = acc;
  }
}

__device__ __forceinline__ double grp4_sum(double x) {    // sum over the 4 lanes sharing lane & 15
  x += __shfl_xor(x, 16, 64);
  x += __shfl_xor(x, 32, 64);
  return x;
}

// Stage one LOWER-TRIANGULAR M x M matrix of the item (Linv or Vt; row-major, leading dimension M) into LDS, zero padded
// to MP x MP with rows padded to LD.  The upper triangle is not fetched (its cache lines -- a quarter of them at M = 32,
// three eighths at M = 64 -- hold zeros by construction).  U independent global loads per thread are in flight before the first LDS store (a plain element loop waits
// for every load before its store: the staging of a 32 x 32 matrix by 128 threads was eight exposed memory latencies).
// (U = 8 at M = 64 measured: k_item_bwd<4> 12.5 -> 14.0 ms at cfg5 -- the registers it takes cost more than the latency.)
template <int NT, int U = 4>
__device__ __forceinline__ void stage_matrix(double* dst, const double* src, int M, int tid, int nth) {
  constexpr int MP = 16 * NT, LD = MP + 2;
  for (int e0 = tid; e0 < MP * MP; e0 += U * nth) {
    double v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int e = e0 + u * nth, i = e / MP, j = e % MP;
      v[u] = (e < MP * MP && i < M && j <= i) ? src[sv_tri_off(i) + j] : 0.0;     // src: packed lower triangle
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int e = e0 + u * nth, i = e / MP, j = e % MP;
      if (e < MP * MP) dst[i * LD + j] = v[u];
    }
  }
}

// The same image by LDS-DMA (global_load_lds, 16 bytes per lane: every wave instruction fills 128 consecutive doubles of
// the padded image; the source address is per lane, so pairs above the diagonal and the row padding come from zeros in
// global memory, and the pair that holds a diagonal element of an odd-length row ends in the pad zero its producer wrote):
// no registers, every load of a matrix in flight at once.  For M > 32, where a block is alone on its CU and the
// register-staged form -- batches of four loads, each waited for -- left the staging of the two 32-KB factors as 10 of the
// block's 73 us at cfg5 (cycle stamps, -DSV_BWD_CLK; with 4-byte pieces the DMA took 11: 132 instructions per matrix).
// The caller waits (vmcnt) and synchronises before reading.
__device__ __attribute__((aligned(16))) const double sv_zero_pad[2] = {0.0, 0.0};
template <int NT>
__device__ __forceinline__ void stage_matrix_dma(double* dst, const double* src, int M, int tid, int nth) {
  constexpr int MP = 16 * NT, LD = MP + 2, TOT2 = MP * LD / 2;       // pairs of doubles in the image (LD is even)
  const int lane = tid & 63, wave = tid >> 6, nw = nth >> 6;
  for (int c = wave; c * 64 < TOT2; c += nw) {
    const int e2 = c * 64 + lane, i = (2 * e2) / LD, j = (2 * e2) % LD;
    if (e2 < TOT2) {
      const double* g = (i < M && j <= i) ? src + sv_tri_off(i) + j : sv_zero_pad;
      __builtin_amdgcn_global_load_lds((const void*)g, (__attribute__((address_space(3))) void*)(dst + c * 128), 16, 0, 0);
    }
  }
}

// ---- M x M matrices of an item in LDS: TILE-PACKED lower triangles (round 5).  A lower-triangular (Linv, Vt) or symmetric
// (Gt, At, C + C^T's inputs) matrix is kept as its NT (NT + 1) / 2 tiles of 16 x 16 on and below the diagonal, tile (a, b),
// a >= b, at sv_tidx(a, b) * SV_TSZ, rows of a tile SV_TS = 18 doubles apart (both MFMA fragment patterns of a tile --
// row-major li * 18 + 4 q + lg and transposed (4 q + lg) * 18 + li -- are conflict-free ds_read_b64).  A symmetric matrix
// is read through whichever of the two patterns the tile's position asks for.  Against full MP x (MP + 2) images: 23 KB
// instead of 34 KB per matrix at M = 64, which is what lets two blocks of k_item_bwd<4> share a CU.
#define SV_TS 18
#define SV_TSZ (16 * SV_TS)
__host__ __device__ constexpr int sv_tidx(int a, int b) { return a * (a + 1) / 2 + b; }      // a >= b
// element (i, j) of a tile-packed lower triangle, block row of i >= block row of j
__device__ __forceinline__ int sv_tp(int i, int j) { return sv_tidx(i >> 4, j >> 4) * SV_TSZ + (i & 15) * SV_TS + (j & 15); }

// blo / bhi (wave-uniform): B[b] is known to be zero outside blo <= b <= bhi (a panel of a triangular matrix): those tile
// products are skipped.
template <int NT, int MODE>
__device__ __forceinline__ void tri_chain_tp(d4 (&D)[NT], const double* Mat, const d4 (&B)[NT], const int orm, const int otr,
                                             const int blo = 0, const int bhi = NT - 1) {
#pragma unroll
  for (int a = 0; a < NT; ++a) {
    d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int b = 0; b < NT; ++b) {
      if ((MODE == 0 ? (b <= a) : (b >= a)) && b >= blo && b <= bhi) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const double av = (MODE == 0) ? lds_frag(&Mat[sv_tidx(a, b) * SV_TSZ + orm + 4 * q])
                                        : lds_frag(&Mat[sv_tidx(b, a) * SV_TSZ + otr + 4 * q * SV_TS]);
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, B[b][q], acc, 0, 0, 0);
        }
      }
    }
    D[a] = acc;
  }
}
// A-operand fragment (row 16 a + li, column 16 b + 4 q + lg) of a SYMMETRIC tile-packed matrix (a, b compile-time)
#define SV_SYMFRAG(Mat, a, b, q) \
  (((a) >= (b)) ? lds_frag(&(Mat)[sv_tidx((a), (b)) * SV_TSZ + orm + 4 * (q)]) : lds_frag(&(Mat)[sv_tidx((b), (a)) * SV_TSZ + otr + 4 * (q) * SV_TS]))

__device__ __forceinline__ void sv_tile_ab(int t, int& a, int& b) {
  a = (t >= 1) + (t >= 3) + (t >= 6);
  b = t - a * (a + 1) / 2;
}

// LDS-DMA staging of a packed lower triangle into the tile-packed image (global_load_lds, 16 bytes per lane: a wave
// instruction fills 128 consecutive doubles of the image; pairs above the diagonal, beyond M and the tiles' row padding
// come from a zero pair in global memory; the pair that holds a diagonal element of an odd-length row ends in the pad zero
// its producer wrote, sv_tri_off).  23 wave instructions per matrix at M = 64.
template <int NT>
__device__ __forceinline__ void stage_matrix_dma_tp(double* dst, const double* src, int M, int tid, int nth) {
  constexpr int TOT2 = NT * (NT + 1) / 2 * SV_TSZ / 2;       // pairs of doubles in the image
  const int lane = tid & 63, wave = tid >> 6, nw = nth >> 6;
  for (int cc = wave; cc * 64 < TOT2; cc += nw) {
    const int e2 = cc * 64 + lane;
    if (e2 < TOT2) {
      const int t = e2 / (SV_TSZ / 2), rem = e2 - t * (SV_TSZ / 2), it = rem / (SV_TS / 2), jp = rem - it * (SV_TS / 2);
      int a, b;
      sv_tile_ab(t, a, b);
      const int i = 16 * a + it, j = 16 * b + 2 * jp;
      const double* g = (jp < 8 && i < M && j <= i) ? src + sv_tri_off(i) + j : sv_zero_pad;
      __builtin_amdgcn_global_load_lds((const void*)g, (__attribute__((address_space(3))) void*)(dst + cc * 128), 16, 0, 0);
    }
  }
}

template <int NT>
__device__ __forceinline__ void load_factors(const EngineView& ev, const LatentMeta& m, int r, double* Li,
                                             double* Vl, double* ul, double* zl, bool want_v) {
  constexpr int MP = 16 * NT;
  const int M = m.M, tid = threadIdx.x;
  const long long base = m.offTT + (long long)r * m.PT;      // Linv, Vt: packed lower triangles
  const long long vbase = m.offM + (long long)r * M;
  if (NT <= 2 && want_v) {
    // both factors and the two vectors requested before anything is stored: ONE exposed memory latency instead of one
    // per batch of four loads (M <= 32: 8 + 8 + 2 values per thread at 128 threads; at M = 64 the registers this takes
    // cost more than the latency, see stage_matrix)
    constexpr int LD = MP + 2;
    const int nth = (int)blockDim.x;
    const double* sl = ev.Linv + base;
    const double* sv = ev.Vt + base;
    const double uv = (tid < M) ? ev.ut[vbase + tid] : 0.0;
    const double zv = (tid < M) ? ev.Zs[vbase + tid] : 0.0;
    for (int e0 = tid; e0 < MP * MP; e0 += 8 * nth) {
      double a[8], b[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int e = e0 + u * nth, i = e / MP, j = e % MP;
        const bool ok = e < MP * MP && i < M && j <= i;
        a[u] = ok ? sl[sv_tri_off(i) + j] : 0.0;
        b[u] = ok ? sv[sv_tri_off(i) + j] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int e = e0 + u * nth, i = e / MP, j = e % MP;
        if (e < MP * MP) {
          Li[i * LD + j] = a[u];
          Vl[i * LD + j] = b[u];
        }
      }
    }
    if (tid < MP) {
      ul[tid] = uv;
      zl[tid] = zv;
    }
    return;
  }
  if (NT >= 3) {
    const double uv = (tid < M) ? ev.ut[vbase + tid] : 0.0;
    const double zv = (tid < M) ? ev.Zs[vbase + tid] : 0.0;
    stage_matrix_dma<NT>(Li, ev.Linv + base, M, tid, (int)blockDim.x);
    if (want_v) stage_matrix_dma<NT>(Vl, ev.Vt + base, M, tid, (int)blockDim.x);
    if (tid < MP) {
      ul[tid] = uv;
      zl[tid] = zv;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the DMA pieces have landed (a barrier follows in every caller)
    return;
  }
  stage_matrix<NT>(Li, ev.Linv + base, M, tid, (int)blockDim.x);
  if (want_v) stage_matrix<NT>(Vl, ev.Vt + base, M, tid, (int)blockDim.x);
  if (tid < MP) {
    ul[tid] = (tid < M) ? ev.ut[vbase + tid] : 0.0;
    zl[tid] = (tid < M) ? ev.Zs[vbase + tid] : 0.0;
  }
}

// ------------------------------------------------------------------------ k_times_fwd_mfma
// Posterior mean / variance of latent k at a CSR list of times (reference
// svPosteriorOnLatents.py:185-216, :265-300).  grid = R*K blocks of 128 or 256 threads; each wavefront
// walks tiles of 16 time points.  Output element (trial r, time i of that trial, latent k) at
// base_r + i*os_t + k*os_k with base_r = toff[r]*os_t (CSR times) or r*os_r (nper times per trial).  The engine's own
// quadrature statistics use [r][k][q] (os_r = K Q, os_t = 1, os_k = Q): the 16 nodes of a tile are 128 contiguous bytes
// for this kernel's stores, for the adjoint loads of k_item_bwd and for the ELL kernel's rows.
//
// The tile loop of one wavefront.  Li, Vl (padded M x M images), ul = ut, zl = Z and (periodic) szl / czl = sin / cos of
// the inducing points' phases are in LDS and synchronised by the caller.  DYN = false: wave w takes tiles w, w + NWB, ...;
// DYN = true: the block's waves draw tiles from the LDS counter `next` (zeroed by the caller in front of its barrier) --
// for blocks whose waves do not start the loop together (k_front: one wave carries a rider).
// TP: Li / Vl are tile-packed lower triangles (sv_tidx) instead of MP x LD images.
template <int NT, int KT, bool DYN, bool TP = false>
__device__ __forceinline__ void times_tiles(const KPar& p, const double* Li, const double* Vl, const double* ul,
                                            const double* zl, const double* szl, const double* czl, const double* times,
                                            long long t0, int nt, double* mu_out, double* var_out, long long obase,
                                            long long os_t, int want_var, int* next, const int tid, double* kwc = nullptr,
                                            const int kw_mode = 0, const long long kw_half = 0) {
  // kwc / kw_mode (k_front only): the item's cached whitened tiles (EngineView::Kwc): 1 store them, 2 load them instead of
  // evaluating the kernel and whitening
  constexpr int MP = 16 * NT, LD = MP + 2;
  const int lane = tid & 63, wave = tid >> 6, li = lane & 15, lg = lane >> 4;
  // per-lane constants: ut and (RBF) Z of this lane's inducing points
  double uj[NT][4], zj[NT][4];
#pragma unroll
  for (int a = 0; a < NT; ++a)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      uj[a][q] = ul[16 * a + 4 * q + lg];
      zj[a][q] = zl[16 * a + 4 * q + lg];
    }
  const int ntiles = (nt + 15) >> 4;
  const int NWB = blockDim.x >> 6;     // waves per block (launch parameter)
  auto draw = [&](int prev) -> int {
    if (!DYN) return prev < 0 ? wave : prev + NWB;
    int v = 0;
    if (lane == 0) v = atomicAdd(next, 1);
    return __builtin_amdgcn_readfirstlane(v);
  };
  // the tile after this one is drawn, and its times fetched, one iteration ahead (global latency hidden behind the MFMA chain)
  int tile = draw(-1);
  double tq_next = (tile < ntiles) ? times[t0 + ((16 * tile + li < nt) ? 16 * tile + li : 0)] : 0.0;
  while (tile < ntiles) {
    const int node = 16 * tile + li;
    const bool ok = node < nt;
    const double tq = tq_next;
    const int tile_n = draw(tile);
    if (tile_n < ntiles) {
      const int nn = 16 * tile_n + li;
      tq_next = times[t0 + ((nn < nt) ? nn : 0)];
    }
    d4 Kw[NT];
    double* kwt = kwc + (long long)tile * NT * 256 + lane;          // this tile's images (kw_mode != 0 only)
    if (kw_mode == 2) {
#pragma unroll
      for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int q = 0; q < 4; ++q) Kw[a][q] = kwt[a * 256 + q * 64];
    } else {
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
          X0[a][q] = p.s2 * sv_exp2_neg(fma(cs, p.h2, -p.h2));     // (cs may exceed 1 by a rounding: 2^(+1e-16 h2) is harmless; k_item_bwd evaluates the same expression)
        }
      }
    if (TP) tri_chain_tp<NT, 0>(Kw, Li, X0, li * SV_TS + lg, lg * SV_TS + li);
    else tri_chain<NT, 0>(Kw, Li, LD, X0, li, lg);
    if (kw_mode == 1) {
      // both images: this kernel's own (linear), and k_item_bwd's transposed one -- register (a, q) of lane (li, lg) is its
      // register (a, li >> 2) of lane (4 q + lg, li & 3): a scattered store, once per kernel build
      double* kw2 = kwc + kw_half + (long long)tile * NT * 256 + (li >> 2) * 64 + (li & 3) * 16 + lg;
#pragma unroll
      for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          kwt[a * 256 + q * 64] = Kw[a][q];
          kw2[a * 256 + 4 * q] = Kw[a][q];
        }
    }
    }
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
      if (TP) tri_chain_tp<NT, 1>(T, Vl, Kw, li * SV_TS + lg, lg * SV_TS + li);
      else tri_chain<NT, 1>(T, Vl, LD, Kw, li, lg);
#pragma unroll
      for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int q = 0; q < 4; ++q) tt = fma(T[a][q], T[a][q], tt);
    }
    pm = grp4_sum(pm);
    n2 = grp4_sum(n2);
    tt = grp4_sum(tt);
    if (lg == 0 && ok) {
      const long long o = obase + (long long)node * os_t;
      mu_out[o] = pm;
      if (want_var) var_out[o] = (p.s2 - n2) + tt;
    }
    tile = tile_n;
  }
}

template <int NT, int KT>
__device__ __forceinline__ void times_fwd_body(const EngineView& ev, const int r, const int k, const double* times,
                               const long long* toff, int nper, double* mu_out, double* var_out, long long os_r,
                               long long os_t, long long os_k, int want_var, double* sm) {
  constexpr int MP = 16 * NT, LD = MP + 2;
  const int tid = threadIdx.x;
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
  const long long obase = (toff ? t0 * os_t : (long long)r * os_r) + (long long)k * os_k;
  times_tiles<NT, KT, false>(p, Li, Vl, ul, zl, szl, czl, times, t0, nt, mu_out, var_out, obase, os_t, want_var, nullptr, tid);
}

// One launch over all (trial, latent) items; the covariance type is resolved per block (wave-uniform branch).
template <int NT>
__global__ __launch_bounds__(SV_TIMES_MAXTH) void k_times_fwd_mfma(EngineView ev, const double* times,
                                                        const long long* toff, int nper, double* mu_out,
                                                        double* var_out, long long os_r, long long os_t,
                                                        long long os_k, int want_var) {
  extern __shared__ double sm[];
  const int r = blockIdx.x / ev.K, k = blockIdx.x % ev.K;
  if (ev.meta[k].ktype == SV_RBF)
    times_fwd_body<NT, SV_RBF>(ev, r, k, times, toff, nper, mu_out, var_out, os_r, os_t, os_k, want_var, sm);
  else
    times_fwd_body<NT, SV_PERIODIC>(ev, r, k, times, toff, nper, mu_out, var_out, os_r, os_t, os_k, want_var, sm);
}

// ----------------------------------------------------------------------------- k_prep_mfma
// ut = Linv m, a = Linv^T ut, Vt = Linv L_S and the KL term per (trial, latent), on the f64 MFMA.
// NT wavefronts per item: wave c produces the 16-column panel c of Vt; wave 0 also carries the vector m as an
// extra panel (column 0) so that ut and a come out of the same tri_chain.  Linv and L_S are staged in LDS with
// coalesced loads by the whole block.
// want_w: wave 0 also projects a on the item's cosine series (spike_pre_item: the class path's W, needed when dC is
// wanted) -- it has a at hand, and the class path is spared a launch behind a cross-stream wait.
template <int NT>
__global__ __launch_bounds__(64 * NT) void k_prep_mfma(EngineView ev, LatentPtrTable ptab, int want_w) {
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
  double* wsc = red + NT;        // 64 + SV_PRE_LDS (want_w)
  const long long base = m.offTT + (long long)r * m.PT;      // Linv, Vt: packed lower triangles
  const long long vbase = m.offM + (long long)r * M;
  const double* cv = pt.cholvec + (long long)r * m.P;
  if (NT >= 3) stage_matrix_dma<NT>(Li, ev.Linv + base, M, tid, NTH);     // (lands behind the loads below; the barrier waits)
  else stage_matrix<NT>(Li, ev.Linv + base, M, tid, NTH);
  for (int e0 = tid; e0 < MP * MP; e0 += 4 * NTH) {      // L_S from the row-major tril vector, four loads in flight
    double v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e = e0 + u * NTH, i = e / MP, j = e % MP;
      v[u] = (e < MP * MP && i < M && j <= i) ? cv[i * (i + 1) / 2 + j] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e = e0 + u * NTH, i = e / MP, j = e % MP;
      if (e < MP * MP) Ls[i * LD + j] = v[u];
    }
  }
  if (NT >= 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the DMA pieces of Linv have landed
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
        if (i < M && j <= i) ev.Vt[base + sv_tri_off(i) + j] = V[a][q];       // (Vt = Linv L_S is lower triangular)
        if (i < M && j == i && !(i & 1)) ev.Vt[base + sv_tri_off(i) + i + 1] = 0.0;   // the pad of an odd-length row
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
          if (want_w) wsc[i] = A[a][q];
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
  if (want_w && c == 0) spike_pre_item(ev, item, r, k, m, (lane < M) ? wsc[lane] : 0.0, wsc + 64);
}

// ------------------------------------------------------------------------------ k_front
// k_prep_mfma and the quadrature pass of k_times_fwd_mfma in ONE block per (trial, latent) item (svgpfa_elbo; round 4): the
// block stages Linv and L_S once, its first NT waves produce the panels of Vt = Linv L_S (-> HBM for k_item_bwd, and into
// L_S's place in LDS), one more wave the mean chain (ut, a) and the KL sum; then all four waves walk the trial's
// quadrature tiles (times_tiles, tiles drawn from an LDS counter) with both factors already in LDS.  Against the two
// launches: one staging of Linv and none of Vt (44 MB of reads at the north-star size), one launch boundary less -- on a
// small shard one whole round of the chip -- and the class path's W (spike_pre_item, when dC is wanted) rides on the
// block's last wave while the others have already started on the tiles.
// One item per block: a version with persistent blocks that requested the NEXT item's inputs into registers ahead of the
// tile loop was built and measured (round 4): 146-171 us against 128 at the north-star size.  The 8-15 us that cycle
// stamps show between a block's start and its first tile are not exposed memory latency but this block's own instructions
// (index arithmetic of the staging, the panel and mean chains, the logs of the KL term) queueing behind three other waves
// per SIMD that are in their tile loops: the kernel is issue-bound, and the item loop only added registers (the compiler
// hoists every fp64 polynomial constant of the body out of such a loop: 94 -> 177 VGPRs).
// grid = R K blocks of 256 threads; dynamic LDS sv_front_lds(NT).
// M > 32: the two factors as tile-packed lower triangles (46 KB instead of 68 at M = 64: three blocks per CU, what the kernel's
// registers allow, instead of two -- round 5)
__host__ __device__ constexpr bool sv_front_tp(int NT) { return NT >= 3; }
__host__ __device__ constexpr size_t sv_front_lds(int NT) {
  const size_t MP = 16 * (size_t)NT, LD = MP + 2;
  const size_t mat = sv_front_tp(NT) ? (size_t)(NT * (NT + 1) / 2) * SV_TSZ : MP * LD;
  return (2 * mat + 5 * MP + 6 + 64 + SV_PRE_LDS) * 8;
}
#ifndef SV_FRONT_CLK
#define SV_FRONT_CLK 0    // diagnosis builds only: 1 = two threads of two blocks print cycle stamps of the phases (device printf)
#endif
#define SV_FCK(i) do { if (SV_FRONT_CLK) fck[i] = clock64(); } while (0)

template <int NT, int KT>
__device__ __forceinline__ void front_body(const EngineView& ev, const LatentPtrTable& ptab, const LatentMeta& m, int r, int k,
                                           int want_w, double* sm) {
  constexpr int MP = 16 * NT, LD = MP + 2, NTH = 256;
  constexpr bool TP = sv_front_tp(NT);
  constexpr int MAT = TP ? NT * (NT + 1) / 2 * SV_TSZ : MP * LD, TS = SV_TS, TSZ = SV_TSZ;
  constexpr int PW = (NT < 4) ? NT : 0;            // the wave that carries the mean chain
  const int tid = threadIdx.x;
  const int item = r * ev.K + k, lane = tid & 63, wave = tid >> 6, li = lane & 15, lg = lane >> 4;
  const int orm = li * TS + lg, otr = lg * TS + li;
  const int M = m.M;
  const LatentPtrs& pt = ptab.p[k];
  double* Li = sm;               // MAT
  double* Ls = Li + MAT;         // MAT: L_S, then Vt
  double* ul = Ls + MAT;
  double* zl = ul + MP;
  double* szl = zl + MP;
  double* czl = szl + MP;
  double* ml = czl + MP;         // the mean
  double* red = ml + MP;         // 4
  int* next = (int*)(red + 4);   // tile counter (one double slot; red + 5 unused)
  double* wsc = red + 6;         // 64 + SV_PRE_LDS (want_w)
  long long fck[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  SV_FCK(0);
  const KPar p = make_kpar(m, ev.theta + 2 * k);
  const long long base = m.offTT + (long long)r * m.PT;      // Linv, Vt: packed lower triangles
  const long long vbase = m.offM + (long long)r * M;
  const double* cv = pt.cholvec + (long long)r * m.P;
  // every global load of the block's inputs is requested before the first LDS store: one exposed memory latency
  const double zv = (tid < M) ? ev.Zs[vbase + tid] : 0.0;
  const double mv = (tid < M) ? pt.mean[(long long)r * M + tid] : 0.0;
  if (NT <= 2) {
    constexpr int NV = (MP * MP + NTH - 1) / NTH;
    const double* sl = ev.Linv + base;
    double vl[NV], vs[NV];
#pragma unroll
    for (int u = 0; u < NV; ++u) {
      const int e = tid + u * NTH, i = e / MP, j = e % MP;
      const bool ok = e < MP * MP && i < M && j <= i;
      vl[u] = ok ? sl[sv_tri_off(i) + j] : 0.0;
      vs[u] = ok ? cv[i * (i + 1) / 2 + j] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < NV; ++u) {
      const int e = tid + u * NTH, i = e / MP, j = e % MP;
      if (e < MP * MP) {
        Li[i * LD + j] = vl[u];
        Ls[i * LD + j] = vs[u];
      }
    }
  } else {
    stage_matrix_dma_tp<NT>(Li, ev.Linv + base, M, tid, NTH);     // (lands behind the loads below; the barrier waits)
    constexpr int NE = NT * (NT + 1) / 2 * 256;
    for (int e0 = tid; e0 < NE; e0 += 4 * NTH) {           // L_S from the row-major tril vector, four loads in flight
      double v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int e = e0 + u * NTH, t = e >> 8, it = (e >> 4) & 15, jt = e & 15;
        int a, b;
        sv_tile_ab(t, a, b);
        const int i = 16 * a + it, j = 16 * b + jt;
        v[u] = (e < NE && i < M && j <= i) ? cv[i * (i + 1) / 2 + j] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int e = e0 + u * NTH, t = e >> 8, it = (e >> 4) & 15, jt = e & 15;
        if (e < NE) Ls[t * TSZ + it * TS + jt] = v[u];
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the DMA pieces of Linv have landed
  }
  if (tid < MP) {
    zl[tid] = zv;
    ml[tid] = mv;
    if (KT == SV_PERIODIC) sv_sincos_quarter(zv * p.q4, szl[tid], czl[tid]);
  }
  if (tid == 0) *next = 0;
  SV_FCK(1);
  __syncthreads();
  SV_FCK(2);

  // ---- waves < NT: panel `wave` of Vt = Linv L_S; wave PW: ut = Linv m, a = Linv^T ut (the mean as column 0 of a panel)
  double part = 0.0;
  d4 V[NT];
  {
    const int c = wave % NT;     // (waves >= NT recompute a panel they never store: no wave-uniform branch around the chain)
    d4 P[NT];
#pragma unroll
    for (int b = 0; b < NT; ++b)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (TP) {                // L_S[:, panel c]: tiles (b, c), b >= c
          const double v = Ls[((b >= c) ? (b * (b + 1) / 2 + c) : 0) * TSZ + otr + 4 * q * TS];
          P[b][q] = (b >= c) ? v : 0.0;
        } else {
          P[b][q] = Ls[(16 * b + 4 * q + lg) * LD + 16 * c + li];
        }
      }
    if (TP) tri_chain_tp<NT, 0>(V, Li, P, orm, otr);
    else tri_chain<NT, 0>(V, Li, LD, P, li, lg);
    if (wave < NT) {
#pragma unroll
      for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int q = 0; q < 4; ++q) part = fma(V[a][q], V[a][q], part);
    }
  }
  d4 U[NT], A[NT];
  if (wave == PW) {
    d4 Pm[NT];
#pragma unroll
    for (int b = 0; b < NT; ++b)
#pragma unroll
      for (int q = 0; q < 4; ++q) Pm[b][q] = (li == 0) ? ml[16 * b + 4 * q + lg] : 0.0;
    if (TP) {
      tri_chain_tp<NT, 0>(U, Li, Pm, orm, otr);
      tri_chain_tp<NT, 1>(A, Li, U, orm, otr);
    } else {
      tri_chain<NT, 0>(U, Li, LD, Pm, li, lg);
      tri_chain<NT, 1>(A, Li, LD, U, li, lg);
    }
    const int dgo = TP ? sv_tp(lane < MP ? lane : 0, lane < MP ? lane : 0) : lane * LD + lane;
    if (lane < M) part += -2.0 * log(Li[dgo]) - 2.0 * log(fabs(Ls[dgo]));
  }
  SV_FCK(3);
  __syncthreads();               // every wave is done reading L_S: its buffer becomes Vt
  if (wave < NT) {
    const int c = wave;
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int i = 16 * a + 4 * q + lg, j = 16 * c + li;
        if (TP) {
          if (a >= c) Ls[sv_tidx(a, c) * TSZ + otr + 4 * q * TS] = V[a][q];      // (tiles on and below the diagonal; the rest of Vt is zero)
        } else {
          Ls[i * LD + j] = V[a][q];                                             // (exact zeros above the diagonal and in the padding)
        }
        if (i < M && j <= i) ev.Vt[base + sv_tri_off(i) + j] = V[a][q];
        if (i < M && j == i && !(i & 1)) ev.Vt[base + sv_tri_off(i) + i + 1] = 0.0;   // the pad of an odd-length row
      }
  }
  if (wave == PW) {
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int i = 16 * a + 4 * q + lg;
        if (li == 0) {
          ul[i] = U[a][q];                                                       // (zero beyond M: padded rows of Linv)
          if (i < M) {
            ev.ut[vbase + i] = U[a][q];
            ev.a[vbase + i] = A[a][q];
            part = fma(U[a][q], U[a][q], part);
            wsc[i] = A[a][q];
          }
        }
      }
  }
  part = wave_sum(part);
  if (lane == 0) red[wave] = part;
  SV_FCK(4);
  __syncthreads();
  SV_FCK(5);
  if (tid == 0) ev.klpart[item] = 0.5 * (((red[0] + red[1]) + (red[2] + red[3])) - (double)M);     // fixed order
  // ---- the class path's W on the last wave, ahead of its share of the tiles
  if (want_w && wave == 3) spike_pre_item(ev, item, r, k, m, (lane < M) ? wsc[lane] : 0.0, wsc + 64);
  // ---- quadrature statistics [r][k][q]
  times_tiles<NT, KT, true, TP>(p, Li, Ls, ul, zl, szl, czl, ev.tq, (long long)r * ev.Q, ev.Q, ev.muq, ev.varq,
                                ((long long)r * ev.K + k) * ev.Q, 1LL, 1, next, tid,
                                ev.kw_mode ? ev.Kwc + (long long)item * ((ev.Q + 15) >> 4) * NT * 256 : nullptr, ev.kw_mode,
                                ev.kw_half);
  if (SV_FRONT_CLK) {
    SV_FCK(6);
    if ((blockIdx.x == 101 || blockIdx.x == 202) && (tid == 0 || tid == 192)) {
      printf("FCK kt %d tid %d: %lld %lld %lld %lld %lld %lld\n", KT, tid, fck[1] - fck[0], fck[2] - fck[0], fck[3] - fck[0],
             fck[4] - fck[0], fck[5] - fck[0], fck[6] - fck[0]);
    }
  }
}

template <int NT>
__global__ __launch_bounds__(256) void k_front(EngineView ev, LatentPtrTable ptab, int want_w) {
  extern __shared__ double sm[];
  const int r = blockIdx.x / ev.K, k = blockIdx.x % ev.K;
  const LatentMeta m = ev.meta[k];
  if (m.ktype == SV_RBF) front_body<NT, SV_RBF>(ev, ptab, m, r, k, want_w, sm);
  else front_body<NT, SV_PERIODIC>(ev, ptab, m, r, k, want_w, sm);
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
// ---- scalar type of the ELL kernel: double, or float in precision mode "mixed" (v_mfma_f32_16x16x4_f32 + v_exp_f32).
// Operand layouts of the two 16x16x4 MFMAs are the same (A: row = lane & 15, k = lane >> 4; B: k = lane >> 4, col = lane & 15);
// the ACCUMULATOR layouts differ: f64 register q holds row (lane >> 4) + 4 q, f32 register q holds row 4 (lane >> 4) + q.
// An accumulator tile is used as the B operand of step q of the next product, so the k index of step q is that row.
#define SV_ELL_OFF (-1.0e5)      // exponent offset of a padded neuron: 2^(h + SV_ELL_OFF) == 0 exactly (fp64: clamped to 2^-1100; fp32: v_exp_f32)
typedef float f4 __attribute__((ext_vector_type(4)));
template <typename T> struct EllT;
template <> struct EllT<double> {
  typedef d4 acc;
  static __device__ __forceinline__ acc mfma(double a, double b, acc c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
  static __device__ __forceinline__ int row(int lg, int q) { return lg + 4 * q; }
  // the unclamped kernel: for every FINITE argument v_cvt_i32_f64 saturates and v_ldexp_f64 overflows to inf / underflows to 0,
  // exactly what the clamp to [-1100, 1100] gave (two more instructions per exponential, a tenth of this loop's VALU work)
  static __device__ __forceinline__ double exp2(double x) { return sv_exp2_neg(x); }
};
template <> struct EllT<float> {
  typedef f4 acc;
  static __device__ __forceinline__ acc mfma(float a, float b, acc c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
  static __device__ __forceinline__ int row(int lg, int q) { return 4 * lg + q; }
  static __device__ __forceinline__ float exp2(float x) { return __builtin_amdgcn_exp2f(x); }   // saturates to 0 / inf
};

template <int KS, int ROLE, typename T>
__device__ __forceinline__ void quad_ell_mfma_body(const EngineView& ev, const double* C, const double* d,
                                                   int want_sums, int write_ell, int parts, int lc, int blk,
                                                   double* sm_d) {
  typedef EllT<T> X;
  typedef typename X::acc accT;
  constexpr int KKP = 4 * KS, LDK = KKP + 2, KKT = (KKP + 15) / 16;
  // Output rows of the sum products (the 2K columns of [mu | var] or [C | C^2/2]): fp64 takes whole 16-row tiles with
  // v_mfma_f64_16x16x4_f64 and the rows left over, four at a time, with v_mfma_f64_4x4x4_4b_f64 -- four independent 4x4x4
  // products whose operand and result lanes (lane = x + 4 block + 16 k; result lane = j + 4 block + 16 i, probed on the chip:
  // tests/tools/probes/mfma_f64_4x4x4_probe.hip) are exactly the 16x16x4 layouts restricted to four output rows, at a quarter
  // of the issue time (7.3 against 29 ns per SIMD).  At 2K = 20 the second 16-row tile carried 4 useful rows: 8 -> 4 + 4
  // quarter-cost instructions per tile pair (round 5).  fp32 keeps whole tiles (its 4x4 instruction has k = 1).
  constexpr bool SPL = sizeof(T) == 8;
  constexpr int FT = SPL ? KS / 4 : KKT;          // whole 16-row tiles
  constexpr int LG = SPL ? KS % 4 : 0;            // groups of four left-over rows
  constexpr int FTA = FT > 0 ? FT : 1, LGA = LG > 0 ? LG : 1;
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
  T* LM = (T*)sm_d;                         // (Lrows + 1) * LDK  looped matrix: ROLE 0 [mu | var], ROLE 1 log2e [C | C^2/2]
  T* lv = LM + (Lrows + 1) * LDK;           // Lp   looped vector: ROLE 0 w_q, ROLE 1 log2e d_n
  T* OW = lv + Lp;                          // NW * 16 * LDK  per-wave rows of the other matrix
  double* red = (double*)(OW + NW * 16 * LDK + ((NW * 16 * LDK + Lp + (Lrows + 1) * LDK) & 1));   // 8 (fp64, 8-B aligned)
  const int lrows_tot = (ROLE == 0) ? Q : N;

  // fill rows [row0, row0 + Lrows) of the looped matrix: zero the padding, then batched independent loads
  // (8 in flight per thread).  ROLE 0: the trial's statistics mu / var, stored [k][q] (K runs of nrow contiguous nodes);
  // ROLE 1: C [N x K], contiguous
  auto fill_lm = [&](int row0) {
    for (int e = tid; e < (Lrows + 1) * LDK; e += NTH) LM[e] = (T)0;
    __syncthreads();
    const int nrow = (lrows_tot - row0 < Lrows) ? (lrows_tot - row0) : Lrows;
    const int tot = nrow * K;
    const double* s0 = (ROLE == 0) ? ev.muq + (long long)r * Q * K + row0 : C + (long long)row0 * K;
    const double* s1 = (ROLE == 0) ? ev.varq + (long long)r * Q * K + row0 : s0;
    for (int b0 = 0; b0 < tot; b0 += NTH * 8) {
      double v0[8], v1[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int idx = b0 + u * NTH + tid;
        const bool ok = idx < tot;
        long long src = idx;
        if (ROLE == 0) {
          const int kk = idx / nrow;
          src = (long long)kk * Q + (idx - kk * nrow);
        }
        v0[u] = ok ? s0[src] : 0.0;
        v1[u] = (ROLE == 0) ? (ok ? s1[src] : 0.0) : v0[u];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int idx = b0 + u * NTH + tid;
        if (idx < tot) {
          int row, c;
          if (ROLE == 0) {
            c = idx / nrow;
            row = idx - c * nrow;
          } else {
            row = idx / K;
            c = idx - row * K;
          }
          if (ROLE == 0) {
            LM[row * LDK + c] = (T)v0[u];
            LM[row * LDK + K + c] = (T)v1[u];
          } else {
            LM[row * LDK + c] = (T)(v0[u] * L2E);
            LM[row * LDK + K + c] = (T)(0.5 * v0[u] * v0[u] * L2E);
          }
        }
      }
    }
    __syncthreads();
  };

  // Padding of the looped index is folded into the data instead of masked per element (round 4: two selects and a compare
  // per element were an eighth of this issue-bound loop's VALU instructions): a padded node has weight 0 (ROLE 0), a padded
  // neuron the offset SV_ELL_OFF = -1e5 in the exponent (ROLE 1), and 2^(h - 1e5) is an exact 0 in either arithmetic.
  for (int i = tid; i < Lp; i += NTH)
    lv[i] = (T)((ROLE == 0) ? ((i < Q) ? ev.wq[(long long)r * Q + i] : 0.0) : ((i < N) ? d[i] * L2E : SV_ELL_OFF));
  if (LC == LT) fill_lm(0);     // common case: the whole looped matrix stays resident
  else __syncthreads();

  double esum = 0.0;
  T* ow = OW + wave * 16 * LDK;
  for (int ot0 = part * NW; ot0 < OT; ot0 += parts * NW) {     // block-uniform rounds; wave `wave` owns tile ot0 + wave
    const int ot = ot0 + wave;
    const bool active = ot < OT;
    // own 16 rows -> per-wave LDS slice (wave-local: no block barrier needed).  ROLE 0: 16 rows of C, contiguous
    // (16 K doubles); ROLE 1: the tile's 16 nodes of mu / var, K runs of nrow contiguous values ([k][q] layout)
    for (int e = lane; e < 16 * LDK; e += 64) ow[e] = (T)0;
    lds_wave_fence();
    if (active) {
      const int orows = (ROLE == 0) ? N : Q;
      const int nrow = (orows - ot * 16 < 16) ? (orows - ot * 16) : 16;
      const int tot = nrow * K;
      const double* s0 = (ROLE == 0) ? C + (long long)ot * 16 * K : ev.muq + (long long)r * Q * K + ot * 16;
      const double* s1 = (ROLE == 0) ? s0 : ev.varq + (long long)r * Q * K + ot * 16;
      double v0[8], v1[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int idx = u * 64 + lane;
        const bool ok = idx < tot;
        long long src = idx;
        if (ROLE == 1) {
          const int kk = idx / nrow;
          src = (long long)kk * Q + (idx - kk * nrow);
        }
        v0[u] = ok ? s0[src] : 0.0;
        v1[u] = (ROLE == 0) ? v0[u] : (ok ? s1[src] : 0.0);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int idx = u * 64 + lane;
        if (idx < tot) {
          int row, c;
          if (ROLE == 0) {
            row = idx / K;
            c = idx - row * K;
          } else {
            c = idx / nrow;
            row = idx - c * nrow;
          }
          if (ROLE == 0) {
            ow[row * LDK + c] = (T)(v0[u] * L2E);
            ow[row * LDK + K + c] = (T)(0.5 * v0[u] * v0[u] * L2E);
          } else {
            ow[row * LDK + c] = (T)v0[u];
            ow[row * LDK + K + c] = (T)v1[u];
          }
        }
      }
    }
    lds_wave_fence();
    const int oi = ot * 16 + li;              // own index of this lane's column
    const bool ook = active && oi < ((ROLE == 0) ? N : Q);
    // (the own index's padding likewise: ROLE 0 the exponent offset, ROLE 1 the weight 0)
    const T ov = (T)(!ook ? ((ROLE == 0) ? SV_ELL_OFF : 0.0) : ((ROLE == 0) ? d[oi] * L2E : ev.wq[(long long)r * Q + oi]));
    T bfrag[KS];                              // own rows as the B operand of the h product
#pragma unroll
    for (int c = 0; c < KS; ++c) bfrag[c] = ow[li * LDK + 4 * c + lg];
    accT acc[FTA];
    double accl[LGA];
#pragma unroll
    for (int t = 0; t < FTA; ++t) acc[t] = (accT){0, 0, 0, 0};
#pragma unroll
    for (int g = 0; g < LGA; ++g) accl[g] = 0.0;
    T colsum = 0;
    for (int lt0 = 0; lt0 < LT; lt0 += LC) {
      if (LC < LT) fill_lm(lt0 * 16);         // block-uniform: every wave takes part, active or not
      const int lt1 = (lt0 + LC < LT) ? lt0 + LC : LT;
      if (active)
      {
        // running pointers into the looped matrix and vector (one add each per tile: with indices recomputed from the tile
        // number the loop carried eight 32-bit adds per tile pair, a ninth of its vector instructions)
        const T* pA = LM + li * LDK + lg;                       // row lr + li, column 4 c + lg
        const T* pT = LM + X::row(lg, 0) * LDK + li;            // row lr + row(lg, c), column 16 t + li
        const T* pv = lv + lt0 * 16 + X::row(lg, 0);            // element lt 16 + row(lg, q)
        const T* pL = LM + X::row(lg, 0) * LDK + 16 * FT + (li & 3);   // row lr + lg + 4 c, column 16 FT + 4 g + (li & 3)
        constexpr int RS = (sizeof(T) == 8) ? 4 : 1;            // row(lg, c) = row(lg, 0) + RS c  (f64: lg + 4 c; f32: 4 lg + c)
        for (int lt = lt0; lt < lt1; ++lt, pA += 16 * LDK, pT += 16 * LDK, pL += 16 * LDK, pv += 16) {
          accT h = {0, 0, 0, 0};
#pragma unroll
          for (int c = 0; c < KS; ++c) h = X::mfma(lds_frag(pA + 4 * c), bfrag[c], h);
          T afrag[FTA][4];                    // looped rows, transposed, for the sum product (independent of E)
          T afl[LGA][4];                      // (left-over rows: i = lane & 3 of each of the four blocks)
          if (want_sums) {
#pragma unroll
            for (int t = 0; t < FT; ++t)
#pragma unroll
              for (int c = 0; c < 4; ++c) afrag[t][c] = pT[RS * c * LDK + 16 * t];
#pragma unroll
            for (int g = 0; g < LG; ++g)
#pragma unroll
              for (int c = 0; c < 4; ++c) afl[g][c] = pL[RS * c * LDK + 4 * g];
          }
          accT E;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            // ROLE 0: E = w_q 2^(h + d_n) ; ROLE 1: E^T = w_q 2^(h^T + d_n); exact 0 where the looped or the own index is padding
            E[q] = (ROLE == 0) ? pv[RS * q] * X::exp2(h[q] + ov) : ov * X::exp2(h[q] + pv[RS * q]);
            colsum += E[q];
          }
          if (want_sums) {
#pragma unroll
            for (int t = 0; t < FT; ++t)
#pragma unroll
              for (int c = 0; c < 4; ++c) acc[t] = X::mfma(afrag[t][c], E[c], acc[t]);
            if constexpr (LG > 0) {
#pragma unroll
              for (int g = 0; g < LG; ++g)
#pragma unroll
                for (int c = 0; c < 4; ++c) accl[g] = __builtin_amdgcn_mfma_f64_4x4x4f64((double)afl[g][c], (double)E[c], accl[g], 0, 0, 0);
            }
          }
        }
      }
      if (LC < LT) __syncthreads();           // all waves done with this chunk before it is overwritten
    }
    esum += (double)colsum;
    if (want_sums && ook) {
      if (ROLE == 0) {
#pragma unroll
        for (int t = 0; t < FT; ++t)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int kk = 16 * t + X::row(lg, q);
            if (kk < K2) ev.gCpart[((long long)r * N + oi) * K2 + kk] = (double)acc[t][q];
          }
#pragma unroll
        for (int g = 0; g < LG; ++g) {                          // (result lane = column + 16 row: row lg of group g)
          const int kk = 16 * FT + 4 * g + lg;
          if (kk < K2) ev.gCpart[((long long)r * N + oi) * K2 + kk] = accl[g];
        }
      } else {
        const long long gi = (long long)r * Q * K + oi;        // [r][k][q]: the 16 nodes of the tile are contiguous
        auto put = [&](const int kk, const double v) {
          if (kk < K) ev.gmuq[gi + (long long)kk * Q] = -v * LN2;
          else if (kk < K2) ev.gvarq[gi + (long long)(kk - K) * Q] = -v * LN2;
        };
#pragma unroll
        for (int t = 0; t < FT; ++t)
#pragma unroll
          for (int q = 0; q < 4; ++q) put(16 * t + X::row(lg, q), (double)acc[t][q]);
#pragma unroll
        for (int g = 0; g < LG; ++g) put(16 * FT + 4 * g + lg, accl[g]);
      }
    }
    if (ROLE == 0 && want_sums) {
      const double cs = grp4_sum((double)colsum);
      if (lg == 0 && ook) ev.gdpart[(long long)r * N + oi] = -cs;
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
// ncomb > 0 (round 4): behind the ELL blocks ceil(ncomb / 4) RIDER blocks combine the class path's spike moments of the
// ncomb = R K (trial, latent) items, one item per wave (spike_comb_item: the work of k_spike_comb, which needs the class
// kernels in front of this launch and k_item_bwd behind it -- exactly this kernel's place on the stream -- and was a
// launch of 9.5 us for 5 000 single-wave blocks of a few microseconds each).
template <int KS, int ROLE, typename T = double>
__global__ __launch_bounds__(512) void k_quad_ell_mfma(EngineView ev, const double* C, const double* d,
                                                       int want_sums, int write_ell, int parts, int lc0, int lc1,
                                                       int ncomb, int comb_full) {
  extern __shared__ double sm[];
  // ROLE 2 without dC / dd (the E-step closure): no role-0 blocks at all -- role 1 evaluates every E[q, n] as well, so it
  // writes the ELL value itself (round 5: until then role 0 ran its h products and exponentials for the sake of that one sum,
  // a third of the launch)
  const int nb0 = (ROLE == 2 && !want_sums) ? 0 : ev.R * parts;
  const int nell = nb0 + (ROLE == 2 ? ev.R * parts : 0);
  if ((int)blockIdx.x >= nell) {                       // rider blocks (only launched with ncomb > 0)
    const int nwb = (int)blockDim.x >> 6, wave = (int)threadIdx.x >> 6;
    const int unit = ((int)blockIdx.x - nell) * nwb + wave;
    sv_comb_rider(ev, unit, ncomb, comb_full, sm, wave);      // (bit 0: full, bit 1: time-weighted moments, bit 2: two items per wave)
    return;
  }
  if (ROLE == 2) {
    if ((int)blockIdx.x < nb0)
      quad_ell_mfma_body<KS, 0, T>(ev, C, d, want_sums, 1, parts, lc0, blockIdx.x, sm);
    else
      quad_ell_mfma_body<KS, 1, T>(ev, C, d, 1, want_sums ? 0 : 1, parts, lc1, blockIdx.x - nb0, sm);
  } else {
    quad_ell_mfma_body<KS, (ROLE == 2 ? 0 : ROLE), T>(ev, C, d, want_sums, write_ell, parts, ROLE == 0 ? lc0 : lc1,
                                                       blockIdx.x, sm);
  }
}

// ------------------------------------------------------------------------- k_quad_ell_fused  (round 5)
// Both sums of the quadrature term from ONE evaluation of E (fp64; dC / dd AND the latent adjoints wanted; N <= 32 NG
// neurons, 2K <= 20).  The two roles of k_quad_ell_mfma each run the h product (KS MFMAs) and the four exponentials of a lane
// for every (node tile, neuron tile) pair -- 5 of 13 + 5 matrix instructions and 68 of ~70 vector instructions per pair and
// role at the north-star shape.  Here a block is (trial, GROUP of <= NG neuron tiles); a pair is evaluated as role 1 does
// (E^T in accumulator layout: rows = neurons, columns = nodes):
//   latent adjoints   [gmu | gvar]^T += [C | C^2/2]^T E^T           E^T is the B operand as it stands (contraction over rows)
//   dC / dd           [A1 | A2]^T    += [mu | var]^T E              E = the tile TRANSPOSED through 2 KB of the wave's LDS (four
//                                                                   ds_write_b64 + four ds_read_b64 per lane instead of the
//                                                                   second h product and its exponentials); the accumulators
//                                                                   of all NG neuron tiles stay in registers across the
//                                                                   wave's node tiles (NG (KS + 1) doubles per lane)
// Both operand matrices of the block sit in LDS (the group's [C | C^2/2] and ALL nodes' [mu | var] of the trial), so a wave
// can take any pair: the block's QT ng pairs, in node-major order, are dealt over the waves in equal contiguous RANGES (13 x 7
// pairs over four waves: 23 / 23 / 23 / 22 -- whole node tiles per wave would be 28 / 21 / 21 / 21).  A node tile cut by a
// range boundary has its latent adjoints in two waves: the second part (the start of the later wave's range) goes to an LDS
// slot, the first part's owner (it ends its range there, accumulators still in registers) adds the slot behind the loop's
// barrier.  The waves' partial [A1 | A2] / column sums are added in wave order through LDS behind the loop (the looped
// matrix's region).  What leaves the block complete: dC / dd rows of the group's neurons, the ELL partial; what leaves it
// PARTIAL over the groups: the latent adjoints of the trial's nodes -- group g writes plane g of gmuq / gvarq (R K Q doubles
// each: 8 MB at the north-star shape) and k_item_bwd's node staging adds the planes (EngineView::glat2).
// Host: QT >= waves per block (a range then spans at least one whole tile's worth of pairs: no tile is cut twice).
#ifndef SV_ELLF_DIAG
#define SV_ELLF_DIAG 0      // diagnosis builds (wrong results): 1 no pair loop, 2 no reduction / output behind the loop
#endif
#ifndef SV_ELLF_CLK
#define SV_ELLF_CLK 0       // diagnosis builds: 1 = threads 0 / 192 of three blocks print cycle stamps of the phases (device printf)
#endif
#define SV_ECK(i) do { if (SV_ELLF_CLK) eck[i] = wall_clock64(); } while (0)
#define SV_ELLF_TRS 18      // row stride of the transpose tile: like LDK = 4 KS + 2, the transposed reads (18 li + lg) of a 32-lane group hit 64 different banks
template <int KS, int NG>
__device__ __forceinline__ void quad_ell_fused_body(const EngineView& ev, const double* C, const double* d, const int ngroups,
                                                    const int blk, double* sm) {
  constexpr int KKP = 4 * KS, LDK = KKP + 2, TRS = SV_ELLF_TRS;
  constexpr int FT = KS / 4, LG = KS % 4, FTA = FT > 0 ? FT : 1, LGA = LG > 0 ? LG : 1;
  constexpr int NACC = 4 * FT + LG;                  // accumulator doubles per lane of one [16 x 2K] sum
  const double L2E = 1.4426950408889634074, LN2 = 0.69314718055994530942;
  const int K = ev.K, N = ev.N, Q = ev.Q, K2 = 2 * K;
  const int QT = (Q + 15) >> 4, NTl = (N + 15) >> 4;
  const int r = blk / ngroups, grp = blk - r * ngroups;
  const int g0 = (NTl * grp) / ngroups, ng = (NTl * (grp + 1)) / ngroups - g0;      // this group's neuron tiles (host: <= NG)
  const int n0 = 16 * g0, nrow = (N - n0 < 16 * ng) ? N - n0 : 16 * ng;
  const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, lg = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int NTH = blockDim.x, NW = NTH >> 6;
  double* TR = sm;                                   // NW x 16 TRS    per wave: the transpose tile
  double* SL = TR + NW * 16 * TRS;                   // (NW - 1) x NACC x 64   second parts of cut node tiles
  double* red = SL + (NW - 1) * NACC * 64;           // 8
  double* LM = red + 8;                              // 16 NG x LDK    log2e [C | C^2/2] of the group's neurons
  double* lv = LM + 16 * NG * LDK;                   // 16 NG          log2e d_n (padding: the exponent offset)
  double* MQ = lv + 16 * NG;                         // 16 QT x LDK    [mu | var] of the trial's nodes
  double* wql = MQ + 16 * QT * LDK;                  // 16 QT          node weights (padding: 0)
  double* EP = LM;                                   // behind the loop, over the operand matrices: the waves' partial sums
  long long eck[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  SV_ECK(0);
  // ---- staging: every element of the operand regions is written exactly once (values, or the zeros of the padding).
  // Thread maps without an integer division per element (a first version spent as many vector instructions on index
  // arithmetic outside the pair loop as the loop itself took: SQ_INSTS_VALU of the profile against the loop's count):
  // C rows 16 at a time with lane & 15 = column (K <= 10 of 16 lanes load), nodes one per thread with the K loads unrolled.
  for (int i = tid; i < 16 * NG; i += NTH) lv[i] = (i < nrow) ? d[n0 + i] * L2E : SV_ELL_OFF;
  for (int i = tid; i < 16 * QT; i += NTH) wql[i] = (i < Q) ? ev.wq[(long long)r * Q + i] : 0.0;
  {
    // (all global loads of a thread -- its C elements and its nodes' statistics -- are requested before the first LDS store:
    // batches of four with their stores in between were five exposed memory latencies per block, 10 us of the kernel)
    const int c = tid & 15, rb = tid >> 4, RPI = NTH >> 4;          // column, row within a pass, rows per pass
    constexpr int CP = (16 * NG + 15) / 16;                         // passes at 16 rows each (NTH >= 256 takes fewer)
    const double* s0 = C + (long long)n0 * K;
    double cv[CP];
#pragma unroll
    for (int u = 0; u < CP; ++u) {
      const int row = u * RPI + rb;
      cv[u] = (row < nrow && c < K) ? s0[row * K + c] : 0.0;
    }
    const double* m0 = ev.muq + (long long)r * Q * K;               // [k][q]: K runs of Q contiguous nodes
    const double* m1 = ev.varq + (long long)r * Q * K;
    for (int node = tid; node < 16 * QT; node += NTH) {
      const bool ok = node < Q;
      double v0[2 * KS], v1[2 * KS];                                // (K <= 2 KS)
#pragma unroll
      for (int u = 0; u < 2 * KS; ++u) {
        const bool in = ok && u < K;
        v0[u] = in ? m0[u * Q + node] : 0.0;
        v1[u] = in ? m1[u * Q + node] : 0.0;
      }
      double* row = MQ + node * LDK;
#pragma unroll
      for (int u = 0; u < 2 * KS; ++u)
        if (u < K) {
          row[u] = v0[u];
          row[K + u] = v1[u];
        }
      for (int cc = K2; cc < LDK; ++cc) row[cc] = 0.0;
    }
#pragma unroll
    for (int u = 0; u < CP; ++u) {
      const int row = u * RPI + rb;
      if (row < 16 * NG) {
        if (c < K) {
          LM[row * LDK + c] = cv[u] * L2E;
          LM[row * LDK + K + c] = 0.5 * cv[u] * cv[u] * L2E;
        } else if (K2 + (c - K) < LDK) {
          LM[row * LDK + K2 + (c - K)] = 0.0;                       // the row's padding columns (LDK - 2K <= 4 <= 16 - K)
        }
      }
    }
  }
  SV_ECK(1);
  __syncthreads();
  SV_ECK(2);

  d4 dacc[NG][FTA];
  double daccl[NG][LGA], ddacc[NG];
#pragma unroll
  for (int lt = 0; lt < NG; ++lt) {
#pragma unroll
    for (int t = 0; t < FTA; ++t) dacc[lt][t] = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int g = 0; g < LGA; ++g) daccl[lt][g] = 0.0;
    ddacc[lt] = 0.0;
  }
  double esum = 0.0;
  double* tr = TR + wave * 16 * TRS;
  const long long plane = (long long)grp * ev.R * Q * K;              // this group's plane of the latent adjoints
  // this wave's range of the block's pairs (node-major)
  const int P = QT * ng;
  const int p0 = (P * wave) / NW, p1 = (P * (wave + 1)) / NW;
  const int qa = p0 / ng, lta = p0 - qa * ng;                            // first pair
  const int qb = (p1 - 1) / ng, ltb = (p1 - 1) - qb * ng + 1;            // last pair (+ 1)
  d4 acc[FTA];
  double accl[LGA];
  auto put_tile = [&](const int q) {                                    // acc / accl -> the group's plane of gmuq / gvarq
    const int oi = 16 * q + li;
    if (oi < Q) {
      const long long gi = plane + (long long)r * Q * K + oi;          // [r][k][q]: the 16 nodes of the tile are contiguous
      auto put = [&](const int kk, const double v) {
        if (kk < K) ev.gmuq[gi + (long long)kk * Q] = -v * LN2;
        else if (kk < K2) ev.gvarq[gi + (long long)(kk - K) * Q] = -v * LN2;
      };
#pragma unroll
      for (int t = 0; t < FT; ++t)
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) put(16 * t + lg + 4 * qq, acc[t][qq]);
#pragma unroll
      for (int g = 0; g < LG; ++g) put(16 * FT + 4 * g + lg, accl[g]);
    }
  };
  bool keep = false;                                  // the last node tile of the range is cut: its first part stays in acc
  for (int q = qa; q <= qb && p0 < p1; ++q) {
    const int lo = (q == qa) ? lta : 0, hi = (SV_ELLF_DIAG == 1) ? 0 : ((q == qb) ? ltb : ng);
    const double* mq = MQ + 16 * q * LDK;
    const double ov = wql[16 * q + li];
    double bfrag[KS];                                 // the tile's nodes as the B operand of the h product
#pragma unroll
    for (int c = 0; c < KS; ++c) bfrag[c] = mq[li * LDK + 4 * c + lg];
    double am[FTA][4], aml[LGA][4];                   // the same nodes, transposed: the A operand of the dC product
#pragma unroll
    for (int t = 0; t < FT; ++t)
#pragma unroll
      for (int c = 0; c < 4; ++c) am[t][c] = mq[(lg + 4 * c) * LDK + 16 * t + li];
#pragma unroll
    for (int g = 0; g < LG; ++g)
#pragma unroll
      for (int c = 0; c < 4; ++c) aml[g][c] = mq[(lg + 4 * c) * LDK + 16 * FT + 4 * g + (li & 3)];
#pragma unroll
    for (int t = 0; t < FTA; ++t) acc[t] = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int g = 0; g < LGA; ++g) accl[g] = 0.0;
    double colsum = 0.0;
#pragma unroll
    for (int lt = 0; lt < NG; ++lt) {
      if (lt >= lo && lt < hi) {                      // (wave-uniform)
        const double* pA = LM + (16 * lt + li) * LDK + lg;
        const double* pT = LM + (16 * lt + lg) * LDK + li;
        const double* pL = LM + (16 * lt + lg) * LDK + 16 * FT + (li & 3);
        const double* pv = lv + 16 * lt + lg;
        d4 h = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int c = 0; c < KS; ++c) h = __builtin_amdgcn_mfma_f64_16x16x4f64(lds_frag(pA + 4 * c), bfrag[c], h, 0, 0, 0);
        double afrag[FTA][4], afl[LGA][4];
#pragma unroll
        for (int t = 0; t < FT; ++t)
#pragma unroll
          for (int c = 0; c < 4; ++c) afrag[t][c] = pT[4 * c * LDK + 16 * t];
#pragma unroll
        for (int g = 0; g < LG; ++g)
#pragma unroll
          for (int c = 0; c < 4; ++c) afl[g][c] = pL[4 * c * LDK + 4 * g];
        double E[4];
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
          E[qq] = ov * sv_exp2_neg(h[qq] + pv[4 * qq]);       // E^T[neuron lg + 4 qq][node li]
          colsum += E[qq];
          tr[(lg + 4 * qq) * TRS + li] = E[qq];
        }
        // (one wavefront's LDS instructions execute in order: the transposed reads see the writes above, and the next pair's
        // writes cannot pass these reads -- only the COMPILER has to be kept from moving them, hence no s_waitcnt-carrying fence)
        asm volatile("" ::: "memory");
        double ET[4];                                 // E[node lg + 4 c][neuron li]
#pragma unroll
        for (int c = 0; c < 4; ++c) ET[c] = lds_frag(tr + li * TRS + lg + 4 * c);
        asm volatile("" ::: "memory");
#pragma unroll
        for (int t = 0; t < FT; ++t)
#pragma unroll
          for (int c = 0; c < 4; ++c) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(afrag[t][c], E[c], acc[t], 0, 0, 0);
#pragma unroll
        for (int g = 0; g < LG; ++g)
#pragma unroll
          for (int c = 0; c < 4; ++c) accl[g] = __builtin_amdgcn_mfma_f64_4x4x4f64(afl[g][c], E[c], accl[g], 0, 0, 0);
        ddacc[lt] += (ET[0] + ET[1]) + (ET[2] + ET[3]);
#pragma unroll
        for (int t = 0; t < FT; ++t)
#pragma unroll
          for (int c = 0; c < 4; ++c) dacc[lt][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(am[t][c], ET[c], dacc[lt][t], 0, 0, 0);
#pragma unroll
        for (int g = 0; g < LG; ++g)
#pragma unroll
          for (int c = 0; c < 4; ++c) daccl[lt][g] = __builtin_amdgcn_mfma_f64_4x4x4f64(aml[g][c], ET[c], daccl[lt][g], 0, 0, 0);
      }
    }
    esum += colsum;
    if (lo == 0 && hi == ng) {
      put_tile(q);                                    // a whole node tile
    } else if (lo > 0) {                              // the second part of a tile cut by the range boundary in front of this wave
      double* sl = SL + (wave - 1) * NACC * 64 + lane;
#pragma unroll
      for (int t = 0; t < FT; ++t)
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) sl[(4 * t + qq) * 64] = acc[t][qq];
#pragma unroll
      for (int g = 0; g < LG; ++g) sl[(4 * FT + g) * 64] = accl[g];
    } else {
      keep = true;                                    // the first part: finished behind the barrier (q == qb, the loop ends)
    }
  }
  SV_ECK(3);
  __syncthreads();                                    // every wave is done with the operand matrices; the slots are written
  SV_ECK(4);
  if (keep) {
    const double* sl = SL + wave * NACC * 64 + lane;
#pragma unroll
    for (int t = 0; t < FT; ++t)
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) acc[t][qq] += sl[(4 * t + qq) * 64];
#pragma unroll
    for (int g = 0; g < LG; ++g) accl[g] += sl[(4 * FT + g) * 64];
    put_tile(qb);
  }
  // ---- the waves' partial [A1 | A2] rows and column sums.  Neuron tile lt is summed, in wave order, and written by wave
  // lt % NW: every wave stores the partials of the tiles it does not own as register images over the operand matrices (dead
  // since the barrier), one barrier, the owners add.  (Round 5, first version: the waves added their partials one after the
  // other into one buffer, four barriers with one wave working at a time -- 13 of the kernel's 85 us.)
  if (SV_ELLF_DIAG == 2) {
#pragma unroll
    for (int lt = 0; lt < NG; ++lt) asm volatile("" ::"v"(dacc[lt][0][0]), "v"(daccl[lt][0]), "v"(ddacc[lt]));
    return;
  }
  double* DD = EP + (NW - 1) * NG * NACC * 64;        // [wave][NG][16] column sums
#pragma unroll
  for (int lt = 0; lt < NG; ++lt) {
    if (lt < ng) {
      const int o = lt % NW;
      if (o != wave) {
        double* sl = EP + (lt * (NW - 1) + (wave < o ? wave : wave - 1)) * NACC * 64 + lane;
#pragma unroll
        for (int t = 0; t < FT; ++t)
#pragma unroll
          for (int qq = 0; qq < 4; ++qq) sl[(4 * t + qq) * 64] = dacc[lt][t][qq];
#pragma unroll
        for (int g = 0; g < LG; ++g) sl[(4 * FT + g) * 64] = daccl[lt][g];
      }
      const double sdd = grp4_sum(ddacc[lt]);
      if (lg == 0) DD[(wave * NG + lt) * 16 + li] = sdd;
    }
  }
  SV_ECK(5);
  __syncthreads();
  SV_ECK(6);
#pragma unroll
  for (int lt = 0; lt < NG; ++lt) {
    if (lt < ng && lt % NW == wave) {
      d4 tot4[FTA];
      double totl[LGA];
      for (int w = 0; w < NW; ++w) {
        const double* sl = EP + (lt * (NW - 1) + (w < wave ? w : w - 1)) * NACC * 64 + lane;
#pragma unroll
        for (int t = 0; t < FT; ++t)
#pragma unroll
          for (int qq = 0; qq < 4; ++qq) {
            const double v = (w == wave) ? dacc[lt][t][qq] : sl[(4 * t + qq) * 64];
            tot4[t][qq] = (w == 0) ? v : tot4[t][qq] + v;
          }
#pragma unroll
        for (int g = 0; g < LG; ++g) {
          const double v = (w == wave) ? daccl[lt][g] : sl[(4 * FT + g) * 64];
          totl[g] = (w == 0) ? v : totl[g] + v;
        }
      }
      const int n = n0 + 16 * lt + li;
      if (n < N) {
        double* dst = ev.gCpart + ((long long)r * N + n) * K2;
#pragma unroll
        for (int t = 0; t < FT; ++t)
#pragma unroll
          for (int qq = 0; qq < 4; ++qq) {
            const int kk = 16 * t + lg + 4 * qq;
            if (kk < K2) dst[kk] = tot4[t][qq];
          }
#pragma unroll
        for (int g = 0; g < LG; ++g) {
          const int kk = 16 * FT + 4 * g + lg;
          if (kk < K2) dst[kk] = totl[g];
        }
        if (lg == 0) {
          double sdd = DD[lt * 16 + li];
          for (int w = 1; w < NW; ++w) sdd += DD[(w * NG + lt) * 16 + li];
          ev.gdpart[(long long)r * N + n] = -sdd;
        }
      }
    }
  }
  const double tot = block_sum(esum, red);
  if (tid == 0) ev.ellq[blk] = -tot;
  if (SV_ELLF_CLK) {
    SV_ECK(7);
    if ((blk == 5 || blk == 300 || blk == 800) && (tid == 0 || tid == 192))
      printf("ellf blk %d tid %d start %lld: staged %lld barrier %lld loop %lld barrier %lld partials %lld barrier %lld end %lld (10 ns ticks)\n",
             blk, tid, eck[0], eck[1] - eck[0], eck[2] - eck[0], eck[3] - eck[0], eck[4] - eck[0], eck[5] - eck[0], eck[6] - eck[0],
             eck[7] - eck[0]);
  }
}

#define SV_ELLF_NG 7        // neuron tiles per group of the fused ELL kernel (accumulators: NG (KS + 1) doubles per lane)
#define SV_ELLF_KSMAX 5     // 2K <= 20 (KS = 6 takes 255 registers and spills four)
#define SV_ELLF_NW 4        // waves per block
__host__ __device__ constexpr int sv_ellf_lds_doubles(int KS, int QT) {
  const int nacc = 4 * (KS / 4) + KS % 4, ldk = 4 * KS + 2;
  const int operands = 16 * SV_ELLF_NG * ldk + 16 * SV_ELLF_NG + 16 * QT * ldk + 16 * QT;
  const int partials = (SV_ELLF_NW - 1) * SV_ELLF_NG * nacc * 64 + SV_ELLF_NW * SV_ELLF_NG * 16;     // the epilogue's, over the operands
  return SV_ELLF_NW * 16 * SV_ELLF_TRS + (SV_ELLF_NW - 1) * nacc * 64 + 8 + (operands > partials ? operands : partials);
}
template <int KS>
__global__ __launch_bounds__(64 * SV_ELLF_NW, 2) void k_quad_ell_fused(EngineView ev, const double* C, const double* d,
                                                                       int ngroups, int ncomb, int comb_full) {
  extern __shared__ double sm[];
  const int nell = ev.R * ngroups;
  if ((int)blockIdx.x >= nell) {                       // rider blocks: the spike moments' combine (as in k_quad_ell_mfma)
    const int nwb = (int)blockDim.x >> 6, wave = (int)threadIdx.x >> 6;
    sv_comb_rider(ev, ((int)blockIdx.x - nell) * nwb + wave, ncomb, comb_full, sm, wave);
    return;
  }
  quad_ell_fused_body<KS, SV_ELLF_NG>(ev, C, d, ngroups, blockIdx.x, sm);
}
