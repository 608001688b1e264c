// Per-(trial, latent) algebra.  Kernels:
//   k_build_reg (this file) : Kzz + eps I, Cholesky L, Linv = L^-1   (reference kernelsMatricesStore.py:107-117, miscUtils.py:209-216)
//   k_prep_mfma (kernels_mfma.h) : ut = Linv m, a = Linv^T ut (= Kzz^-1 m), Vt = Linv L_S, KL
//             (reference svPosteriorOnLatents.py:185-216 algebra in whitened form, klDivergence.py:31-44)
//   k_item_bwd (kernels_item_bwd.h) : adjoints of (ut, Vt Vt^T, KL, direct kappa terms) -> gradients wrt m, cholVecs, Z, theta
// Derivation (per trial/latent; Kzz = L L^T, S = L_S L_S^T, kt = Linv k(t)):
//   mu(t) = kt.ut,  var(t) = s2 - |kt|^2 + |Vt^T kt|^2,  KL = (|Vt|_F^2 + |ut|^2 + logdet Kzz - logdet S - M)/2
//   with gt = dELL/d ut and Gt = sum_t dELL/dvar(t) kt kt^T (adjoint of Vt Vt^T):
//   dELBO/dm    = Linv^T (gt - ut)
//   dELBO/dL_S  = tril(Linv^T (2 Gt - I) Vt) + diag(1 / L_S,ii)
//   dELBO/dKzz  = Linv^T At Linv,  At = -(gt ut^T + ut gt^T)/2 + ut ut^T/2 + Gt + (Vt Vt^T - I)/2 - C - C^T,  C = Vt Vt^T Gt
#pragma once
#include "engine_common.h"
#include "spike_cls_common.h"

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
// dynamic LDS of k_build_reg<MP> (bytes); cls: with the class riders, whose series workspace has one row per latent
__host__ __device__ constexpr size_t sv_build_lds(int MP, int K, bool cls) {
  const size_t lt = (size_t)32 * (MP + 1) + 64, rider = cls ? (sizeof(SpikeClassScratch) + 7) / 8 + (size_t)K * (SV_FH + 3) : 0;
  return 8 * (lt > rider ? lt : rider);
}

// WPE: waves per SIMD the compiler leaves room for.  At MP = 32 the kernel takes 185 registers (two waves per SIMD: 2 048
// single-wave blocks resident), and the north-star size launches 3 010 blocks -- TWO rounds of a latency-bound kernel (39.9 us
// against 19.4 us for the one round of an 8-way shard).  The WPE = 3 instantiation (168 registers, 10 spilled) holds 3 072
// blocks at once: one round of 31 us.  On a launch that is one round either way it is the slower one (23.7 against ~20 us
// at 1 516 blocks), so the host picks it only above 2 048 blocks (launch_build).
template <int MP, int WPE = 1>
__global__ __launch_bounds__(64, WPE) void k_build_reg(EngineView ev, int write_aux, int snap, SnapTable st, int cls) {
  constexpr int IPW = 64 / MP;
  // Lt: column q of an item's L at q MP - q (q - 1) / 2, rows i >= q only (column-major: conflict-free writes; packed:
  // 8.4 KB per block at MP = 32, so that LDS does not cap the resident blocks below what the registers allow); the rider
  // blocks in front use the same bytes
  extern __shared__ double smraw[];        // max(IPW MP (MP + 1) / 2, class rider: sv_build_lds) doubles
  constexpr int PK = MP * (MP + 1) / 2;
  const int lane = threadIdx.x, sub = lane / MP, i = lane % MP, lbase = lane - i;
  const int rp = (ev.R + IPW - 1) / IPW;
  const int nser = snap ? ev.K : 0, ncb = (snap && cls) ? ev.R : 0;
  const int nx = nser + ncb;
  if ((int)blockIdx.x < nser) {
    // K extra blocks in front of the item blocks (snap launches only): the cosine series of latent k's periodic kernel
    // for the spike pass (engine_common.h periodic_series) -- a serial recurrence on one lane that runs beside the item
    // blocks instead of lengthening one of them
    const int ks = blockIdx.x;
    if (lane == 0) {
      const LatentMeta ms = ev.meta[ks];
      const KPar ps = make_kpar(ms, st.theta[ks]);
      ev.fourH[ks] = (ms.ktype == SV_PERIODIC && ev.per_series)
                         ? periodic_series(0.5 * ps.h, ms.M, ev.four + (size_t)ks * SV_FROWS * (SV_FH + 2), smraw)
                         : -1;
    }
    return;
  }
  if ((int)blockIdx.x < nx) {
    // R more (cls launches): the spike pass's class table and trial r's entries of it (spike_cls_common.h), from the
    // caller's Z / theta -- the table depends on nothing this launch computes, and riding here saves the class path a
    // launch, an event and a cross-stream wait per kernel build
    SpikeClassScratch& cs = *reinterpret_cast<SpikeClassScratch*>(smraw);
    double* fws = smraw + (sizeof(SpikeClassScratch) + 7) / 8;
    const int rc = blockIdx.x - nser;
    spike_class_table(ev, cs, &st, fws);
    lds_wave_fence();
    if (rc == 0) spike_class_store(ev, cs);
    spike_class_trial(ev, cs, &st, rc);
    return;
  }
  const int bid = blockIdx.x - nx;
  const int k = bid / rp;
  int r = (bid % rp) * IPW + sub;
  const bool active = r < ev.R;
  if (!active) r = ev.R - 1;
  const LatentMeta m = ev.meta[k];
  const int M = m.M;
  const KPar p = make_kpar(m, snap ? st.theta[k] : ev.theta + 2 * k);
  const long long base = m.offMM + (long long)r * M * M;
  const long long tbase = m.offTT + (long long)r * m.PT;
  const double* z = (snap ? st.Z[k] : ev.Zs + m.offM) + (long long)r * M;
  const bool rowok = i < M;
  const double zi = rowok ? z[i] : 0.0;
  if (snap) {
    if (active && rowok) ev.Zs[m.offM + (long long)r * M + i] = zi;
    if (bid % rp == 0 && lane < m.nth) ev.theta[2 * k + lane] = st.theta[k][lane];
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
  double* Lrow = smraw + sub * PK;
#pragma unroll
  for (int j = 0; j < MP; ++j) {
    double s = a[j];
    double sx = (i == j) ? 1.0 : 0.0;
#pragma unroll
    for (int q = 0; q < j; ++q) {
      const double l = Lrow[q * MP - q * (q - 1) / 2 + (j - q)];        // L[j][q]: same address in every lane of the item
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
    // (lanes above the diagonal store to a dump slot of their own: a select on the address, where a conditional store
    // put a branch into each of the MP unrolled steps and the register allocation fell apart -- 112 spills at MP = 32)
    smraw[(i >= j) ? sub * PK + j * MP - j * (j - 1) / 2 + (i - j) : IPW * PK + lane] = a[j];
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
        if (i <= q) ev.Linv[tbase + sv_tri_off(q) + i] = x[q];     // column i of the packed lower triangle: lanes write consecutive addresses
        if (i == q && !(q & 1)) ev.Linv[tbase + sv_tri_off(q) + q + 1] = 0.0;   // the pad of an odd-length row
        if (write_aux) ev.L[base + i * M + q] = a[q];            // row i
      }
  }
}

// ------------------------------------------------------------------------------ k_build_mfma
// The same build for 16 < M <= 64 in 16 x 16 TILES (round 4): right-looking blocked Cholesky with the explicit inverse,
//   for jb:  L_jj = chol(A_jj), X_jj = L_jj^-1        16 lanes per item, the register algorithm of k_build_reg on one tile
//            L_ib,jb = A_ib,jb X_jj^T   (ib > jb)      f64 MFMA
//            A_ib,kb -= L_ib,jb L_kb,jb^T              f64 MFMA
//   then     X_ib,jb = -X_ib,ib sum_{kb = jb}^{ib - 1} L_ib,kb X_kb,jb
// One wavefront works on IPW items of ONE latent: the diagonal-tile steps of all of them side by side (lanes 16 g .. 16 g +
// 15 = item g), the tile products item after item on all 64 lanes.  Against k_build_reg<32> -- 32 pivot steps with inner
// loops of up to 31 links, fully unrolled -- a diagonal tile is 16 steps with at most 15 links, a wave carries four items
// instead of two, and everything of order M^3 is on the matrix unit: ~800 instead of ~1500 issued instructions per item.
// Tiles of an item live in LDS, lower block triangle, rows padded to 18 doubles (conflict-free MFMA fragment reads); a tile's
// memory doubles as the row-broadcast buffer of its own factorisation and ends up holding the tile of Linv.
// grid = K ceil(R / IPW) blocks of 64 threads behind the same series / class rider blocks as k_build_reg.
#define SV_BT_LD 18
#define SV_BT_SZ (16 * SV_BT_LD)
#ifndef SV_BUILD_IPW2
#define SV_BUILD_IPW2 4      // items per wavefront at 16 < M <= 32 (experiment builds: 1, 2)
#endif
__host__ __device__ constexpr int sv_build_ipw(int NT) { return NT <= 2 ? SV_BUILD_IPW2 : (NT == 3 ? 2 : 1); }
__host__ __device__ constexpr size_t sv_build_mfma_lds(int NT, int K, bool cls) {
  const size_t ntri = (size_t)NT * (NT + 1) / 2;
  const size_t it = (size_t)sv_build_ipw(NT) * (ntri * SV_BT_SZ + 3 * 16 * (size_t)NT);
  const size_t rider = cls ? (sizeof(SpikeClassScratch) + 7) / 8 + (size_t)K * (SV_FH + 3) : (size_t)(SV_FH + 3);
  return 8 * (it > rider ? it : rider);
}

// broadcast of lane (lane & 48) + J's value inside every group of 16 lanes (ds_swizzle, bit mode: no address register)
template <int J>
__device__ __forceinline__ double bcast16(double x) {
  constexpr int PAT = 0x10 | (J << 5);        // and_mask 0x10 (keep the group bit inside a half-wave), or_mask J
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_ds_swizzle(lo, PAT);
  hi = __builtin_amdgcn_ds_swizzle(hi, PAT);
  return __hiloint2double(hi, lo);
}

// Cholesky factor and inverse of one 16 x 16 tile per group of 16 lanes.  T: the group's tile in LDS (symmetric, rows padded
// to SV_BT_LD); on return it holds X = L^-1 row-major; the rows of L stay in a[] (lane i: row i).  Every lane of the wave
// calls (a group without an item works on a copy of another group's tile and stores nothing).
template <int J>
struct DiagStep {
  static __device__ __forceinline__ void run(double (&a)[16], double (&x)[16], double* T, const int i, bool& bad) {
    double s = a[J];
    double sx = (i == J) ? 1.0 : 0.0;
#pragma unroll
    for (int q = 0; q < J; ++q) {
      const double l = T[q * SV_BT_LD + J];       // L[J][q], written as column q at step q: one address per group
      s = fma(-a[q], l, s);
      sx = fma(-l, x[q], sx);
    }
    const double d = bcast16<J>(s);
    if (!(d > 0.0)) bad = true;
    double ri = __builtin_amdgcn_rsq(d);
    ri = ri * fma(-0.5 * d * ri, ri, 1.5);
    ri = ri * fma(-0.5 * d * ri, ri, 1.5);
    a[J] = (i > J) ? s * ri : ((i == J) ? d * ri : 0.0);
    x[J] = sx * ri;
    asm volatile("" : "+v"(x[J]));       // (evaluated HERE: left alone, the compiler defers the whole sx chain of the 16 steps and keeps every l alive: 256 registers and 160 spilled)
    T[J * SV_BT_LD + i] = a[J];                   // column J of L for the later steps
    lds_wave_fence();
    DiagStep<J + 1>::run(a, x, T, i, bad);
  }
};
template <>
struct DiagStep<16> {
  static __device__ __forceinline__ void run(double (&)[16], double (&)[16], double*, const int, bool&) {}
};

#ifndef SV_BUILD_WPE2
#define SV_BUILD_WPE2 2      // waves per SIMD the compiler leaves room for at 16 < M <= 32 (experiment builds)
#endif
template <int NT>
__global__ __launch_bounds__(64, (NT <= 2 ? SV_BUILD_WPE2 : 2)) void k_build_mfma(EngineView ev, int write_aux, int snap, SnapTable st, int cls) {
  constexpr int IPW = sv_build_ipw(NT), MP = 16 * NT, NTRI = NT * (NT + 1) / 2, LDT = SV_BT_LD, TS = SV_BT_SZ;
  constexpr int ITEM_SZ = NTRI * TS + 3 * MP;          // doubles of LDS per item: tiles, Z, sin / cos of the phases
  extern __shared__ double smraw[];
  const int lane = threadIdx.x, li = lane & 15, lg = lane >> 4;
  const int rp = (ev.R + IPW - 1) / IPW;
  const int nser = snap ? ev.K : 0, ncb = (snap && cls) ? ev.R : 0;
  const int nx = nser + ncb;
  if ((int)blockIdx.x < nser) {                          // the periodic kernels' cosine series (see k_build_reg)
    const int ks = blockIdx.x;
    if (lane == 0) {
      const LatentMeta ms = ev.meta[ks];
      const KPar ps = make_kpar(ms, st.theta[ks]);
      ev.fourH[ks] = (ms.ktype == SV_PERIODIC && ev.per_series)
                         ? periodic_series(0.5 * ps.h, ms.M, ev.four + (size_t)ks * SV_FROWS * (SV_FH + 2), smraw)
                         : -1;
    }
    return;
  }
  if ((int)blockIdx.x < nx) {                            // the spike pass's class table and trial entries (see k_build_reg)
    SpikeClassScratch& cs = *reinterpret_cast<SpikeClassScratch*>(smraw);
    double* fws = smraw + (sizeof(SpikeClassScratch) + 7) / 8;
    const int rc = blockIdx.x - nser;
    spike_class_table(ev, cs, &st, fws);
    lds_wave_fence();
    if (rc == 0) spike_class_store(ev, cs);
    spike_class_trial(ev, cs, &st, rc);
    return;
  }
  const int bid = blockIdx.x - nx;
  const int k = bid / rp, r0 = (bid % rp) * IPW;
  const int nit = (ev.R - r0 < IPW) ? ev.R - r0 : IPW;  // items of this wave (>= 1)
  const LatentMeta m = ev.meta[k];
  const int M = m.M;
  const KPar p = make_kpar(m, snap ? st.theta[k] : ev.theta + 2 * k);
  auto tile_of = [](int ib, int jb) { return ib * (ib + 1) / 2 + jb; };

  // ---- Z of the wave's items -> LDS (and the snapshot), sin / cos of the phases for periodic kernels
  for (int e = lane; e < IPW * MP; e += 64) {
    const int g = e / MP, j = e % MP;
    double z = 0.0;
    if (g < nit && j < M) {
      z = (snap ? st.Z[k] : ev.Zs + m.offM)[(long long)(r0 + g) * M + j];
      if (snap) ev.Zs[m.offM + (long long)(r0 + g) * M + j] = z;
    }
    double* zl = smraw + g * ITEM_SZ + NTRI * TS;
    zl[j] = z;
    if (m.ktype == SV_PERIODIC) sv_sincos_quarter(z * p.q4, zl[MP + j], zl[2 * MP + j]);
  }
  if (snap && bid % rp == 0 && lane < m.nth) ev.theta[2 * k + lane] = st.theta[k][lane];
  lds_wave_fence();

  // ---- Kzz + eps I, lower block triangle, four entries per lane and tile; padding = identity
  for (int g = 0; g < nit; ++g) {
    double* it = smraw + g * ITEM_SZ;
    const double* zl = it + NTRI * TS;
    const long long base = m.offMM + (long long)(r0 + g) * M * M;
#pragma unroll
    for (int ib = 0; ib < NT; ++ib)
#pragma unroll
      for (int jb = 0; jb <= ib; ++jb) {
        double* T = it + tile_of(ib, jb) * TS;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int rr = lg + 4 * u, i = 16 * ib + rr, j = 16 * jb + li;
          double v;
          if (m.ktype == SV_RBF) {
            const double d = zl[i] - zl[j];
            v = p.s2 * sv_exp2_neg(-(d * d) * p.h2);
          } else {
            const double cs = fma(zl[2 * MP + i], zl[2 * MP + j], zl[MP + i] * zl[MP + j]);
            v = p.s2 * sv_exp2_neg(fmin(fma(cs, p.h2, -p.h2), 0.0));
          }
          const bool in = i < M && j < M;
          v = (i == j) ? (in ? p.s2 + ev.reg : 1.0) : (in ? v : 0.0);
          T[rr * LDT + li] = v;
          if (write_aux && in) {
            ev.Kzz[base + (long long)i * M + j] = v;
            if (ib != jb) {
              ev.Kzz[base + (long long)j * M + i] = v;
              ev.L[base + (long long)j * M + i] = 0.0;      // L's tiles above the diagonal (the factorisation below writes the rest;
            }                                               //  round 5: they were left as the allocation held them)
          }
        }
      }
  }
  lds_wave_fence();

  // ---- blocked factorisation
  const int gd = (lg < nit) ? lg : nit - 1;              // the item this lane's group factors (idle groups shadow the last item)
  double* itd = smraw + gd * ITEM_SZ;
  const bool own = lg < nit;
  bool bad = false;
#pragma unroll
  for (int jb = 0; jb < NT; ++jb) {
    {
      double* T = itd + tile_of(jb, jb) * TS;
      double a[16], x[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) a[q] = T[li * LDT + q];
      lds_wave_fence();                                  // (the tile becomes the broadcast buffer of its own columns)
      DiagStep<0>::run(a, x, T, li, bad);
      if (write_aux && own && 16 * jb + li < M) {
        const long long base = m.offMM + (long long)(r0 + lg) * M * M;
#pragma unroll
        for (int q = 0; q < 16; ++q)
          if (16 * jb + q < M) ev.L[base + (long long)(16 * jb + li) * M + 16 * jb + q] = a[q];
      }
      if (own) {
#pragma unroll
        for (int q = 0; q < 16; ++q) T[q * LDT + li] = x[q];      // X_jj row-major: x[q] of lane i is X[q][i]
      }
      lds_wave_fence();
    }
    if (jb + 1 < NT) {
      for (int g = 0; g < nit; ++g) {                    // panel below the diagonal tile: L_ib,jb = A_ib,jb X_jj^T
        double* it = smraw + g * ITEM_SZ;
        const double* X = it + tile_of(jb, jb) * TS;
#pragma unroll
        for (int ib = jb + 1; ib < NT; ++ib) {
          double* T = it + tile_of(ib, jb) * TS;
          d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
          for (int s = 0; s < 4; ++s)
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(T[li * LDT + 4 * s + lg], X[li * LDT + 4 * s + lg], acc, 0, 0, 0);
          lds_wave_fence();
#pragma unroll
          for (int q = 0; q < 4; ++q) T[(lg + 4 * q) * LDT + li] = acc[q];
          if (write_aux) {
            const long long base = m.offMM + (long long)(r0 + g) * M * M;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const int i = 16 * ib + lg + 4 * q, j = 16 * jb + li;
              if (i < M && j < M) ev.L[base + (long long)i * M + j] = acc[q];
            }
          }
        }
      }
      lds_wave_fence();
      for (int g = 0; g < nit; ++g) {                    // trailing update: A_ib,kb -= L_ib,jb L_kb,jb^T
        double* it = smraw + g * ITEM_SZ;
#pragma unroll
        for (int ib = jb + 1; ib < NT; ++ib)
#pragma unroll
          for (int kb = jb + 1; kb <= ib; ++kb) {
            double* T = it + tile_of(ib, kb) * TS;
            const double* Li = it + tile_of(ib, jb) * TS;
            const double* Lk = it + tile_of(kb, jb) * TS;
            d4 acc;
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = T[(lg + 4 * q) * LDT + li];
#pragma unroll
            for (int s = 0; s < 4; ++s)
              acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-Li[li * LDT + 4 * s + lg], Lk[li * LDT + 4 * s + lg], acc, 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 4; ++q) T[(lg + 4 * q) * LDT + li] = acc[q];
          }
      }
      lds_wave_fence();
    }
  }
  if (bad && own && li == 0) atomicCAS(ev.status, 0, 1 + (r0 + lg) * ev.K + k);

  // ---- the tiles of Linv below the diagonal, in place over L's: increasing jb inside increasing ib never overwrites a tile
  //      of L that a later product still needs
  for (int g = 0; g < nit; ++g) {
    double* it = smraw + g * ITEM_SZ;
#pragma unroll
    for (int ib = 1; ib < NT; ++ib)
#pragma unroll
      for (int jb = 0; jb < ib; ++jb) {
        d4 S = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int kb = jb; kb < ib; ++kb) {
          const double* Lt = it + tile_of(ib, kb) * TS;
          const double* Xt = it + tile_of(kb, jb) * TS;
#pragma unroll
          for (int s = 0; s < 4; ++s)
            S = __builtin_amdgcn_mfma_f64_16x16x4f64(Lt[li * LDT + 4 * s + lg], Xt[(4 * s + lg) * LDT + li], S, 0, 0, 0);
        }
        const double* Xd = it + tile_of(ib, ib) * TS;
        d4 Xn = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int q = 0; q < 4; ++q)                      // register q of S holds row lg + 4 q: the contraction index of step q
          Xn = __builtin_amdgcn_mfma_f64_16x16x4f64(-Xd[li * LDT + lg + 4 * q], S[q], Xn, 0, 0, 0);
        lds_wave_fence();                                // every read of tile (ib, jb) as L is done
        double* T = it + tile_of(ib, jb) * TS;
#pragma unroll
        for (int q = 0; q < 4; ++q) T[(lg + 4 * q) * LDT + li] = Xn[q];
        lds_wave_fence();
      }
  }

  // ---- Linv -> HBM, packed lower triangle (rows padded to an even length with a zero)
  for (int g = 0; g < nit; ++g) {
    const double* it = smraw + g * ITEM_SZ;
    const long long tbase = m.offTT + (long long)(r0 + g) * m.PT;
    for (int i = 0; i < M; ++i) {
      const int ib = i >> 4;
      if (lane <= i) {
        const int jb = lane >> 4;
        ev.Linv[tbase + sv_tri_off(i) + lane] = it[tile_of(ib, jb) * TS + (i & 15) * LDT + (lane & 15)];
      } else if (lane == i + 1 && !(i & 1)) {
        ev.Linv[tbase + sv_tri_off(i) + lane] = 0.0;       // the pad of an odd-length row
      }
    }
  }
}
