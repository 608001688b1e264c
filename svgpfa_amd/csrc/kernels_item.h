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
  const size_t lt = (size_t)64 * MP, rider = cls ? (sizeof(SpikeClassScratch) + 7) / 8 + (size_t)K * (SV_FH + 3) : 0;
  return 8 * (lt > rider ? lt : rider);
}

template <int MP>
__global__ __launch_bounds__(64) void k_build_reg(EngineView ev, int write_aux, int snap, SnapTable st, int cls) {
  constexpr int IPW = 64 / MP;
  // Lt[item][q][i] = L[i][q]  (column-major: conflict-free writes); the rider blocks in front use the same bytes
  extern __shared__ double smraw[];        // max(IPW MP MP, class rider: sv_build_lds) doubles
  double (*Lt)[MP][MP] = reinterpret_cast<double (*)[MP][MP]>(smraw);
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
        if (i <= q) ev.Linv[tbase + sv_tri_off(q) + i] = x[q];     // column i of the packed lower triangle: lanes write consecutive addresses
        if (i == q && !(q & 1)) ev.Linv[tbase + sv_tri_off(q) + q + 1] = 0.0;   // the pad of an odd-length row
        if (write_aux) ev.L[base + i * M + q] = a[q];            // row i
      }
  }
}
