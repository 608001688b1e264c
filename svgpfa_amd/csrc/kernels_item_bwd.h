// k_item_bwd: everything downstream of the expected log-likelihood for one (trial, latent) item, in ONE kernel:
//   phase 1  adjoints of the quadrature term wrt ut (gt), Vt Vt^T (Gt) and -- FULL -- the direct dependence on
//            Z / theta through kappa(t_q, Z)                       (was k_quad_bwd_mfma)
//   phase 2  backward through the whitened algebra: gradients wrt m, cholVecs, Z, theta of the item
//            (derivation: kernels_item.h header)                  (was k_bwd_mfma)
//   phase 3  the fixed-order reductions of the per-trial dC / dd partials into the `shared` vector
//            [ELBO, ELL, KL, dC, dd, dtheta]                      (was most of k_finalize; the rest: k_fin_items)
// Linv, a, ut, Z and the adjoints gt / Gt stay in LDS between phases 1 and 2 (the two-kernel version wrote Gt to HBM and
// re-staged Linv / Gt: 164 MB and a 34 us fill phase per evaluation at the north-star size).
// Phase 3: the dC / dd sums (and the KL / quadrature-ELL scalars) only read partials of EARLIER kernels (spike pass, ELL
// kernel, k_prep), so they are dealt out over the blocks of this grid; dtheta and the spike ELL need this kernel's own
// per-item results and are summed by the small k_fin_items launch behind it.  Every sum keeps a fixed order, so results
// stay bit-reproducible.
// Replaces autograd backward of the reference (stats/svEM.py:274-288) for stats/svPosteriorOnLatents.py:185-216,
// stats/klDivergence.py:18-44, stats/kernelsMatricesStore.py:107-117.
#pragma once
#include "engine_common.h"
#include "kernels_mfma.h"
#include "spike_cls_common.h"

#ifndef SV_BWD_CLK
#define SV_BWD_CLK 0    // diagnosis builds only: 1 = one thread of one block prints cycle stamps of the phases (device printf)
#endif
#define SV_CK(i) do { if (SV_BWD_CLK) ck[i] = clock64(); } while (0)
#ifndef SV_BWD_STOP
#define SV_BWD_STOP 0   // diagnosis builds only (tests/tools/phase_times.sh): 1 stop after phase 1, 2 after the spike gather, 3 before phase 3, 6 / 7 / 8 inside phase 2, 9 / 10 inside phase 1
#endif
#define SV_MASK_STATS 16u   // internal mask bit: svgpfa_latent_stats_backward (adjoints at quadrature nodes AND spikes given)
#define SV_FIN_OUT 8   // outputs per reduction group of phase 3 (8 consecutive doubles = one 64-B segment per trial)

// doubles of scratch the gather of phase 2's inputs needs when it combines the spike moments itself (gather_item_inputs:
// the class's moments + three coefficient rows); it borrows the Gt region, which nothing else uses before the reductions
#define SV_GATHER_SCR (SV_ACOLS + 3 * (SV_FH + 1))

// Phase 1 as panel owners (teams of NT waves, item_bwd_body) from this many tiles per side on; below, a wave owns whole
// node tiles (no exchange, no barrier inside the loop: with one or two panels the team would be one or two waves, and the
// per-tile set-up -- node loads, the node's sincos -- would be repeated by every wave of the team)
#ifndef SV_BWD_PO_MIN
#define SV_BWD_PO_MIN 3
#endif
__host__ __device__ constexpr bool sv_bwd_po(int NT) { return NT >= SV_BWD_PO_MIN; }
// transcendental kernels of the node-tile loop: polynomial coefficients in SGPR pairs (sv_math.h) or left to the compiler
#ifndef SV_BWD_SC
#define SV_BWD_SC 0
#endif
#if SV_BWD_SC
#define SV_BWD_EXP2 sv_exp2_neg_sc
#define SV_BWD_SINCOS sv_sincos_quarter_sc
#else
#define SV_BWD_EXP2 sv_exp2_neg
#define SV_BWD_SINCOS sv_sincos_quarter
#endif

// Nodes of an item's list staged in LDS at a time (times + the two adjoints, 24 bytes per node): the tile loops read them
// with immediate offsets instead of eleven predicated global loads, their 64-bit address arithmetic and their selects per
// tile and wave.  256 covers the quadrature nodes of every BASELINE config in one chunk; M = 64 has 3 KB to spare.
__host__ __device__ constexpr int sv_bwd_qc(int NT) { return NT >= 4 ? 128 : 256; }

// LDS layout of one block (offsets in doubles) -- ONE constexpr function shared by the kernel (which places its buffers with
// it) and the host (which sizes the launch with it), so the two cannot disagree.  A block is G = NWB / NT teams of NT
// waves; wave c of a team owns column panel c of everything (see item_bwd_body).  Region X is, in turn: the gather's
// scratch, the per-team exchange tiles of phase 1 (kernel tile X0 and whitened tile KwT: 2 NT register images of 256
// doubles per team), then Gt / At of phase 2.  NT = 4, one team: 9 800 doubles = 78 KB, two blocks per CU.
struct ItemBwdLayout {
  size_t Li, Vl, X, El, tql, gml, gvl, ul, al, zl, gt, gs, gq, gzq, szl, czl, wpart, tred, redz, redt, thq, p1l, cvd, tpl, total;
};
__host__ __device__ constexpr ItemBwdLayout sv_item_bwd_layout(int NT, int NWB, bool full) {
  const size_t MP = 16 * NT, TM = (size_t)(NT * (NT + 1) / 2) * SV_TSZ;
  const bool po = sv_bwd_po(NT);
  const size_t G = (NWB / NT) > 0 ? (size_t)(NWB / NT) : 1;
  size_t XS = TM;
  if (po && XS < G * NT * 2 * 256) XS = G * NT * 2 * 256;
  if (XS < SV_GATHER_SCR) XS = SV_GATHER_SCR;
  ItemBwdLayout o{};
  size_t p = 0;
  o.Li = p; p += TM;               // Linv
  o.Vl = p; p += TM;               // Vt; later the tiles of C on and above the diagonal
  o.X = p; p += XS;                // gather scratch; exchange tiles of phase 1; Gt, later At
  o.El = p; p += (!po && full) ? TM : 0;   // whole-tile owners: E (symmetric, tile-packed); panel owners keep it in registers
  const size_t QP = sv_bwd_qc(NT) + (po ? 16 * (G - 1) : 0);   // (panel owners pad a chunk to a whole step of the G teams)
  o.tql = p; p += QP;              // the node chunk in flight: times and the two adjoints, padded to whole tiles
  o.gml = p; p += QP;
  o.gvl = p; p += QP;
  o.ul = p; p += MP;               // ut
  o.al = p; p += MP;               // a = Linv^T ut
  o.zl = p; p += MP;               // Z
  o.gt = p; p += MP;
  o.gs = p; p += MP;
  o.gq = p; p += MP;               // quadrature part of the adjoint of ut
  o.gzq = p; p += MP;              // quadrature part of dZ (direct)
  o.szl = p; p += MP;              // sin of the inducing-point phases (periodic)
  o.czl = p; p += MP;              // cos
  o.wpart = p; p += (po ? G : (size_t)NWB) * 2 * MP;    // per-team (whole-tile owners: per-wave) partial sums of gu / dz
  o.tred = p; p += (size_t)NWB * 2;
  o.redz = p; p += (size_t)NT * MP;   // per-panel row sums of the dZ contraction
  o.redt = p; p += 2 * (size_t)NT;    // per-panel dtheta partials
  o.thq = p; p += 8;               // quadrature part of dtheta (direct), 2 used
  o.p1l = p; p += MP;              // spike partial sums P1 of the trial
  o.cvd = p; p += MP;              // 1 / diag(L_S)
  o.tpl = p; p += 8;               // spike-direct theta partials of the trial, 2 used
  o.total = p;
  return o;
}
__host__ __device__ constexpr size_t sv_item_bwd_lds_doubles(int NT, int NWB, bool full = true) {
  return sv_item_bwd_layout(NT, NWB, full).total;
}

// ---- phase 3 -------------------------------------------------------------------------------------------------------
// dC / dd: group g of SV_FIN_OUT consecutive outputs; thread (o, pg) sums the trials pg, pg + NG, ... and the NG partial
// sums are added in order through LDS.  red needs blockDim.x doubles.
__device__ __forceinline__ void fin_cd_group(const EngineView& ev, const double* C, unsigned mask, double* shared, int g,
                                             double* red) {
  const int tid = threadIdx.x, o = tid % SV_FIN_OUT, pg = tid / SV_FIN_OUT, NG = blockDim.x / SV_FIN_OUT;
  const int NK = ev.N * ev.K, N = ev.N, K = ev.K, R = ev.R;
  const int nbC = (NK + SV_FIN_OUT - 1) / SV_FIN_OUT;
  double s = 0.0;
  int out = -1;
  if (g < nbC) {
    const int id = g * SV_FIN_OUT + o;
    if (id < NK) {
      out = 3 + id;
      if (mask & 2u) {
        const int n = id / K, k = id % K;
        const double cnk = C[id];
        // four trials per step: sixteen independent loads in flight per thread (one trial per step left this block
        // waiting on memory ~30 times in a row), partial sums added in a fixed order
        for (int r0 = pg; r0 < R; r0 += 4 * NG) {
          double v[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int r = r0 + u * NG;
            double t = 0.0;
            if (r < R) {
              const long long o2 = ((long long)r * N + n) * 2 * K + k;      // ncd == R: one partial row per trial
              t = sv_dc_spike(ev, r, n, k) - (ev.gCpart[o2] + cnk * ev.gCpart[o2 + K]);     // dC = spikes - (E^T mu + C . E^T var)
            }
            v[u] = t;
          }
          s += (v[0] + v[1]) + (v[2] + v[3]);
        }
        if (pg == 0) s += split_sums(ev, n, k);
      }
    }
  } else {
    const int n = (g - nbC) * SV_FIN_OUT + o;
    if (n < N) {
      out = 3 + NK + n;
      if (mask & 2u) {
        for (int b0 = pg; b0 < ev.ncd; b0 += 4 * NG) {
          double v[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int b = b0 + u * NG;
            v[u] = (b < ev.ncd) ? ev.gdpart[(long long)b * N + n] : 0.0;
          }
          s += (v[0] + v[1]) + (v[2] + v[3]);
        }
        if (pg == 0) s += ev.spk_cnt[n];
      }
    }
  }
  __syncthreads();
  red[pg * SV_FIN_OUT + o] = s;
  __syncthreads();
  if (pg == 0 && out >= 0) {
    double t = red[o];
    for (int q = 1; q < NG; ++q) t += red[q * SV_FIN_OUT + o];
    shared[out] = t;
  }
}

// Scalars that only need EARLIER kernels' partials (KL per item from k_prep, the quadrature ELL per block, sum_n count_n d_n):
// summed by one designated block of k_item_bwd and left in fin_part[0..1] for k_fin_items (a later launch).
__device__ __forceinline__ void fin_early_scalars(const EngineView& ev, const double* d, double* red) {
  const int tid = threadIdx.x, NTH = blockDim.x, RK = ev.R * ev.K;
  double e = 0.0, kl = 0.0;
  for (int i0 = 0; i0 < RK; i0 += 8 * NTH) {       // 8 independent loads in flight per thread, fixed order
    double v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = i0 + u * NTH + tid;
      v[u] = (i < RK) ? ev.klpart[i] : 0.0;
    }
    kl += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
  }
  for (int i = tid; i < ev.nell; i += NTH) e += ev.ellq[i];
  if (d)
    for (int n = tid; n < ev.N; n += NTH) e += ev.spk_cnt[n] * d[n];
  e = block_sum(e, red);
  kl = block_sum(kl, red);
  if (tid == 0) {
    ev.fin_part[0] = e;
    ev.fin_part[1] = kl;
  }
}

// Phase 3 inside k_item_bwd: only sums whose inputs are complete when the kernel starts.
__device__ __forceinline__ void fin_tail(const EngineView& ev, const double* C, const double* d, unsigned mask,
                                         double* shared, double* red) {
  const int NK = ev.N * ev.K;
  const int nbC = (NK + SV_FIN_OUT - 1) / SV_FIN_OUT, nbD = (ev.N + SV_FIN_OUT - 1) / SV_FIN_OUT;
  const int nblk = gridDim.x;
  for (int g = blockIdx.x; g < nbC + nbD; g += nblk) fin_cd_group(ev, C, mask, shared, g, red);
  if ((int)blockIdx.x == (nbC + nbD) % nblk) fin_early_scalars(ev, d, red);
}

// ---- k_fin_items: the sums that need k_item_bwd's own per-item results -- dtheta (sum over trials of the per-item values)
// and the spike part of the ELL -- then [ELBO, ELL, KL].  A grid of its own, a few blocks: doing this behind a ticket at
// the end of k_item_bwd made every one of its blocks wait for its last stores to land plus an atomic round trip
// (+27..33 us at the north-star size, measured with stop-after-phase builds), a separate launch costs ~6 us.
//   blocks [0, nth_total)      : one kernel parameter each
//   blocks [nth_total, +NS)    : 1 / NS of the items' spike ELL each; the last of them to finish (ticket over NS blocks,
//                                sc1 hand-off as in MI355X_MICROARCH.md) adds the NS partials in order and writes the scalars
#define SV_FIN_NS 8
__global__ __launch_bounds__(256) void k_fin_items(EngineView ev, unsigned mask, double* shared) {
  __shared__ double red[4];
  const int tid = threadIdx.x, K = ev.K, R = ev.R, RK = R * K, NK = ev.N * K, nth = ev.nth_total;
  const int b = blockIdx.x;
  if (b < nth) {
    double s = 0.0;
    if (mask & 4u) {
      int k = 0;
      while (k + 1 < K && ev.meta[k + 1].thoff <= b) ++k;
      const int t = b - ev.meta[k].thoff;
      for (int r = tid; r < R; r += 256) s += ev.gthitem[((long long)r * K + k) * 2 + t];
    }
    s = block_sum(s, red);
    if (tid == 0) shared[3 + NK + ev.N + b] = s;
    return;
  }
  const int part = b - nth;
  const int i0 = (int)((long long)RK * part / SV_FIN_NS), i1 = (int)((long long)RK * (part + 1) / SV_FIN_NS);
  double e = 0.0;
  for (int i = i0 + tid; i < i1; i += 256) e += ev.ells[i];
  e = block_sum(e, red);
  if (tid == 0) {
    __hip_atomic_store(&ev.fin_part[2 + part], e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned t = __hip_atomic_fetch_add(ev.fin_ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int last = (t == SV_FIN_NS - 1);
    if (last) {
      __hip_atomic_store(ev.fin_ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // for the next launch
      double ell = ev.fin_part[0];                                                          // from k_item_bwd (earlier launch)
      for (int q = 0; q < SV_FIN_NS; ++q)
        ell += __hip_atomic_load(&ev.fin_part[2 + q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const double kl = ev.fin_part[1];
      shared[0] = ell - kl;
      shared[1] = ell;
      shared[2] = kl;
    }
  }
}

// Inputs of phase 2 that are complete when the kernel starts: the spike sums of the item -> gs = s2 P0, p1l = P1, the
// spike-direct theta partials -> tpl; 1 / diag(L_S) -> cvd.
//   * items of the spike pass's class path (ev.ser): the class's trigonometric moments A_n, B_n (At_n, Bt_n) = the trial's
//     tile partials added in tile order, combined per inducing point j (phase phi_j, harmonics by rotation), with
//     Cc = cos(n phi_j) A_n + sin(n phi_j) B_n and Ss = cos(n phi_j) B_n - sin(n phi_j) A_n (kernels_spike.h header):
//       P0_j = sum_n c_n Cc,  P1_j = sum_n e_n Ss,  T0 = sum_j a_j sum_n g_n Cc,
//       T1 (periodic) = sum_j a_j sum_n e_n [cos(n phi_j) Bt_n - sin(n phi_j) At_n - z_j Ss]
//     (round 3's k_spike_comb, folded in here: one launch and one dependency less per evaluation; the moments and the
//     coefficient table live in the transpose scratch `scr`, which phase 1 only takes over behind the next barrier);
//   * pair-by-pair items: the chunks' partial sums of the trial in chunk order (thread j < M owns inducing point j).
// Statistics mode: the spike term arrives through the segment-1 adjoints, there is nothing to gather.
#ifdef SV_GATHER_NOINLINE     // diagnosis builds only (tests/tools/gather_early_diag.py)
#define SV_GATHER_ATTR __attribute__((noinline))
#else
#define SV_GATHER_ATTR __forceinline__
#endif
// The gather's first loads, requested at the top of the kernel IN FRONT of the factor staging's loads (every address is
// valid whatever path the item takes): behind the staging, where they used to be issued, the item's path flag, then the
// sums it points to, then the division were three memory latencies in a row (6 us of a 64-us block at the north-star size).
struct GatherPre {
  int ser, c0, c1;
  double p0t, p1t, tt0, tt1, cd;
};
__device__ __forceinline__ GatherPre gather_issue(const EngineView& ev, const LatentMeta& m, const LatentPtrs& pt, int r,
                                                  int k, unsigned mask, bool want_zt) {
  const int tid = threadIdx.x, M = m.M;
  GatherPre g;
  g.ser = ev.ser[r * ev.K + k];
  g.c0 = ev.chunk_first[r];
  g.c1 = ev.chunk_first[r + 1];
  const long long o = (long long)r * ev.sumM + m.colM + (tid < M ? tid : 0);
  g.p0t = ev.P0t[o];
  g.p1t = want_zt ? ev.P1t[o] : 0.0;
  g.tt0 = ev.Tt[((long long)r * ev.K + k) * 2 + 0];
  g.tt1 = ev.Tt[((long long)r * ev.K + k) * 2 + 1];
  g.cd = ((mask & 1u) && tid < M) ? pt.cholvec[(long long)r * m.P + tid * (tid + 1) / 2 + tid] : 1.0;
  return g;
}

template <int MP>
__device__ SV_GATHER_ATTR void gather_item_inputs(const EngineView& ev, const LatentMeta& m, const LatentPtrs& pt, int r,
                                                   int k, unsigned mask, bool want_zt, const KPar& kp, const GatherPre& pre,
                                                   double* gs, double* p1l, double* cvd, double* gt, double* tpl,
                                                   double* scr) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, NTHB = blockDim.x, NWB = NTHB >> 6, M = m.M;
  const double s2 = kp.s2;
  const bool stats = (mask & SV_MASK_STATS) != 0;
  const bool cls = !stats && pre.ser != 0;   // block-uniform
  const int c0 = pre.c0, c1 = (stats || cls) ? c0 : pre.c1;
  double p0 = 0.0, p1sum = 0.0, t0acc = 0.0, t1acc = 0.0;
  if (cls && !ev.comb_fold) {                 // k_spike_comb left the item's sums per trial
    if (tid < M) {
      p0 = pre.p0t;
      p1sum = pre.p1t;
    }
    if (wave == 0 && lane == 0 && want_zt) {
      t0acc = pre.tt0;                        // lane 0 carries the whole sum: the wave tree below adds zeros
      t1acc = pre.tt1;
    }
  } else if (cls) {
    const SpikeItem it = ev.sitem[r * ev.K + k];
    const int H = it.H;
    const bool TW = (mask & 4u) != 0 && m.ktype == SV_PERIODIC;       // what the class kernels produced for this mask (time-weighted moments: kernel-parameter gradient only)
    const int NV = TW ? 4 : 2, ncol = NV * (H + 1);
    double* Fs = scr;                         // SV_ACOLS moments
    double* cf = scr + SV_ACOLS;              // 3 x (SV_FH + 1) coefficients c_n | g_n | e_n
    const int tl0 = ev.tile_first[r], tl1 = ev.tile_first[r + 1];
    double zj = it.t0;
    if (tid < M) zj = ev.Zs[m.offM + (long long)r * M + tid];
    for (int c = tid; c < ncol; c += NTHB) {
      double s = 0.0;
      for (int t = tl0; t < tl1; t += 8) {            // eight tiles' partials in flight, added in tile order
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = ev.Apart[((long long)(t + u < tl1 ? t + u : t) * ev.K + k) * SV_ACOLS + c];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += (t + u < tl1) ? v[u] : 0.0;
      }
      Fs[c] = s;
    }
    if (wave == 0) cls_coefficients(ev, k, m.ktype, kp.h, it.sc, H, cf);
    __syncthreads();
    if (tid < M) {
      const double aj = ev.a[m.offM + (long long)r * M + tid];
      double sphi, cphi;
      sv_sincos_quarter((zj - it.t0) * it.sc, sphi, cphi);
      const double* cN = cf;
      const double* gN = cf + (SV_FH + 1);
      const double* eN = cf + 2 * (SV_FH + 1);
      double cj = 1.0, sj = 0.0, P0 = 0.0, P1 = 0.0, d0 = 0.0, d1 = 0.0;
      for (int n = 0; n <= H; ++n) {
        const double An = Fs[n * NV + 0], Bn = Fs[n * NV + 1];
        const double Cc = fma(cj, An, sj * Bn), Ss = fma(cj, Bn, -(sj * An));
        P0 = fma(cN[n], Cc, P0);
        if (want_zt) {
          const double en = eN[n];
          P1 = fma(en, Ss, P1);
          d0 = fma(gN[n], Cc, d0);
          if (TW) d1 = fma(en, fma(cj, Fs[n * NV + 3], -(sj * Fs[n * NV + 2])) - zj * Ss, d1);
        }
        const double tq = cj * cphi - sj * sphi;
        sj = fma(sj, cphi, cj * sphi);
        cj = tq;
      }
      p0 = P0;
      p1sum = P1;
      t0acc = aj * d0;
      t1acc = aj * d1;
    }
  }
  if (tid < MP) {
    double dinv = 0.0;
    if (tid < M) {
      constexpr int GB = 8;      // chunks per step: 16 independent loads in flight
      for (int ch = c0; ch < c1; ch += GB) {
        double v[GB], w[GB];
#pragma unroll
        for (int u = 0; u < GB; ++u) {
          const bool in = ch + u < c1;
          const long long o = (long long)(in ? ch + u : ch) * ev.sumM + m.colM + tid;
          v[u] = ev.P0[o];
          w[u] = want_zt ? ev.P1[o] : 0.0;
          if (!in) { v[u] = 0.0; w[u] = 0.0; }
        }
        p0 += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
        p1sum += ((w[0] + w[1]) + (w[2] + w[3])) + ((w[4] + w[5]) + (w[6] + w[7]));
      }
      if (mask & 1u) dinv = 1.0 / pre.cd;
    }
    gs[tid] = s2 * p0;
    p1l[tid] = p1sum;
    cvd[tid] = dinv;
    gt[tid] = 0.0;
  }
  if (cls) {
    if (wave == 0) {             // spike-direct theta sums of a class-path item: over its inducing points (lanes < M)
      const double ts0 = wave_sum(t0acc), ts1 = wave_sum(t1acc);
      if (lane == 0) {
        tpl[0] = want_zt ? ts0 : 0.0;
        tpl[1] = want_zt ? ts1 : 0.0;
      }
    }
  } else if (wave == NWB - 1) {   // pair-by-pair item: one chunk per lane, then a fixed-order wave tree
    double tp0 = 0.0, tp1 = 0.0;
    if (want_zt) {
      for (int ch = c0 + lane; ch < c1; ch += 64) {
        tp0 += ev.Tpart[((long long)ch * ev.K + k) * 2 + 0];
        tp1 += ev.Tpart[((long long)ch * ev.K + k) * 2 + 1];
      }
      tp0 = wave_sum(tp0);
      tp1 = wave_sum(tp1);
    }
    if (lane == 0) {
      tpl[0] = tp0;
      tpl[1] = tp1;
    }
  }
}

// ---- phases 1 + 2 --------------------------------------------------------------------------------------------------
// Round 5: PANEL OWNERS.  A block is G teams of NT waves; wave c of a team owns the 16 inducing points of panel c in every
// node tile the team processes (team g takes tiles g, g + G, ...):
//   (1) it evaluates the raw kernel tile X0_c = kappa(nodes, z_{16 c ..}) -- every (node, inducing point) pair is evaluated
//       once per item, whatever the number of waves -- and hands its register image to the team through LDS;
//   (2) with all X0_b of the tile: Gk_c = E[c, :] kappa (FULL; its panel of the symmetric E = Linv^T (Vt Vt^T - I) Linv stays
//       in REGISTERS for the whole item: the accumulator layout of E[:, c] is the A-operand layout of E[c, :]) and the
//       whitened tile KwT_c = (Linv kappa)_c^T (row block c of Linv: c + 1 tile products), handed over likewise;
//   (3) column c of Gt += KwT^T diag(gv) KwT: tiles (a, c), a >= c (NT - c tile products) -- every tile of Gt has ONE owner,
//       complete at the end of the loop -- and the adjoint sums of its own inducing points.
// Per node tile a wave issues NT + (c + 1) + (NT - c) = 2 NT + 1 tile products whatever its panel: the team is balanced by
// construction (round 4 dealt whole node tiles over the waves: 13 tiles on 2 or 4 waves, every wave carrying all
// NT (NT + 1) / 2 accumulator tiles of Gt -- 80 registers at M = 64, which is what kept k_item_bwd<4> at one wave per SIMD).
// Two block barriers per node tile order the two hand-overs; no exchange tile is double-buffered (a tile's readers finish
// in front of the barrier its next writer waits behind).
// Linv, Vt (tile-packed), ut, Z of the item into LDS.  M <= 32: both factors and the two vectors are requested before
// anything is stored (one exposed memory latency); above: LDS-DMA.  The caller synchronises.
template <int NT>
__device__ __forceinline__ void load_factors_tp(const EngineView& ev, const LatentMeta& m, int r, double* Li, double* Vl,
                                                double* ul, double* zl) {
  constexpr int MP = 16 * NT, NE = NT * (NT + 1) / 2 * 256;
  const int M = m.M, tid = threadIdx.x, nth = (int)blockDim.x;
  const long long base = m.offTT + (long long)r * m.PT;      // Linv, Vt: packed lower triangles
  const long long vbase = m.offM + (long long)r * M;
  const double* sl = ev.Linv + base;
  const double* sv = ev.Vt + base;
  const double uv = (tid < M) ? ev.ut[vbase + tid] : 0.0;
  const double zv = (tid < M) ? ev.Zs[vbase + tid] : 0.0;
  if (NT <= 2) {
    constexpr int U = (NT == 1) ? 2 : 6;
    for (int e0 = tid; e0 < NE; e0 += U * nth) {
      double va[U], vb[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int e = e0 + u * nth, t = e >> 8, it = (e >> 4) & 15, jt = e & 15;
        int a, b;
        sv_tile_ab(t, a, b);
        const int i = 16 * a + it, j = 16 * b + jt;
        const bool ok = e < NE && i < M && j <= i;
        const int o = ok ? sv_tri_off(i) + j : 0;
        const double x = sl[o], y = sv[o];
        va[u] = ok ? x : 0.0;
        vb[u] = ok ? y : 0.0;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int e = e0 + u * nth, t = e >> 8, it = (e >> 4) & 15, jt = e & 15;
        if (e < NE) {
          Li[t * SV_TSZ + it * SV_TS + jt] = va[u];
          Vl[t * SV_TSZ + it * SV_TS + jt] = vb[u];
        }
      }
    }
  } else {
    stage_matrix_dma_tp<NT>(Li, sl, M, tid, nth);
    stage_matrix_dma_tp<NT>(Vl, sv, M, tid, nth);
  }
  if (tid < MP) {
    ul[tid] = uv;
    zl[tid] = zv;
  }
  if (NT >= 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the DMA pieces have landed (a barrier follows in the caller)
}

// Nodes [n0, n0 + cnt) of a node list -> LDS, padded to a multiple of `quant` nodes: a padded node repeats the chunk's first
// time with zero adjoints (it adds exact zeros to every sum).
// p2 != 0: the adjoints arrive in two planes p2 doubles apart (the fused ELL kernel's neuron groups: kernels_mfma.h
// k_quad_ell_fused), added here in plane order.
__device__ __forceinline__ void stage_nodes(double* tql, double* gml, double* gvl, const double* tqr, const double* gmq,
                                            const double* gvq, int n0, int cnt, int quant, int tid, int nth,
                                            const long long p2 = 0) {
  const int np = (cnt + quant - 1) / quant * quant;
  for (int i = tid; i < np; i += nth) {
    const bool ok = i < cnt;
    const int src = n0 + (ok ? i : 0);
    const double t = tqr[src];
    double x = gmq[src], y = gvq[src];
    if (p2) {                                          // (block-uniform)
      const double x2 = gmq[src + p2], y2 = gvq[src + p2];
      x += x2;
      y += y2;
    }
    tql[i] = t;
    gml[i] = ok ? x : 0.0;
    gvl[i] = ok ? y : 0.0;
  }
}

template <int NT, int KT, bool FULL>
__device__ __forceinline__ void item_bwd_body(const EngineView& ev, const int r, const int k, const LatentPtrs pt, unsigned mask,
                              double* sm) {
  constexpr int MP = 16 * NT, TS = SV_TS, TSZ = SV_TSZ;
  constexpr bool PO = sv_bwd_po(NT);
  constexpr int NTRI = NT * (NT + 1) / 2;
  const int tid = threadIdx.x;
  const int lane = tid & 63, li = lane & 15, lg = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int NTHB = blockDim.x, NWB = NTHB >> 6;   // NT G waves per block (host: G = 1, 2 or 4 teams)
  const int G = NWB / NT;
  const int team = wave / NT, c = wave - team * NT;        // wave-uniform (SGPRs): team and panel of this wave
  const int orm = li * TS + lg, otr = lg * TS + li;        // row-major / transposed fragment offsets inside a tile
  const int item = r * ev.K + k;
  const LatentMeta m = ev.meta[k];
  const int M = m.M, Q = ev.Q, K = ev.K;
  const ItemBwdLayout lay = sv_item_bwd_layout(NT, NWB, FULL);   // buffer meanings: see the layout function
  double* Li = sm + lay.Li;
  double* Vl = sm + lay.Vl;
  double* X = sm + lay.X;
  double* Gl = X;                     // (symmetric, tile-packed: phase 2's Gt, later At)
  double* El = sm + lay.El;
  double* tql = sm + lay.tql;
  double* gml = sm + lay.gml;
  double* gvl = sm + lay.gvl;
  constexpr int QC = sv_bwd_qc(NT);
  const int nquant = 16 * (PO ? G : 1);     // a chunk is padded to whole tiles (panel owners: to whole steps of the G teams)
  // Value-only variant with the whitened tiles of the quadrature nodes cached by k_front (EngineView::Kwc, the E-step closure):
  // the tile loop loads its whitened tile -- KwT[a][q] of lane (li, lg) = Ktil[inducing 16 a + li][node lg + 4 q], the second
  // image of the cache, in this kernel's register layout -- instead of evaluating the kernel and whitening it
  const bool kwl = !FULL && ev.kw_use != 0;
  const double* kwi = ev.Kwc + ev.kw_half + (long long)(r * ev.K + k) * ((ev.Q + 15) >> 4) * NT * 256 + lane;
  double* ul = sm + lay.ul;
  double* al = sm + lay.al;
  double* zl = sm + lay.zl;
  double* gt = sm + lay.gt;
  double* gs = sm + lay.gs;
  double* gq = sm + lay.gq;
  double* gzq = sm + lay.gzq;
  double* szl = sm + lay.szl;
  double* czl = sm + lay.czl;
  double* wpart = sm + lay.wpart;
  double* tred = sm + lay.tred;
  double* redz = sm + lay.redz;
  double* redt = sm + lay.redt;
  double* thq = sm + lay.thq;
  double* p1l = sm + lay.p1l;
  double* cvd = sm + lay.cvd;
  double* tpl = sm + lay.tpl;
  const KPar p = make_kpar(m, ev.theta + 2 * k);
  const long long vbase = m.offM + (long long)r * M;
  const bool want_lat = (mask & (1u | 4u | 8u)) != 0;
  const bool want_zt = (mask & (4u | 8u)) != 0;
  long long ck[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  SV_CK(0);
  const GatherPre gpre = gather_issue(ev, m, pt, r, k, mask, want_zt);

  if (want_lat) {
    load_factors_tp<NT>(ev, m, r, Li, Vl, ul, zl);
    // the first chunk of the quadrature nodes, requested with the factors (the barrier in front of phase 1 publishes it)
    stage_nodes(tql, gml, gvl, ev.tq + (long long)r * Q, ev.gmuq + ((long long)r * K + k) * Q,
                ev.gvarq + ((long long)r * K + k) * Q, 0, Q < QC ? Q : QC, nquant, tid, NTHB, ev.glat2);
  } else if (tid < MP) {
    ul[tid] = 0.0;
    zl[tid] = 0.0;
  }
  if (tid < MP) al[tid] = (tid < M) ? ev.a[vbase + tid] : 0.0;
  // The gather of phase 2's inputs (for class-path items: the combine of the spike moments) runs HERE, beside the
  // staging of the factors: between the phases, where the values are used, its memory latency was exposed (12 us of the
  // kernel at the north-star size).
  SV_CK(1);
  gather_item_inputs<MP>(ev, m, pt, r, k, mask, want_zt, p, gpre, gs, p1l, cvd, gt, tpl, X);
  SV_CK(2);
  if (SV_BWD_STOP == 9) return;      // after the staging of the factors and the gather

  if (want_lat) {
    // ================================================================= phase 1: quadrature adjoints
    __syncthreads();
    if (KT == SV_PERIODIC && tid < MP) sv_sincos_quarter(zl[tid] * p.q4, szl[tid], czl[tid]);
    // The adjoint of kappa(t, Z) at a node is gk = Linv^T [gm ut + 2 gv (Vt Vt^T - I) kt], kt = Linv kappa(t, Z), i.e.
    //     gk = gm a + 2 gv E kappa(t, Z)      with the per-item SYMMETRIC matrix  E = Linv^T (Vt Vt^T - I) Linv
    // one dense M x M x 16 product per tile on the RAW kernel values, independent of the whitening chain of the same tile.
    // Rounding: E kappa and D kt (D = Linv^T (Vt Vt^T - I), round 3) both carry an absolute error of order u cond(Kzz); the
    // MATLAB-fixture parity test (eps = 1e-5, cond 1e6) bounds it either way.  Every wave builds the panel it owns (the
    // waves of further teams repeat their team-0 twin's chain: no wave-uniform branch around this register-heavy section,
    // DESIGN.md section 9) and keeps it: Ep[b][q] = E[16 b + 4 q + lg][16 c + li] = E[16 c + li][16 b + 4 q + lg].
    d4 Ep[NT];
    if (FULL) {
      d4 P[NT], T1[NT], T2[NT];
#pragma unroll
      for (int b = 0; b < NT; ++b)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const double v = Li[((b >= c) ? (b * (b + 1) / 2 + c) : 0) * TSZ + otr + 4 * q * TS];      // Linv[:, panel c]
          P[b][q] = (b >= c) ? v : 0.0;
        }
      tri_chain_tp<NT, 1>(T1, Vl, P, orm, otr, c);    // Vt^T Linv[:, c]  (the panel's tiles above the diagonal are zero: skipped)
      tri_chain_tp<NT, 0>(T2, Vl, T1, orm, otr);      // Vt Vt^T Linv[:, c]
#pragma unroll
      for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int q = 0; q < 4; ++q) T2[a][q] -= P[a][q];
      tri_chain_tp<NT, 1>(Ep, Li, T2, orm, otr);      // Linv^T (Vt Vt^T - I) Linv[:, c]
      if (!PO) {                                      // whole-tile owners read E from LDS: the panel's tiles on and below the diagonal
#pragma unroll
        for (int a = 0; a < NT; ++a) {
          if (a >= c) {
#pragma unroll
            for (int q = 0; q < 4; ++q) El[(a * (a + 1) / 2 + c) * TSZ + otr + 4 * q * TS] = Ep[a][q];
          }
        }
      }
    }
    __syncthreads();                   // (the gather's scratch in region X is dead: the exchange tiles take its place)
    SV_CK(3);
    if (SV_BWD_STOP == 10) return;   // after the E panels
    const double inv_s2 = 1.0 / p.s2;
    const int nseg = (mask & SV_MASK_STATS) ? 2 : 1;
    if constexpr (!PO) {
    // ------------------------------------------------------------ whole-tile owners (M <= 32): wave w takes node tiles
    // w, w + NWB, ...; all NT (NT + 1) / 2 accumulator tiles of Gt per wave, added over the waves behind the loop
    d4 dz[NT], GW[NTRI];
    double gu[NT];                      // lane (li, lg): partial of gu_{16 a + li} over the nodes lg + 4 q of every tile
#pragma unroll
    for (int a = 0; a < NT; ++a) {
      gu[a] = 0.0;
      dz[a] = (d4){0.0, 0.0, 0.0, 0.0};
    }
#pragma unroll
    for (int t = 0; t < NTRI; ++t) GW[t] = (d4){0.0, 0.0, 0.0, 0.0};
    double th0 = 0.0, th1 = 0.0;
    double* Sw = wpart + wave * 2 * MP;    // this wave's partial sums of gu / dz (written behind its last tile)
    for (int seg = 0; seg < nseg; ++seg) {
      const long long s0 = seg ? ev.spk_off[r] : 0;
      const int Qs = seg ? (int)(ev.spk_off[r + 1] - s0) : Q;
      const double* gmq = seg ? ev.gmuspk + (long long)k * ev.S + s0 : ev.gmuq + ((long long)r * K + k) * Q;
      const double* gvq = seg ? ev.gvarspk + (long long)k * ev.S + s0 : ev.gvarq + ((long long)r * K + k) * Q;
      const double* tqr = seg ? ev.spk_t + s0 : ev.tq + (long long)r * Q;
      for (int n0 = 0; n0 < Qs; n0 += QC) {      // chunks of the node list (block-uniform)
      const int cnt = (Qs - n0 < QC) ? Qs - n0 : QC;
      if (seg | n0) {                            // (the first chunk of the quadrature nodes was staged at the top)
        __syncthreads();
        stage_nodes(tql, gml, gvl, tqr, gmq, gvq, n0, cnt, nquant, tid, NTHB, seg ? 0 : ev.glat2);
        __syncthreads();
      }
      const int ntiles = (cnt + 15) >> 4;
      for (int tile = wave; tile < ntiles; tile += NWB) {
        const double tq = tql[16 * tile + li], gm = gml[16 * tile + li], gvv = gvl[16 * tile + li];
        double stq = 0.0, ctq = 0.0;
        int jo = lg;
        asm volatile("" : "+v"(jo));      // (opaque: keeps the per-j constants out of long-lived registers)
        d4 X0[NT];
        d4 csv[(KT == SV_PERIODIC && FULL) ? NT : 1];
        if (!kwl) {
        if (KT == SV_PERIODIC) SV_BWD_SINCOS(tq * p.q4, stq, ctq);
#pragma unroll
        for (int a = 0; a < NT; ++a)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int j = 16 * a + 4 * q + jo;
            if (KT == SV_RBF) {
              const double dd = tq - zl[j];
              X0[a][q] = p.s2 * SV_BWD_EXP2(-(dd * dd) * p.h2);
            } else {
              const double cs = fma(stq, szl[j], ctq * czl[j]);
              if (FULL) csv[a][q] = cs;
              X0[a][q] = p.s2 * SV_BWD_EXP2(fma(cs, p.h2, -p.h2));
            }
          }
        }
        // adjoints of the four nodes lg + 4 q of this tile
        double gm4[4], gv4[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          gm4[q] = gml[16 * tile + lg + 4 * q];
          gv4[q] = gvl[16 * tile + lg + 4 * q];
        }
        // Whitened tile TRANSPOSED: KwT[t][i] = sum_j kappa(t, z_j) Linv[i][j] -- the raw kernel tile is the A operand as it
        // stands, Linv's fragments the B operand; the accumulator holds node lg + 4 q of inducing point 16 a + li, which is
        // BOTH operand layouts of the contraction over the nodes Gt += KwT^T diag(gv) KwT
        d4 KwT[NT];
        if (kwl) {
          const double* kt = kwi + (long long)((n0 >> 4) + tile) * NT * 256;
#pragma unroll
          for (int a = 0; a < NT; ++a)
#pragma unroll
            for (int q = 0; q < 4; ++q) KwT[a][q] = kt[a * 256 + 64 * q];
        } else {
#pragma unroll
        for (int a = 0; a < NT; ++a) {
          d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
          for (int b = 0; b <= a; ++b)
#pragma unroll
            for (int q = 0; q < 4; ++q)
              acc = __builtin_amdgcn_mfma_f64_16x16x4f64(X0[b][q], lds_frag(&Li[sv_tidx(a, b) * TSZ + orm + 4 * q]), acc, 0, 0, 0);
          KwT[a] = acc;
        }
        }
        d4 Gk[NT];
        if (FULL) {
#pragma unroll
          for (int a = 0; a < NT; ++a) {
            d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int b = 0; b < NT; ++b)
#pragma unroll
              for (int q = 0; q < 4; ++q)
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(SV_SYMFRAG(El, a, b, q), X0[b][q], acc, 0, 0, 0);
            Gk[a] = acc;                                            // E kappa(t, Z): independent of KwT
          }
        }
        {
          d4 gK[NT];
#pragma unroll
          for (int a = 0; a < NT; ++a)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              gu[a] = fma(gm4[q], KwT[a][q], gu[a]);
              gK[a][q] = gv4[q] * KwT[a][q];
            }
          int t = 0;
#pragma unroll
          for (int a = 0; a < NT; ++a)
#pragma unroll
            for (int b = 0; b <= a; ++b) {
#pragma unroll
              for (int q = 0; q < 4; ++q) GW[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(KwT[a][q], gK[b][q], GW[t], 0, 0, 0);
              ++t;
            }
        }
        if (FULL) {
#pragma unroll
          for (int a = 0; a < NT; ++a)
#pragma unroll
            for (int q = 0; q < 4; ++q) Gk[a][q] = fma(2.0 * gvv, Gk[a][q], gm * al[16 * a + 4 * q + jo]);
          asm volatile("" : "+v"(jo));
#pragma unroll
          for (int a = 0; a < NT; ++a)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const int j = 16 * a + 4 * q + jo;
              const double gkk = Gk[a][q] * X0[a][q];               // gk * kappa: the factor 1 / s2 (kappa -> ktil) goes onto the sums
              const double dd = tq - zl[j];
              if (KT == SV_RBF) {
                const double gd = gkk * dd;
                dz[a][q] += gd;                                    // bz = delta
                th0 = fma(gd, dd, th0);                            // b0 = delta^2
              } else {
                const double cs = csv[a][q];
                const double sn = fma(stq, czl[j], -(ctq * szl[j]));
                const double gsn = gkk * sn;
                dz[a][q] += gsn;                                   // bz = sin(2 pi delta / p)
                th0 = fma(gkk, 0.5 - 0.5 * cs, th0);               // b0 = sin^2(pi delta / p)
                th1 = fma(gsn, dd, th1);                           // b1 = delta sin(2 pi delta / p)
              }
            }
        }
      }
      }   // chunks
    }   // node segments
    SV_CK(9);
    // ---- reductions: gu over the four node groups (lg), dz over the 16 node lanes; then over the wavefronts (fixed order)
#pragma unroll
    for (int a = 0; a < NT; ++a) {
      const double g = grp4_sum(gu[a]);
      if (lg == 0) Sw[16 * a + li] = g;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const double z = FULL ? row16_sum(dz[a][q]) : 0.0;
        if (li == 0) Sw[MP + 16 * a + 4 * q + lg] = z;
      }
    }
    th0 = wave_sum(th0);
    th1 = wave_sum(th1);
    if (lane == 0) {
      tred[wave * 2 + 0] = th0;
      tred[wave * 2 + 1] = th1;
    }
    // Gt: the waves' accumulator tiles added in wave order into region X (the gather's scratch is dead since the barrier
    // behind the E panels); wave 0 writes ahead of the barrier
    auto gt_tiles = [&](const bool first) {
      int t = 0;
#pragma unroll
      for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int b = 0; b <= a; ++b) {
          double* gtile = Gl + sv_tidx(a, b) * TSZ + otr;          // element (16 a + lg + 4 q, 16 b + li)
#pragma unroll
          for (int q = 0; q < 4; ++q) gtile[4 * q * TS] = first ? GW[t][q] : gtile[4 * q * TS] + GW[t][q];
          ++t;
        }
    };
    if (wave == 0) gt_tiles(true);
    __syncthreads();   // all waves done with their tiles; wave 0's Gt tiles and every wave's partials are in LDS
    if (tid < MP) {    // gu / dz: the waves' partials, in order
      double g = 0.0, z = 0.0;
      if (tid < M) {
        g = wpart[tid];
        for (int w = 1; w < NWB; ++w) g += wpart[w * 2 * MP + tid];
        if (FULL) {
          z = wpart[MP + tid];
          for (int w = 1; w < NWB; ++w) z += wpart[w * 2 * MP + MP + tid];
          z *= p.cz * inv_s2;
        }
      }
      gq[tid] = g;
      gzq[tid] = z;
    }
    if (tid == 0) {
      double t0s = 0.0, t1s = 0.0;
      if (FULL) {
        t0s = tred[0];
        t1s = tred[1];
        for (int w = 1; w < NWB; ++w) {
          t0s += tred[2 * w];
          t1s += tred[2 * w + 1];
        }
      }
      thq[0] = p.c0 * inv_s2 * t0s;
      thq[1] = p.c1 * inv_s2 * t1s;
    }
    for (int w = 1; w < NWB; ++w) {
      if (wave == w) gt_tiles(false);
      __syncthreads();
    }
    } else {
    // ------------------------------------------------------------ panel owners (M > 32)
    d4 GW[NT];                          // column c of Gt: tile (a, c), a >= c
    d4 dz = {0.0, 0.0, 0.0, 0.0};       // lane (li, lg): partial of dz_{16 c + 4 q + lg} over the nodes li of every tile
    double gu = 0.0;                    // partial of gu_{16 c + li} over the nodes lg + 4 q of every tile
#pragma unroll
    for (int a = 0; a < NT; ++a) GW[a] = (d4){0.0, 0.0, 0.0, 0.0};
    double th0 = 0.0, th1 = 0.0;
    double* Xb = X + (size_t)team * 2 * NT * 256;     // this team's raw kernel tiles: register image of panel b at [b][q][lane]
    double* Kb = Xb + NT * 256;                       // ... whitened tiles
    const int jb = 16 * c + lg;                       // this lane's inducing points: jb + 4 q
    // Node lists of the item.  Segment 0: the trial's Q quadrature nodes with the adjoints the ELL kernel left in
    // [r][k][q] (contiguous for this item).  Segment 1 (statistics mode only, mask bit SV_MASK_STATS): the trial's spike
    // times with caller-given adjoints of the posterior mean / variance at the spikes, [k][S] -- the general form of the
    // spike term's backward, for likelihoods whose spike adjoint is not known a priori (svgpfa_latent_stats_backward).
    for (int seg = 0; seg < nseg; ++seg) {
      const long long s0 = seg ? ev.spk_off[r] : 0;
      const int Qs = seg ? (int)(ev.spk_off[r + 1] - s0) : Q;
      const double* gmq = seg ? ev.gmuspk + (long long)k * ev.S + s0 : ev.gmuq + ((long long)r * K + k) * Q;
      const double* gvq = seg ? ev.gvarspk + (long long)k * ev.S + s0 : ev.gvarq + ((long long)r * K + k) * Q;
      const double* tqr = seg ? ev.spk_t + s0 : ev.tq + (long long)r * Q;
      for (int n0 = 0; n0 < Qs; n0 += QC) {      // chunks of the node list (block-uniform)
      const int cnt = (Qs - n0 < QC) ? Qs - n0 : QC;
      if (seg | n0) {                            // (the first chunk of the quadrature nodes was staged at the top)
        __syncthreads();
        stage_nodes(tql, gml, gvl, tqr, gmq, gvq, n0, cnt, nquant, tid, NTHB, seg ? 0 : ev.glat2);
        __syncthreads();
      }
      // A team without a tile in the chunk's last step works on padding: nodes with zero adjoints (stage_nodes) that add
      // exact zeros -- no wave-uniform branch around the register-heavy step, and every wave meets every barrier.
      const int nsteps = (cnt + nquant - 1) / nquant;
      // state of the step in flight (stage1 -> stage2 -> stage3)
      double tq = 0.0, gm = 0.0, gvv = 0.0, stq = 0.0, ctq = 0.0;
      d4 X0 = {0.0, 0.0, 0.0, 0.0}, csv = {0.0, 0.0, 0.0, 0.0}, gK = {0.0, 0.0, 0.0, 0.0};
      d4 gm4 = {0.0, 0.0, 0.0, 0.0}, gv4 = {0.0, 0.0, 0.0, 0.0};
      // ---- (1) the raw kernel tile of this wave's panel: lane = node li, registers = inducing points jb + 4 q.
      // periodic: kappa and its derivative bases follow from the node's own sin / cos and the inducing points'
      // (szl, czl) by angle addition (no per-pair sincos, no second exp)
      auto stage1 = [&](const int s) {
        const int tile = s * G + team;
        tq = tql[16 * tile + li];
        gm = gml[16 * tile + li];
        gvv = gvl[16 * tile + li];
        if (KT == SV_PERIODIC) SV_BWD_SINCOS(tq * p.q4, stq, ctq);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int j = jb + 4 * q;
          if (KT == SV_RBF) {
            const double dd = tq - zl[j];
            X0[q] = p.s2 * SV_BWD_EXP2(-(dd * dd) * p.h2);
          } else {
            const double cs = fma(stq, szl[j], ctq * czl[j]);
            csv[q] = cs;
            X0[q] = p.s2 * SV_BWD_EXP2(fma(cs, p.h2, -p.h2));
          }
        }
        double* xw = Xb + c * 256 + lane;
#pragma unroll
        for (int q = 0; q < 4; ++q) xw[q * 64] = X0[q];
        // adjoints of the four nodes lg + 4 q of this tile: what this lane's registers of the TRANSPOSED whitened tile
        // belong to
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          gm4[q] = gml[16 * tile + lg + 4 * q];
          gv4[q] = gvl[16 * tile + lg + 4 * q];
        }
      };
      // ---- (2) Gk_c = E[c, :] kappa (A: this wave's E panel, B: every panel's raw tile) and the whitened tile TRANSPOSED,
      // KwT_c[t][i] = sum_j kappa(t, z_j) Linv[16 c + i][j] (A: the raw tiles b <= c as they stand, B: row block c of Linv):
      // the accumulator holds node lg + 4 q of inducing point 16 c + li, which is BOTH operand layouts of the
      // contraction over the nodes Gt += KwT^T diag(gv) KwT -- no transpose anywhere
      auto stage2 = [&](const int s) {
        const double* xr = Xb + lane;
        d4 Gk = {0.0, 0.0, 0.0, 0.0}, KwT = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int b = 0; b < NT; ++b) {
          d4 xb;
#pragma unroll
          for (int q = 0; q < 4; ++q) xb[q] = xr[b * 256 + q * 64];
          if (FULL) {
#pragma unroll
            for (int q = 0; q < 4; ++q) Gk = __builtin_amdgcn_mfma_f64_16x16x4f64(Ep[b][q], xb[q], Gk, 0, 0, 0);
          }
          if (b <= c) {
            const double* lt = Li + (c * (c + 1) / 2 + b) * TSZ + orm;
#pragma unroll
            for (int q = 0; q < 4; ++q) KwT = __builtin_amdgcn_mfma_f64_16x16x4f64(xb[q], lds_frag(lt + 4 * q), KwT, 0, 0, 0);
          }
        }
        double* kw = Kb + c * 256 + lane;
#pragma unroll
        for (int q = 0; q < 4; ++q) kw[q * 64] = KwT[q];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          gu = fma(gm4[q], KwT[q], gu);
          gK[q] = gv4[q] * KwT[q];
        }
        if (FULL) {
          // adjoint sums of this wave's inducing points (the raw tile and E kappa are complete; nothing here waits for
          // the other waves, so it sits in front of the barrier)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int j = jb + 4 * q;
            const double gk = fma(2.0 * gvv, Gk[q], gm * al[j]);
            const double gkk = gk * X0[q];                        // gk * kappa: the factor 1 / s2 (kappa -> ktil) goes onto the sums
            const double dd = tq - zl[j];
            if (KT == SV_RBF) {
              const double gd = gkk * dd;
              dz[q] += gd;                                        // bz = delta
              th0 = fma(gd, dd, th0);                             // b0 = delta^2
            } else {
              const double cs = csv[q];
              const double sn = fma(stq, czl[j], -(ctq * szl[j]));
              const double gsn = gkk * sn;
              dz[q] += gsn;                                       // bz = sin(2 pi delta / p)
              th0 = fma(gkk, 0.5 - 0.5 * cs, th0);                // b0 = sin^2(pi delta / p)
              th1 = fma(gsn, dd, th1);                            // b1 = delta sin(2 pi delta / p)
            }
          }
        }
      };
      // ---- (3) column c of Gt: tiles (a, c), a >= c: A = KwT_a (from its owner), B = diag(gv) KwT_c (own)
      auto stage3 = [&](const int s) {
        const double* kr = Kb + lane;
#pragma unroll
        for (int a = 0; a < NT; ++a) {
          if (a >= c) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
              GW[a] = __builtin_amdgcn_mfma_f64_16x16x4f64(kr[a * 256 + q * 64], gK[q], GW[a], 0, 0, 0);
          }
        }
      };
      if (kwl) {
        // whitened tiles cached by k_front: a wave loads its own panel's tile (512 contiguous bytes per load instruction) and
        // hands it to the team through LDS as in the computing form -- no kernel evaluation, no whitening, and ONE barrier per
        // node tile: the hand-over alternates between the two exchange buffers (the raw tiles' one is idle here)
        for (int s = 0; s < nsteps; ++s) {
          const int tile = s * G + team;
          const double* kt = kwi + (long long)((n0 >> 4) + tile) * NT * 256;
          const bool live = 16 * tile < cnt;             // (a team beyond the chunk's last tile: padded nodes, zero adjoints)
          double* kb = (s & 1) ? Xb : Kb;
          d4 KwT;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            gm4[q] = gml[16 * tile + lg + 4 * q];
            gv4[q] = gvl[16 * tile + lg + 4 * q];
            KwT[q] = live ? kt[c * 256 + 64 * q] : 0.0;
          }
#pragma unroll
          for (int q = 0; q < 4; ++q) kb[c * 256 + q * 64 + lane] = KwT[q];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            gu = fma(gm4[q], KwT[q], gu);
            gK[q] = gv4[q] * KwT[q];
          }
          __syncthreads();
#pragma unroll
          for (int a = 0; a < NT; ++a) {
            if (a >= c) {
#pragma unroll
              for (int q = 0; q < 4; ++q)
                GW[a] = __builtin_amdgcn_mfma_f64_16x16x4f64(kb[a * 256 + q * 64 + lane], gK[q], GW[a], 0, 0, 0);
            }
          }
        }
      } else {
        for (int s = 0; s < nsteps; ++s) {
          stage1(s);
          __syncthreads();
          stage2(s);
          __syncthreads();
          stage3(s);
        }
      }
      }   // chunks
    }   // node segments
    SV_CK(9);
    // ---- reductions: gu over the four node groups (lg), dz over the 16 node lanes; then over the teams (fixed order)
    {
      double* Sw = wpart + team * 2 * MP;
      const double g = grp4_sum(gu);
      if (lg == 0) Sw[16 * c + li] = g;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const double z = FULL ? row16_sum(dz[q]) : 0.0;
        if (li == 0) Sw[MP + jb + 4 * q] = z;
      }
    }
    th0 = wave_sum(th0);
    th1 = wave_sum(th1);
    if (lane == 0) {
      tred[wave * 2 + 0] = th0;
      tred[wave * 2 + 1] = th1;
    }
    __syncthreads();   // every wave is done with the exchange tiles: region X becomes Gt
    // Gt: every tile on or below the diagonal has one owner per team; the teams' tiles are added in team order.  Gt's
    // padding rows / columns are exact zeros: sums of products with zero-padded rows of Linv
    for (int t = 0; t < G; ++t) {
      if (team == t) {
#pragma unroll
        for (int a = 0; a < NT; ++a) {
          if (a >= c) {
            double* gtile = Gl + (a * (a + 1) / 2 + c) * TSZ + otr;       // element (16 a + lg + 4 q, 16 c + li)
#pragma unroll
            for (int q = 0; q < 4; ++q) gtile[4 * q * TS] = (t == 0) ? GW[a][q] : gtile[4 * q * TS] + GW[a][q];
          }
        }
      }
      __syncthreads();
    }
    if (tid < MP) {    // gu / dz: the teams' partials, in order
      double g = 0.0, z = 0.0;
      if (tid < M) {
        g = wpart[tid];
        for (int w = 1; w < G; ++w) g += wpart[w * 2 * MP + tid];
        if (FULL) {
          z = wpart[MP + tid];
          for (int w = 1; w < G; ++w) z += wpart[w * 2 * MP + MP + tid];
          z *= p.cz * inv_s2;
        }
      }
      gq[tid] = g;
      gzq[tid] = z;
    }
    if (tid == 0) {
      double t0s = 0.0, t1s = 0.0;
      if (FULL) {
        t0s = tred[0];
        t1s = tred[1];
        for (int w = 1; w < NWB; ++w) {
          t0s += tred[2 * w];
          t1s += tred[2 * w + 1];
        }
      }
      thq[0] = p.c0 * inv_s2 * t0s;
      thq[1] = p.c1 * inv_s2 * t1s;
    }
    }   // whole-tile / panel owners
  }

  SV_CK(10);
  if (SV_BWD_STOP == 1) return;
  // =================================================================== phase 2: per-item backward
  // ---- spike part of the ELL of this item: sum_j a_j gs_j
  double p1sum = 0.0, espk = 0.0, a_l = 0.0;
  if (tid < M) {
    a_l = al[tid];
    espk = a_l * gs[tid];
    p1sum = p1l[tid];
  }
  if (wave == 0) {
    espk = wave_sum(espk);
    if (lane == 0) ev.ells[item] = espk;
  }
  if (!want_lat) return;
  if (SV_BWD_STOP == 2) return;

  __syncthreads();
  // the two triangular matrix-vector products of the item, LPR lanes per row (one thread per row walked up to M
  // LDS-latency-bound iterations: a microsecond each at M = 32)
  const int LPR = (NTHB >= 16 * MP) ? 16 : (NTHB >= 8 * MP) ? 8 : (NTHB >= 4 * MP) ? 4 : (NTHB >= 2 * MP) ? 2 : 1;
  const int mrow = tid / LPR, msub = tid % LPR;
  {   // gt = gq + Linv gs
    double s = 0.0;
    if (mrow < M)
      for (int q = msub; q <= mrow; q += LPR) s = fma(Li[sv_tp(mrow, q)], gs[q], s);
    for (int o = 1; o < LPR; o <<= 1) s += __shfl_xor(s, o, 64);
    if (msub == 0 && mrow < M) gt[mrow] = gq[mrow] + s;
  }
  __syncthreads();

  const bool pan = team == 0;             // the panel work of phase 2: wave c of the first team (the others idle through it)
  if (mask & 1u) {
    {   // dm = Linv^T (gt - ut)
      double s = 0.0;
      if (mrow < M)
        for (int q = mrow + msub; q < M; q += LPR) s = fma(Li[sv_tp(q, mrow)], gt[q] - ul[q], s);
      for (int o = 1; o < LPR; o <<= 1) s += __shfl_xor(s, o, 64);
      if (msub == 0 && mrow < M) pt.g_mean[(long long)r * M + mrow] = s;
    }
    if (pan) {
      // dL_S = tril(Linv^T (2 Gt - I) Vt) + diag(1 / L_S,ii): this wave's 16-column panel
      double* gcv = pt.g_cholvec + (long long)r * m.P;
      d4 P[NT], Xm[NT];
#pragma unroll
      for (int b = 0; b < NT; ++b)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const double v = Vl[((b >= c) ? (b * (b + 1) / 2 + c) : 0) * TSZ + otr + 4 * q * TS];      // Vt[:, panel c]
          P[b][q] = (b >= c) ? v : 0.0;
        }
#pragma unroll
      for (int a = 0; a < NT; ++a) {
        d4 acc = {0.0, 0.0, 0.0, 0.0};
        if (a >= c) {
#pragma unroll
          for (int b = 0; b < NT; ++b) {
            if (b >= c) {   // Vt lower triangular: panel c has zero tiles above the diagonal
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                const double av = 2.0 * SV_SYMFRAG(Gl, a, b, q) - ((a == b && li == 4 * q + lg) ? 1.0 : 0.0);
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, P[b][q], acc, 0, 0, 0);
              }
            }
          }
        }
        Xm[a] = acc;
      }
#pragma unroll
      for (int a = 0; a < NT; ++a) {
        if (a >= c) {
          d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
          for (int b = a; b < NT; ++b)
#pragma unroll
            for (int q = 0; q < 4; ++q)
              acc = __builtin_amdgcn_mfma_f64_16x16x4f64(lds_frag(&Li[sv_tidx(b, a) * TSZ + otr + 4 * q * TS]), Xm[b][q], acc, 0, 0, 0);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            int i = 16 * a + 4 * q + lg, j = 16 * c + li;
            if (i < M && j <= i) {
              double g = acc[q];
              if (i == j) g += cvd[i];
              gcv[i * (i + 1) / 2 + j] = g;
            }
          }
        }
      }
    }
  }
  if (!want_zt) return;
  if (SV_BWD_STOP == 6) return;      // after dm / dL_S

  // ---- panel c of C = Vt Vt^T Gt and of VV = Vt Vt^T, kept in registers until Vt is dead in every wave
  d4 VV[NT], Cm[NT];
  if (pan) {
    d4 P[NT], Bm[NT];
#pragma unroll
    for (int b = 0; b < NT; ++b)
#pragma unroll
      for (int q = 0; q < 4; ++q)      // Gt[:, panel c] (symmetric: tile (b, c) transposed-read, or tile (c, b) row-major)
        P[b][q] = Gl[(b >= c) ? (b * (b + 1) / 2 + c) * TSZ + otr + 4 * q * TS : (c * (c + 1) / 2 + b) * TSZ + orm + 4 * q];
    tri_chain_tp<NT, 1>(Bm, Vl, P, orm, otr);
    tri_chain_tp<NT, 0>(Cm, Vl, Bm, orm, otr);
#pragma unroll
    for (int b = 0; b < NT; ++b)
#pragma unroll
      for (int q = 0; q < 4; ++q) {      // Vt^T panel: Vt[16 c + li][16 b + 4 q + lg], b <= c
        const double v = Vl[((b <= c) ? (c * (c + 1) / 2 + b) : 0) * TSZ + orm + 4 * q];
        P[b][q] = (b <= c) ? v : 0.0;
      }
    tri_chain_tp<NT, 0>(VV, Vl, P, orm, otr, 0, c);      // (Vt^T[:, c] is zero below tile c: skipped)
  }
  __syncthreads();
  // At needs C + C^T: the tiles of C on and above the diagonal go through LDS (Vt's place, Vt is dead): tile (a, c), a <= c,
  // of this wave's panel row-major into slot (c, a); its reader -- the owner of At's tile (c, a) -- reads it transposed
  double* Cs = Vl;
  if (pan) {
#pragma unroll
    for (int a = 0; a < NT; ++a) {
      if (a <= c) {
#pragma unroll
        for (int q = 0; q < 4; ++q) Cs[(c * (c + 1) / 2 + a) * TSZ + otr + 4 * q * TS] = Cm[a][q];
      }
    }
  }
  __syncthreads();
  if (SV_BWD_STOP == 7) return;      // after C, Vt Vt^T
  // ---- At, tiles (a, c), a >= c, in place over Gt: each element is read and written by exactly one lane
  if (pan) {
#pragma unroll
    for (int a = 0; a < NT; ++a) {
      if (a >= c) {
        const int tb = (a * (a + 1) / 2 + c) * TSZ;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int i = 16 * a + 4 * q + lg, j = 16 * c + li;
          const double ct = Cs[tb + orm + 4 * q];                  // C[j][i]
          const double v = -0.5 * (gt[i] * ul[j] + ul[i] * gt[j]) + 0.5 * ul[i] * ul[j] + Gl[tb + otr + 4 * q * TS] +
                           0.5 * (VV[a][q] - (i == j ? 1.0 : 0.0)) - Cm[a][q] - ct;
          Gl[tb + otr + 4 * q * TS] = v;
        }
      }
    }
  }
  __syncthreads();
  if (SV_BWD_STOP == 8) return;      // after At
  // ---- panel c of A = Linv^T At Linv, contracted on the fly with dKzz/dZ and dKzz/dtheta
  if (pan) {
    d4 P[NT], Dm[NT], Am[NT];
#pragma unroll
    for (int b = 0; b < NT; ++b)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const double v = Li[((b >= c) ? (b * (b + 1) / 2 + c) : 0) * TSZ + otr + 4 * q * TS];
        P[b][q] = (b >= c) ? v : 0.0;
      }
#pragma unroll
    for (int a = 0; a < NT; ++a) {
      d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int b = 0; b < NT; ++b)
        if (b >= c) {   // Linv lower triangular
#pragma unroll
          for (int q = 0; q < 4; ++q)
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(SV_SYMFRAG(Gl, a, b, q), P[b][q], acc, 0, 0, 0);
        }
      Dm[a] = acc;
    }
    tri_chain_tp<NT, 1>(Am, Li, Dm, orm, otr);
    const double zc = zl[16 * c + li];
    const double szc = (KT == SV_PERIODIC) ? szl[16 * c + li] : 0.0, czc = (KT == SV_PERIODIC) ? czl[16 * c + li] : 0.0;
    double t0 = 0.0, t1 = 0.0;
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        int i = 16 * a + 4 * q + lg, j = 16 * c + li;
        double gz = 0.0;
        if (i < M && j < M) {
          double kt, bz, b0, b1;
          if (KT == SV_PERIODIC) {      // sin / cos of the inducing points' phases are in LDS since phase 1: angle addition
            const double cs = fma(czl[i], czc, szl[i] * szc), sn = fma(szl[i], czc, -(czl[i] * szc));
            kt = sv_exp2_neg_sc(fma(cs, p.h2, -p.h2));      // (coefficients in SGPRs: in these conditional blocks the compiler
            bz = sn;                                        //  rebuilt every one of them in a VGPR pair, 18 v_mov per element)
            b0 = 0.5 - 0.5 * cs;
            b1 = (zl[i] - zc) * sn;
          } else {
            const double dl = zl[i] - zc, q2 = dl * dl;
            kt = sv_exp2_neg_sc(-q2 * p.h2);
            bz = dl;
            b0 = q2;
            b1 = 0.0;
          }
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
    pt.g_Z[(long long)r * M + tid] = g * 2.0 * p.cz + gzq[tid] + a_l * p.cz * p1sum;
  }
  if (tid == 0) {
    double t0 = 0.0, t1 = 0.0;
#pragma unroll
    for (int w = 0; w < NT; ++w) {
      t0 += redt[2 * w + 0];
      t1 += redt[2 * w + 1];
    }
    ev.gthitem[item * 2 + 0] = t0 * p.c0 + thq[0] + p.c0 * tpl[0];
    ev.gthitem[item * 2 + 1] = t1 * p.c1 + thq[1] + p.c1 * tpl[1];
  }
  if (SV_BWD_CLK) {
    SV_CK(11);
    if (blockIdx.x == 101 && tid == 0) {
      printf("CLK kt %d:", KT);
      for (int i = 1; i < 12; ++i) printf(" %lld", ck[i] - ck[0]);
      printf("\n");
    }
  }
}

// grid = R*K blocks of 64 NT G threads (G = 1, 2 or 4 teams of NT waves; the host picks G, sv_item_bwd_waves).
// FULL = gradients wrt Z / theta wanted (mask & 12).  M > 32: 78 KB of LDS and <= 256 registers: two blocks per CU.
#ifndef SV_BWD_WPE
#define SV_BWD_WPE 2    // waves per SIMD the compiler must leave room for (experiment builds: 3)
#endif
template <int NT, bool FULL>
__global__ __launch_bounds__(256, SV_BWD_WPE) void k_item_bwd(EngineView ev, LatentPtrTable ptab, unsigned mask,
                                                              const double* C, const double* d, double* shared) {
  extern __shared__ double sm[];
  const int r = blockIdx.x / ev.K, k = blockIdx.x % ev.K;
  if (ev.meta[k].ktype == SV_RBF)
    item_bwd_body<NT, SV_RBF, FULL>(ev, r, k, ptab.p[k], mask, sm);
  else
    item_bwd_body<NT, SV_PERIODIC, FULL>(ev, r, k, ptab.p[k], mask, sm);
  if (SV_BWD_STOP) return;
  fin_tail(ev, C, d, mask, shared, sm);    // sm is free again: every wave passes the barriers inside
}
