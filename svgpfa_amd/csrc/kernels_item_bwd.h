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

// LDS layout of one block (offsets in doubles) -- ONE constexpr function shared by the kernel (which places its buffers with
// it) and the host (which sizes the launch with it), so the two cannot disagree.  The per-wave partial sums of phase 1, the
// gather's scratch and the Gt / At matrix of phase 2 share one region: at the north-star shape (NT = 2, 2 waves) a block
// needs 38 KB, so four blocks (8 waves, what the registers allow) stay resident per CU.  The FULL variant keeps E in a
// buffer of its own at every M: Vt stays staged for phase 2 (until round 4 E took Vt's place at M <= 32 and Vt was fetched
// again behind the node tiles -- a second exposed memory latency per block for 9 KB of LDS that four blocks per CU leave
// unused anyway).
struct ItemBwdLayout {
  size_t Li, Vl, S, ul, al, zl, gt, gs, gq, gzq, szl, czl, wpart, tred, redz, redt, thq, p1l, cvd, tpl, Dl, total;
};
__host__ __device__ constexpr ItemBwdLayout sv_item_bwd_layout(int NT, int NWB, bool full) {
  const size_t MP = 16 * NT, LD = MP + 2;
  const size_t SZ = (MP * LD > SV_GATHER_SCR) ? MP * LD : SV_GATHER_SCR;
  const bool dsep = full;
  ItemBwdLayout o{};
  size_t p = 0;
  o.Li = p; p += MP * LD;          // Linv
  o.Vl = p; p += MP * LD;          // Vt; later C
  o.S = p; p += SZ;                // gather scratch; then Gt summed over the waves (rows padded to LD), later At
  o.ul = p; p += MP;               // ut
  o.al = p; p += MP;               // a = Linv^T ut
  o.zl = p; p += MP;               // Z
  o.gt = p; p += MP;
  o.gs = p; p += MP;
  o.gq = p; p += MP;               // quadrature part of the adjoint of ut
  o.gzq = p; p += MP;              // quadrature part of dZ (direct)
  o.szl = p; p += MP;              // sin of the inducing-point phases (periodic)
  o.czl = p; p += MP;              // cos
  o.wpart = p; p += (size_t)NWB * 2 * MP;   // per-wave partial sums of gu / dz behind the node tiles
  o.tred = p; p += (size_t)NWB * 2;
  o.redz = p; p += (size_t)NT * MP;   // per-panel row sums of the dZ contraction
  o.redt = p; p += 2 * (size_t)NT;    // per-panel dtheta partials
  o.thq = p; p += 8;               // quadrature part of dtheta (direct), 2 used
  o.p1l = p; p += MP;              // spike partial sums P1 of the trial
  o.cvd = p; p += MP;              // 1 / diag(L_S)
  o.tpl = p; p += 8;               // spike-direct theta partials of the trial, 2 used
  o.Dl = dsep ? p : o.Vl;          // E = Linv^T (Vt Vt^T - I) Linv (FULL): its own MP x LD buffer
  if (dsep) p += MP * LD;
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
              t = (ev.gCspk[(long long)r * NK + id] + ev.gCspkB[(long long)r * NK + id]) -
                  (ev.gCpart[o2] + cnk * ev.gCpart[o2 + K]);               // dC = spikes - (E^T mu + C . E^T var)
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
    const bool TW = (mask & 14u) != 0 && m.ktype == SV_PERIODIC;      // what the class kernels produced for this mask
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
template <int NT, int KT, bool FULL>
__device__ __forceinline__ void item_bwd_body(const EngineView& ev, const int r, const int k, const LatentPtrs pt, unsigned mask,
                              double* sm) {
  constexpr int MP = 16 * NT, LD = MP + 2, NTRI = NT * (NT + 1) / 2;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6, li = lane & 15, lg = lane >> 4;
  const int NTHB = blockDim.x, NWB = NTHB >> 6;   // 2 or 4 waves per block (host: 2 when the grid is large and M <= 32)
  const int item = r * ev.K + k;
  const LatentMeta m = ev.meta[k];
  const int M = m.M, Q = ev.Q, K = ev.K;
  const ItemBwdLayout lay = sv_item_bwd_layout(NT, NWB, FULL);   // buffer meanings: see the layout function
  double* Li = sm + lay.Li;
  double* Vl = sm + lay.Vl;
  double* S = sm + lay.S;
  double* Gl = S;                     // (symmetric, rows padded to LD: phase 2's Gt, later At)
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
  double* Dl = sm + lay.Dl;
  const KPar p = make_kpar(m, ev.theta + 2 * k);
  const long long base = m.offTT + (long long)r * m.PT;      // Vt: packed lower triangle
  const long long vbase = m.offM + (long long)r * M;
  const bool want_lat = (mask & (1u | 4u | 8u)) != 0;
  const bool want_zt = (mask & (4u | 8u)) != 0;
  long long ck[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  SV_CK(0);
  const GatherPre gpre = gather_issue(ev, m, pt, r, k, mask, want_zt);

  if (want_lat) {
    load_factors<NT>(ev, m, r, Li, Vl, ul, zl, true);
  } else if (tid < MP) {
    ul[tid] = 0.0;
    zl[tid] = 0.0;
  }
  if (tid < MP) al[tid] = (tid < M) ? ev.a[vbase + tid] : 0.0;
  // The gather of phase 2's inputs (for class-path items: the combine of the spike moments) runs HERE, beside the
  // staging of the factors: between the phases, where the values are used, its memory latency was exposed (12 us of the
  // kernel at the north-star size).  (Round 2 kept it between the phases above M = 32 after an intermediate version
  // returned wrong sums there: a compiler defect)
  SV_CK(1);
  gather_item_inputs<MP>(ev, m, pt, r, k, mask, want_zt, p, gpre, gs, p1l, cvd, gt, tpl, S);
  SV_CK(2);
  if (SV_BWD_STOP == 9) return;      // after the staging of the factors and the gather

  if (want_lat) {
    // ================================================================= phase 1: quadrature adjoints
    __syncthreads();
    if (KT == SV_PERIODIC && tid < MP) sv_sincos_quarter(zl[tid] * p.q4, szl[tid], czl[tid]);
    if (FULL) {
      // The adjoint of kappa(t, Z) at a node is gk = Linv^T [gm ut + 2 gv (Vt Vt^T - I) kt], kt = Linv kappa(t, Z), i.e.
      //     gk = gm a + 2 gv E kappa(t, Z)      with the per-item SYMMETRIC matrix  E = Linv^T (Vt Vt^T - I) Linv
      // one dense M x M x 16 product per tile on the RAW kernel values -- independent of the whitening chain of the same
      // tile (round 3 multiplied D = Linv^T (Vt Vt^T - I) with kt: the same 16 MFMAs, but behind the 12 of kt).  Rounding:
      // E kappa and D kt both carry an absolute error of order u cond(Kzz) (kt itself is computed with u sqrt(cond) and then
      // multiplied by D ~ sqrt(cond)); the MATLAB-fixture parity test (eps = 1e-5, cond 1e6) bounds it either way.
      // E is built once per item, one 16-column panel per wavefront.
      // Every wave builds a panel (NWB >= NT in every launch configuration; waves >= NT repeat panel wave % NT and store
      // the same values again): no wave-uniform branch around this register-heavy section.  ROCm 7.2's compiler was caught
      // placing a spill of threadIdx.x at the JOIN of such a branch in front of the exec restore (DESIGN.md section 9,
      // tests/tools/isa_exec_guard.py), where a wave that skipped the branch loses it.
      d4 Ep[NT];
      const int c = wave % NT;
      {
        d4 P[NT], T1[NT], T2[NT];
#pragma unroll
        for (int b = 0; b < NT; ++b)
#pragma unroll
          for (int q = 0; q < 4; ++q) P[b][q] = Li[(16 * b + 4 * q + lg) * LD + 16 * c + li];   // Linv[:, panel c]
        tri_chain<NT, 1>(T1, Vl, LD, P, li, lg);       // Vt^T Linv[:, c]
        tri_chain<NT, 0>(T2, Vl, LD, T1, li, lg);      // Vt Vt^T Linv[:, c]
#pragma unroll
        for (int a = 0; a < NT; ++a)
#pragma unroll
          for (int q = 0; q < 4; ++q) T2[a][q] -= P[a][q];
        tri_chain<NT, 1>(Ep, Li, LD, T2, li, lg);      // Linv^T (Vt Vt^T - I) Linv[:, c]
      }
#pragma unroll
      for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int q = 0; q < 4; ++q) Dl[(16 * a + 4 * q + lg) * LD + 16 * c + li] = Ep[a][q];
    }
    __syncthreads();
    SV_CK(3);
    if (SV_BWD_STOP == 10) return;   // after the D panels
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
    // periodic: kappa and its derivative bases follow from the node's own sin / cos and the inducing
    // points' (szl, czl) by angle addition (no per-pair sincos, no second exp).  Per-j constants are read
    // from LDS at their use sites (jo is an opaque 0: it keeps them out of long-lived registers).
    const double inv_s2 = 1.0 / p.s2;
    // Node lists of the item.  Segment 0: the trial's Q quadrature nodes with the adjoints the ELL kernel left in
    // [r][k][q] (contiguous for this item).  Segment 1 (statistics mode only, mask bit SV_MASK_STATS): the trial's spike
    // times with caller-given adjoints of the posterior mean / variance at the spikes, [k][S] -- the general form of the
    // spike term's backward, for likelihoods whose spike adjoint is not known a priori (svgpfa_latent_stats_backward).
    const int nseg = (mask & SV_MASK_STATS) ? 2 : 1;
    for (int seg = 0; seg < nseg; ++seg) {
    const long long s0 = seg ? ev.spk_off[r] : 0;
    const int Qs = seg ? (int)(ev.spk_off[r + 1] - s0) : Q;
    const double* gmq = seg ? ev.gmuspk + (long long)k * ev.S + s0 : ev.gmuq + ((long long)r * K + k) * Q;
    const double* gvq = seg ? ev.gvarspk + (long long)k * ev.S + s0 : ev.gvarq + ((long long)r * K + k) * Q;
    const double* tqr = seg ? ev.spk_t + s0 : ev.tq + (long long)r * Q;

    const int ntiles = (Qs + 15) >> 4;
    // inputs of the NEXT tile are fetched one iteration ahead (global latency hidden behind the MFMA chain)
    double tq_n = 0.0, gm_n = 0.0, gv_n = 0.0;
    if (wave < ntiles) {
      const int nn = 16 * wave + li;
      tq_n = tqr[(nn < Qs) ? nn : 0];
      gm_n = (nn < Qs) ? gmq[nn] : 0.0;
      gv_n = (nn < Qs) ? gvq[nn] : 0.0;
    }
    for (int tile = wave; tile < ntiles; tile += NWB) {
      const double tq = tq_n, gm = gm_n, gvv = gv_n;
      if (tile + NWB < ntiles) {
        const int nn = 16 * (tile + NWB) + li;
        tq_n = tqr[(nn < Qs) ? nn : 0];
        gm_n = (nn < Qs) ? gmq[nn] : 0.0;
        gv_n = (nn < Qs) ? gvq[nn] : 0.0;
      }
      if (tile == wave) SV_CK(4);
      double stq = 0.0, ctq = 0.0;
      if (KT == SV_PERIODIC) sv_sincos_quarter(tq * p.q4, stq, ctq);
      int jo = lg;
      asm volatile("" : "+v"(jo));
      d4 X0[NT];
      d4 csv[(KT == SV_PERIODIC && FULL) ? NT : 1];      // cos of the pair's phase, kept for the adjoint (FULL; round 4: 16 registers
                                                          // the kernel has to spare since phase 1 lost its transposes, two FMAs per pair)
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
            if (FULL) csv[a][q] = cs;
            X0[a][q] = p.s2 * sv_exp2_neg(fma(cs, p.h2, -p.h2));
          }
        }
      // adjoints of the four nodes lg + 4 q of this tile: what this lane's registers of the TRANSPOSED whitened tile
      // below belong to (L1 / L2 hits: the tile's 16 values are one 128-byte line, fetched by the prefetch above)
      double gm4[4], gv4[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int n4 = 16 * tile + lg + 4 * q;
        gm4[q] = (n4 < Qs) ? gmq[n4] : 0.0;
        gv4[q] = (n4 < Qs) ? gvq[n4] : 0.0;
      }
      if (tile == wave) SV_CK(5);
      // Whitened tile TRANSPOSED: KwT[t][i] = sum_j kappa(t, z_j) Linv[i][j] -- the raw kernel tile is the A operand as it
      // stands (lane = node, registers = inducing points), Linv's fragments the B operand; the accumulator then holds
      // node lg + 4 q of inducing point 16 a + li, which is BOTH operand layouts of the contraction over the nodes
      //     Gt += KwT^T diag(gv) KwT,
      // so that product needs no transpose through LDS (round 3: 8 ds_write + 12 ds_read + two fences per tile, and the
      // kernel's LDS bank conflicts).
      d4 KwT[NT];
#pragma unroll
      for (int a = 0; a < NT; ++a) {
        d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int b = 0; b <= a; ++b)
#pragma unroll
          for (int q = 0; q < 4; ++q)
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(X0[b][q], lds_frag(&Li[(16 * a + li) * LD + 16 * b + 4 * q + lg]), acc, 0, 0, 0);
        KwT[a] = acc;
      }
      if (tile == wave) SV_CK(6);
      d4 Gk[NT];
      if (FULL) {
#pragma unroll
        for (int a = 0; a < NT; ++a) {
          d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
          for (int b = 0; b < NT; ++b)
#pragma unroll
            for (int q = 0; q < 4; ++q)
              acc = __builtin_amdgcn_mfma_f64_16x16x4f64(lds_frag(&Dl[(16 * a + li) * LD + 16 * b + 4 * q + lg]), X0[b][q], acc, 0, 0, 0);
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
      if (tile == wave) SV_CK(7);
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
      if (tile == wave) SV_CK(8);
    }
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
    // Gt: the waves' accumulator tiles added in wave order into the (idle) scratch region, both triangles (tile (a, b),
    // a > b, is mirrored into (b, a)), rows padded to LD; wave 0 writes ahead of the barrier.  Gt's padding rows / columns
    // are exact zeros: sums of products with zero-padded rows of Linv
    auto gt_tiles = [&](const bool first) {
      int t = 0;
#pragma unroll
      for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int b = 0; b <= a; ++b) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int row = 16 * a + lg + 4 * q, col = 16 * b + li;
            const double v = first ? GW[t][q] : Gl[row * LD + col] + GW[t][q];
            Gl[row * LD + col] = v;
            if (a != b) Gl[col * LD + row] = v;
          }
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
      for (int q = msub; q <= mrow; q += LPR) s = fma(Li[mrow * LD + q], gs[q], s);
    for (int o = 1; o < LPR; o <<= 1) s += __shfl_xor(s, o, 64);
    if (msub == 0 && mrow < M) gt[mrow] = gq[mrow] + s;
  }
  __syncthreads();

  const int c = wave;                     // panel index of this wavefront (waves >= NT idle through the panel work)
  const bool pan = c < NT;
  if (mask & 1u) {
    {   // dm = Linv^T (gt - ut)
      double s = 0.0;
      if (mrow < M)
        for (int q = mrow + msub; q < M; q += LPR) s = fma(Li[q * LD + mrow], gt[q] - ul[q], s);
      for (int o = 1; o < LPR; o <<= 1) s += __shfl_xor(s, o, 64);
      if (msub == 0 && mrow < M) pt.g_mean[(long long)r * M + mrow] = s;
    }
    if (pan) {
      // dL_S = tril(Linv^T (2 Gt - I) Vt) + diag(1 / L_S,ii): this wave's 16-column panel
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
  double* Cs = Vl;
  if (pan) {
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
      for (int q = 0; q < 4; ++q) Cs[(16 * a + 4 * q + lg) * LD + 16 * c + li] = Cm[a][q];
  }
  __syncthreads();
  if (SV_BWD_STOP == 7) return;      // after C, Vt Vt^T
  // ---- At (in place over Gt): each element is read and written by exactly one lane; the transposed reads of Gt
  //      itself do not occur (only C is read transposed), so no wave sees a half-updated panel
  if (pan) {
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        int i = 16 * a + 4 * q + lg, j = 16 * c + li;
        double v = -0.5 * (gt[i] * ul[j] + ul[i] * gt[j]) + 0.5 * ul[i] * ul[j] + Gl[i * LD + j] +
                   0.5 * (VV[a][q] - (i == j ? 1.0 : 0.0)) - Cs[i * LD + j] - Cs[j * LD + i];
        Gl[i * LD + j] = v;
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

// grid = R*K blocks of 64 NWB threads (NWB = 2 or 4 >= NT); FULL = gradients wrt Z / theta wanted (mask & 12).
// M > 32 (cfg5): one block per CU (115 / 151 KB of LDS), one wave per SIMD.  Eight waves per block were measured for the
// value-only variant (it fits 256 registers): no gain -- 13 node tiles give the busiest SIMD four tiles either way, and the
// item's serial sections (staging, cross-wave sums) overlap with nothing while a block is alone on its CU.
#ifndef SV_BWD_WPE
#define SV_BWD_WPE 2    // waves per SIMD the compiler must leave room for at M <= 32 (experiment builds: 3)
#endif
template <int NT, bool FULL>
__global__ __launch_bounds__(256, (NT <= 2 ? SV_BWD_WPE : 1)) void k_item_bwd(EngineView ev, LatentPtrTable ptab, unsigned mask,
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
