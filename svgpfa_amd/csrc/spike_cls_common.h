// Device functions of the spike pass's class path that kernels of OTHER launches run as riders (kernels_spike_cls.h has the
// class kernels themselves and the description of the method):
//   spike_class_table / spike_class_trial   the class table and the per-(trial, class) window / period / harmonics, by the
//                                           class blocks in front of k_build_reg's item blocks (with every kernel build),
//                                           or by k_spike_classes when the spikes arrive after the build
//   spike_pre_item                          W = s2 c_n (Abar_n, Bbar_n) of one (trial, latent), by wave 0 of k_prep_mfma's
//                                           block for that item (it has a = Kzz^-1 m at hand), or by k_spike_pre
// Both were launches of their own on a side stream until the kernel trace of an 8-way shard (63 trials) showed what that
// costs where every kernel is one under-filled round: 10 + 6 us of kernels, but ~45 us of events, cross-queue waits and
// launch gaps on the critical chain.
#pragma once
#include "engine_common.h"

struct SpikeClassScratch {               // LDS of one class-table block
  SpikeClassTab tab;
  double par[SV_MAX_K];                  // per latent: RBF h = 1 / (2 l^2); periodic 4 / period
  double lead[SV_MAX_K];                 // per class: RBF h of the leader (its longest lengthscale); periodic 4 / period
  long long offMk[SV_MAX_K];
  int Mk[SV_MAX_K];
  int clsH[SV_MAX_K], clsM[SV_MAX_K];    // per class: periodic harmonics; sum of the members' M
};

// The class table, by ONE wavefront (all 64 lanes call), one lane per latent (K <= 32), by wave votes: no serial walk
// through LDS.  snap != nullptr: kernel parameters from the caller's tensors (the build that snapshots them is in flight in
// this very launch) and the periodic kernels' harmonic counts from the series recurrence itself (fws: K rows of SV_FH + 3
// doubles of LDS); else from the engine's snapshot and fourH.  The caller synchronises before the table is read.
__device__ inline void spike_class_table(const EngineView& ev, SpikeClassScratch& cs, const SnapTable* snap, double* fws) {
  SpikeClassTab& tab = cs.tab;
  const int lane = threadIdx.x & 63, K = ev.K;
  const bool in = lane < K;
  const int kl = in ? lane : 0;
  LatentMeta m = ev.meta[kl];
  const KPar p = make_kpar(m, snap ? snap->theta[kl] : ev.theta + 2 * kl);
  const double pk = (m.ktype == SV_RBF) ? p.h : 4.0 * p.ip;
  int Hl = -1;
  if (in && m.ktype == SV_PERIODIC) {
    if (!snap) Hl = ev.fourH[lane];
    else if (ev.per_series) Hl = periodic_series(0.5 * p.h, m.M, nullptr, fws + lane * (SV_FH + 3));
  }
  int cls = -1, slot = 0, ncls = 0;
  // periodic: one class per distinct period (the lowest latent with that period leads it; at most SV_CLS_MAXMEM members,
  // further ones stay with the pair loop)
  {
    const bool el = in && m.ktype == SV_PERIODIC && Hl >= 0;
    int ld = -1;
    for (int j = 0; j < K; ++j) {
      const double pj = __shfl(pk, j, 64);
      const int ej = __shfl((int)el, j, 64);
      if (el && ej && pj == pk && ld < 0) ld = j;
    }
    const unsigned long long leaders = __ballot(el && ld == lane);
    for (int j = 0; j < K; ++j) {                    // members of leader j, in latent order
      const unsigned long long mem = __ballot(el && ld == j);
      if ((leaders >> j) & 1ull) {
        const int g = __popcll(leaders & ((1ull << j) - 1ull));
        const int sl = __popcll(mem & ((1ull << lane) - 1ull));
        if (el && ld == j && sl < SV_CLS_MAXMEM) {
          cls = g;
          slot = sl;
        }
      }
    }
    ncls = __popcll(leaders);
    if (el && cls >= 0 && ld == lane) {
      tab.type[cls] = SV_PERIODIC;
      cs.lead[cls] = pk;
    }
  }
  // RBF: greedy classes from the longest lengthscale (smallest h) down, members within a factor 2 of the leader
  if (ev.rbf_series) {
    const bool el = in && m.ktype == SV_RBF && pk > 0.0 && pk < 1e300;
    for (;;) {
      const bool open = el && cls < 0;
      double hmin = open ? pk : 1e308;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) hmin = fmin(hmin, __shfl_xor(hmin, o, 64));
      if (!(hmin < 1e308)) break;
      const unsigned long long cand = __ballot(open && pk <= 4.0 * hmin);
      const int sl = __popcll(cand & ((1ull << lane) - 1ull));
      if (open && pk <= 4.0 * hmin && sl < SV_CLS_MAXMEM) {
        cls = ncls;
        slot = sl;
      }
      if (lane == 0) {
        tab.type[ncls] = SV_RBF;
        cs.lead[ncls] = hmin;
      }
      ++ncls;
    }
  }
  if (in) {
    tab.cls_of[lane] = cls;
    tab.slot_of[lane] = slot;
    cs.par[lane] = pk;
    cs.Mk[lane] = m.M;
    cs.offMk[lane] = m.offM;
    if (cls >= 0) tab.mem[cls][slot] = lane;
  }
  for (int g = 0; g < ncls; ++g) {                    // per class: members, harmonics (periodic), sum of M
    const bool mine = in && cls == g;
    const int nm = __popcll(__ballot(mine));
    int hh = mine ? Hl : 0, mm = mine ? m.M : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      hh = max(hh, __shfl_xor(hh, o, 64));
      mm += __shfl_xor(mm, o, 64);
    }
    if (lane == 0) {
      tab.nmem[g] = nm;
      cs.clsH[g] = hh;
      cs.clsM[g] = mm;
    }
  }
  if (lane == 0) tab.ncls = ncls;
}

// Per-trial entries of the class path -- window / period / harmonics of every (trial, class), ser / sitem of every latent,
// the trial's pair flag -- by ONE wavefront (all 64 lanes call) from a finished table in LDS.
__device__ inline void spike_class_trial(const EngineView& ev, const SpikeClassScratch& cs, const SnapTable* snap, int r) {
  const SpikeClassTab& tab = cs.tab;
  const int lane = threadIdx.x & 63, K = ev.K, R = ev.R;
  const double slo = ev.spk_lo[r], shi = ev.spk_hi[r];
  const bool has_spikes = ev.spk_off[r + 1] > ev.spk_off[r];
  bool pair = false;                                   // some item of this trial is left to the pair-by-pair pass
  for (int k = lane; k < K; k += 64)
    if (tab.cls_of[k] < 0) {
      ev.ser[r * K + k] = 0;
      ev.sitem[r * K + k] = SpikeItem{-1, -1, 0.0, 0.0};
      pair = true;
    }
  const double PI = 3.14159265358979323846;
  // Extent of every RBF member's inducing points in this trial: all latents' loads in flight at once (one latent after the
  // other, each behind its own dependent load, made this function 15 us long), reduced per latent; lane k keeps latent k's.
  double zlo_k = slo, zhi_k = slo;
  {
    double zv[SV_MAX_K];
#pragma unroll
    for (int k = 0; k < SV_MAX_K; ++k) {
      zv[k] = slo;                                     // neutral: the window contains [slo, shi] anyway
      if (k < K && tab.cls_of[k] >= 0 && tab.type[tab.cls_of[k]] == SV_RBF && lane < cs.Mk[k]) {
        const double* zk = snap ? snap->Z[k] : ev.Zs + cs.offMk[k];
        zv[k] = zk[(long long)r * cs.Mk[k] + lane];
      }
    }
#pragma unroll
    for (int k = 0; k < SV_MAX_K; ++k)
      if (k < K) {                                     // wave-uniform
        double lo = zv[k], hi = zv[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
          lo = fmin(lo, __shfl_xor(lo, o, 64));
          hi = fmax(hi, __shfl_xor(hi, o, 64));
        }
        if (lane == k) {
          zlo_k = lo;
          zhi_k = hi;
        }
      }
  }
  for (int g = 0; g < tab.ncls; ++g) {
    double t0 = 0.0, sc = cs.lead[g];
    int H = cs.clsH[g];
    if (tab.type[g] == SV_RBF) {
      // window of the (trial, class): all spike times of the trial and all inducing points of the members
      double lo = slo, hi = shi;
      for (int s = 0; s < tab.nmem[g]; ++s) {
        const int k = tab.mem[g][s];
        lo = fmin(lo, __shfl(zlo_k, k, 64));
        hi = fmax(hi, __shfl(zhi_k, k, 64));
      }
      const double L = (hi - lo) + sqrt(41.5 / cs.lead[g]);     // wrapped copies below e^-41.5 = 1e-18 for every member
      H = 0;
      for (int s = 0; s < tab.nmem[g]; ++s) {
        const double q = PI * PI / (cs.par[tab.mem[g][s]] * L * L);   // c_n ~ exp(-q n^2)
        const double nmax = sqrt(46.0 / q);                           // n c_n / c_0 < 64 e^-46 = 7e-19 beyond
        if (!(nmax < (double)SV_FH)) H = SV_FH + 1;
        else H = ((int)nmax + 1 > H) ? (int)nmax + 1 : H;
      }
      // per-spike cost of the series (6 instructions per harmonic, rows ~60 % full) against the members' pair loops
      if (H > SV_FH || 10 * (H + 2) > 21 * cs.clsM[g]) H = -1;
      t0 = lo;
      sc = 4.0 / L;
    }
    if (lane == 0) {
      ev.cls_t0[(long long)g * R + r] = t0;
      ev.cls_sc[(long long)g * R + r] = sc;
      ev.cls_H[(long long)g * R + r] = H;
    }
    if (lane < tab.nmem[g]) {
      const int k = tab.mem[g][lane];
      ev.ser[r * K + k] = H >= 0 ? 1 : 0;
      ev.sitem[r * K + k] = SpikeItem{H, g, t0, sc};
      if (H < 0) pair = true;
    }
  }
  // one flag per trial, no reset and no atomics: the pair pass ORs them (k_spike)
  const bool anyp = __any(pair);
  if (lane == 0) ev.pair_any[r] = (has_spikes && anyp) ? 1 : 0;
}

// device copy of the table (the class kernels and the gather read it), by the 64 lanes of one wavefront
__device__ inline void spike_class_store(const EngineView& ev, const SpikeClassScratch& cs) {
  int* dst = (int*)ev.cls;
  const int* src = (const int*)&cs.tab;
  for (int i = threadIdx.x & 63; i < (int)(sizeof(SpikeClassTab) / sizeof(int)); i += 64) dst[i] = src[i];
}

// Series coefficients of latent k on the (trial, class) period: rows c_n | g_n | e_n for n = 0 .. H into cf (LDS,
// 3 x (SV_FH + 1)); periodic latents copy their table (k_build_reg / periodic_series), RBF latents evaluate
//   c_n = (2) sqrt(pi / h) / L exp(-pi^2 n^2 / (h L^2)),  g_n = -dc_n/dh = c_n (1/(2h) - pi^2 n^2 / (h^2 L^2)),
//   e_n = n w c_n / (2 h),  w = 2 pi / L,  L = 4 / sc.     All 64 lanes call it; the caller synchronises.
__device__ __forceinline__ void cls_coefficients(const EngineView& ev, int k, int ktype, double h, double sc, int H,
                                                 double* cf) {
  const int lane = threadIdx.x & 63;
  constexpr int NF = SV_FH + 2;
  if (ktype == SV_PERIODIC) {
    const double* tab = ev.four + (size_t)k * SV_FROWS * NF;
    for (int n = lane; n <= H; n += 64) {
      cf[n] = tab[n];
      cf[(SV_FH + 1) + n] = tab[NF + n];
      cf[2 * (SV_FH + 1) + n] = tab[2 * NF + n];
    }
  } else {
    const double PI = 3.14159265358979323846;
    const double L = 4.0 / sc, q = PI * PI / (h * L * L), w = 2.0 * PI / L, c0 = sqrt(PI / h) / L;
    for (int n = lane; n <= H; n += 64) {
      const double cn = (n ? 2.0 : 1.0) * c0 * exp(-q * (double)n * (double)n);
      cf[n] = cn;
      cf[(SV_FH + 1) + n] = cn * (0.5 / h - q * (double)n * (double)n / h);
      cf[2 * (SV_FH + 1) + n] = (double)n * w * cn * (0.5 / h);
    }
  }
}

// W[r][k][2 n + {0, 1}] = s2 c_n (Abar_n, Bbar_n),  Abar_n = sum_j a_j cos(n phi_j),  Bbar_n = sum_j a_j sin(n phi_j),
// phi_j = w (z_j - t0): the posterior mean at a spike is mu_s = sum_n [cos(n theta_s) W_2n + sin(n theta_s) W_2n+1].
// Entries beyond H are written as zeros (k_spike_cls works in whole register passes).  By ONE wavefront (all 64 lanes
// call); aj = a_lane of the item (0 beyond M); scratch: SV_PRE_LDS doubles of LDS.
// Lane = HARMONIC: lane n walks the item's inducing points (a_j and phi_j are broadcast LDS reads) with one direct sincos of
// n phi_j each -- M independent iterations per lane, no sum across lanes.  (Round 3 had lane = inducing point, harmonics
// by rotation, and a transposed LDS reduction per four harmonics: a dependent chain of ~10 us on the one wave of
// k_prep_mfma that carried it.)
#define SV_PRE_LDS (2 * 64 + 3 * (SV_FH + 1))
__device__ inline void spike_pre_item(const EngineView& ev, int item, int r, int k, const LatentMeta& m, double aj,
                                      double* scratch) {
  double* al = scratch;                    // a_j
  double* pl = scratch + 64;               // phi_j in quarter turns
  double* cf = scratch + 128;
  const int lane = threadIdx.x & 63;
  const SpikeItem it = ev.sitem[item];
  if (it.H < 0) return;
  const int H = it.H, M = m.M;
  const double sc = it.sc, t0 = it.t0;
  const KPar p = make_kpar(m, ev.theta + 2 * k);
  double zj = t0;
  if (lane < M) zj = ev.Zs[m.offM + (long long)r * M + lane];
  al[lane] = (lane < M) ? aj : 0.0;
  pl[lane] = (zj - t0) * sc;
  cls_coefficients(ev, k, m.ktype, p.h, sc, H, cf);
  double* W = ev.Wb + (long long)item * SV_WCOLS;
  for (int c = 2 * (H + 1) + lane; c < SV_WCOLS; c += 64) W[c] = 0.0;
  lds_wave_fence();
  const int NH = H + 1;
  if (NH <= 32) {
    // Few harmonics (the usual case: 14 .. 21): with lane = harmonic alone a third of the wave walks all M inducing points
    // -- ~1 100 issued instructions per item, a sixth of k_front's VALU work at the north-star size (round 4).  The
    // inducing points are dealt over G groups of NH lanes (lane = harmonic n + NH g walks j = g, g + G, ...), and the G
    // partial sums of a harmonic are added in group order.
    const int G = (NH <= 16) ? 4 : ((NH <= 21) ? 3 : 2);
    const int g = lane / NH, n = lane - g * NH;
    double sa = 0.0, sb = 0.0;
    if (g < G) {
      const double dn = (double)n;
      for (int j = g; j < M; j += G) {
        double sn, cs;
        sv_sincos_quarter(pl[j] * dn, sn, cs);
        sa = fma(al[j], cs, sa);
        sb = fma(al[j], sn, sb);
      }
    }
    double ta = 0.0, tb = 0.0;
    for (int gg = 0; gg < G; ++gg) {                 // (every lane takes part in the shuffles; sources n + NH gg < 64)
      ta += __shfl(sa, n + NH * gg, 64);
      tb += __shfl(sb, n + NH * gg, 64);
    }
    if (g == 0) {
      const double w = p.s2 * cf[n];
      W[2 * n + 0] = w * ta;
      W[2 * n + 1] = w * tb;
    }
    return;
  }
  for (int n = lane; n <= H; n += 64) {
    const double dn = (double)n;
    double sa = 0.0, sb = 0.0;
    for (int j = 0; j < M; ++j) {
      double sn, cs;
      sv_sincos_quarter(pl[j] * dn, sn, cs);
      sa = fma(al[j], cs, sa);
      sb = fma(al[j], sn, sb);
    }
    const double w = p.s2 * cf[n];
    W[2 * n + 0] = w * sa;
    W[2 * n + 1] = w * sb;
  }
}

// The combine of one (trial, latent) item by ONE wavefront (all 64 lanes call; lds: SV_COMB_LDS doubles of its own) -- the
// body of k_spike_comb (kernels_spike_cls.h), and of the rider blocks of k_quad_ell_mfma that do the same work inside the
// ELL kernel's launch at full size (kernels_mfma.h).
#define SV_COMB_LDS (SV_ACOLS + 3 * (SV_FH + 1))
// tw: the class kernels carried the time-weighted moments of periodic classes (four values per harmonic instead of two):
// only when the gradient wrt the kernel parameters is wanted -- the period's is their one consumer (T1); round 5: until
// then every pass with any gradient carried them, at half again the price per spike and harmonic.
template <bool FULL>
__device__ __forceinline__ void spike_comb_item(const EngineView& ev, const int item, double* lds, const int tw) {
  double* Fs = lds;                        // SV_ACOLS moments
  double* cf = lds + SV_ACOLS;             // 3 x (SV_FH + 1) coefficients
  const int lane = threadIdx.x & 63, K = ev.K;
  const SpikeItem it = ev.sitem[item];
  if (it.H < 0) return;
  const int r = item / K, k = item % K;
  const LatentMeta m = ev.meta[k];
  const int M = m.M;
  const int H = it.H;
  const double sc = it.sc, t0 = it.t0;
  const KPar p = make_kpar(m, ev.theta + 2 * k);
  const bool TW = FULL && tw && m.ktype == SV_PERIODIC;
  const int NV = TW ? 4 : 2, ncol = NV * (H + 1);
  const int tl0 = ev.tile_first[r], tl1 = ev.tile_first[r + 1];
  double zj = t0, aj = 0.0;
  if (lane < M) {                          // requested before the coefficients are computed
    zj = ev.Zs[m.offM + (long long)r * M + lane];
    aj = ev.a[m.offM + (long long)r * M + lane];
  }
  cls_coefficients(ev, k, m.ktype, p.h, sc, H, cf);
  for (int c = lane; c < ncol; c += 64) {
    double s = 0.0;
    for (int t = tl0; t < tl1; t += 8) {            // eight tiles' partials in flight, added in tile order
      double v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = ev.Apart[((long long)(t + u < tl1 ? t + u : t) * K + k) * SV_ACOLS + c];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += (t + u < tl1) ? v[u] : 0.0;
    }
    Fs[c] = s;
  }
  lds_wave_fence();
  double t0acc = 0.0, t1acc = 0.0;
  if (lane < M) {
    double sphi, cphi;
    sv_sincos_quarter((zj - t0) * sc, sphi, cphi);
    const double* cN = cf;
    const double* gN = cf + (SV_FH + 1);
    const double* eN = cf + 2 * (SV_FH + 1);
    double cj = 1.0, sj = 0.0, P0 = 0.0, P1 = 0.0, d0 = 0.0, d1 = 0.0;
    for (int n = 0; n <= H; ++n) {
      const double An = Fs[n * NV + 0], Bn = Fs[n * NV + 1];
      const double Cc = fma(cj, An, sj * Bn), Ss = fma(cj, Bn, -(sj * An));
      P0 = fma(cN[n], Cc, P0);
      if (FULL) {
        const double en = eN[n];
        P1 = fma(en, Ss, P1);
        d0 = fma(gN[n], Cc, d0);
        if (TW) d1 = fma(en, fma(cj, Fs[n * NV + 3], -(sj * Fs[n * NV + 2])) - zj * Ss, d1);
      }
      const double tq = cj * cphi - sj * sphi;
      sj = fma(sj, cphi, cj * sphi);
      cj = tq;
    }
    const long long o = (long long)r * ev.sumM + m.colM + lane;
    ev.P0t[o] = P0;
    if (FULL) {
      ev.P1t[o] = P1;
      t0acc = aj * d0;
      t1acc = aj * d1;
    }
  }
  if (FULL) {
    const double ts0 = wave_sum(t0acc), ts1 = wave_sum(t1acc);
    if (lane == 0) {
      ev.Tt[(long long)item * 2 + 0] = ts0;
      ev.Tt[(long long)item * 2 + 1] = ts1;
    }
  }
}

// Two items per wavefront (round 5; M <= 32): lanes 0 .. 31 combine item0, lanes 32 .. 63 item0 + 1 -- the per-inducing-point
// half of spike_comb_item keeps 32 of its 64 lanes busy at most, and as riders of the ELL kernels the combine's blocks hold a
// CU's LDS two at a time: half the waves are half the rounds.  Same sums in the same order as spike_comb_item (the half-wave
// sum adds the two rows of 16 exactly as wave_sum adds them to two rows of zeros).  lds: 2 SV_COMB_LDS doubles.
template <bool FULL>
__device__ __forceinline__ void spike_comb_pair(const EngineView& ev, const int item0, const int nitems, double* lds, const int tw) {
  const int lane = threadIdx.x & 63, hl = lane & 31, half = lane >> 5, K = ev.K;
  double* Fs = lds + half * SV_COMB_LDS;   // SV_ACOLS moments
  double* cf = Fs + SV_ACOLS;              // 3 x (SV_FH + 1) coefficients
  const bool there = item0 + half < nitems;
  const int item = there ? item0 + half : item0;
  const SpikeItem it = ev.sitem[item];
  const bool live = there && it.H >= 0;
  if (!__any(live)) return;
  const int r = item / K, k = item % K;
  const LatentMeta m = ev.meta[k];
  const int M = m.M;
  const int H = live ? it.H : -1;          // (a half without an item runs empty loops)
  const double sc = it.sc, t0 = it.t0;
  const KPar p = make_kpar(m, ev.theta + 2 * k);
  const bool TW = FULL && tw && m.ktype == SV_PERIODIC;
  const int NV = TW ? 4 : 2, ncol = NV * (H + 1);
  const int tl0 = ev.tile_first[r], tl1 = ev.tile_first[r + 1];
  double zj = t0, aj = 0.0;
  if (live && hl < M) {
    zj = ev.Zs[m.offM + (long long)r * M + hl];
    aj = ev.a[m.offM + (long long)r * M + hl];
  }
  {   // the series coefficients of this half's latent (cls_coefficients with 32 lanes)
    constexpr int NF = SV_FH + 2;
    if (m.ktype == SV_PERIODIC) {
      const double* tab = ev.four + (size_t)k * SV_FROWS * NF;
      for (int n = hl; n <= H; n += 32) {
        cf[n] = tab[n];
        cf[(SV_FH + 1) + n] = tab[NF + n];
        cf[2 * (SV_FH + 1) + n] = tab[2 * NF + n];
      }
    } else {
      const double PI = 3.14159265358979323846, h = p.h;
      const double L = 4.0 / sc, q = PI * PI / (h * L * L), w = 2.0 * PI / L, c0 = sqrt(PI / h) / L;
      for (int n = hl; n <= H; n += 32) {
        const double cn = (n ? 2.0 : 1.0) * c0 * exp(-q * (double)n * (double)n);
        cf[n] = cn;
        cf[(SV_FH + 1) + n] = cn * (0.5 / h - q * (double)n * (double)n / h);
        cf[2 * (SV_FH + 1) + n] = (double)n * w * cn * (0.5 / h);
      }
    }
  }
  for (int c = hl; c < ncol; c += 32) {
    double s = 0.0;
    for (int t = tl0; t < tl1; t += 8) {            // eight tiles' partials in flight, added in tile order
      double v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = ev.Apart[((long long)(t + u < tl1 ? t + u : t) * K + k) * SV_ACOLS + c];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += (t + u < tl1) ? v[u] : 0.0;
    }
    Fs[c] = s;
  }
  lds_wave_fence();
  double t0acc = 0.0, t1acc = 0.0;
  if (live && hl < M) {
    double sphi, cphi;
    sv_sincos_quarter((zj - t0) * sc, sphi, cphi);
    const double* cN = cf;
    const double* gN = cf + (SV_FH + 1);
    const double* eN = cf + 2 * (SV_FH + 1);
    double cj = 1.0, sj = 0.0, P0 = 0.0, P1 = 0.0, d0 = 0.0, d1 = 0.0;
    for (int n = 0; n <= H; ++n) {
      const double An = Fs[n * NV + 0], Bn = Fs[n * NV + 1];
      const double Cc = fma(cj, An, sj * Bn), Ss = fma(cj, Bn, -(sj * An));
      P0 = fma(cN[n], Cc, P0);
      if (FULL) {
        const double en = eN[n];
        P1 = fma(en, Ss, P1);
        d0 = fma(gN[n], Cc, d0);
        if (TW) d1 = fma(en, fma(cj, Fs[n * NV + 3], -(sj * Fs[n * NV + 2])) - zj * Ss, d1);
      }
      const double tq = cj * cphi - sj * sphi;
      sj = fma(sj, cphi, cj * sphi);
      cj = tq;
    }
    const long long o = (long long)r * ev.sumM + m.colM + hl;
    ev.P0t[o] = P0;
    if (FULL) {
      ev.P1t[o] = P1;
      t0acc = aj * d0;
      t1acc = aj * d1;
    }
  }
  if (FULL) {
    double ts0 = row16_sum(t0acc), ts1 = row16_sum(t1acc);
    ts0 += __shfl_xor(ts0, 16, 64);
    ts1 += __shfl_xor(ts1, 16, 64);
    if (hl == 0 && live) {
      ev.Tt[(long long)item * 2 + 0] = ts0;
      ev.Tt[(long long)item * 2 + 1] = ts1;
    }
  }
}

// One rider wave of an ELL kernel: its unit of the combine -- one item, or (flags bit 2, M <= 32) two.  flags: bit 0 full,
// bit 1 time-weighted moments.  sm: the block's dynamic LDS, a wave's area is SV_COMB_LDS doubles per item.
__device__ __forceinline__ void sv_comb_rider(const EngineView& ev, const int unit, const int ncomb, const int flags, double* sm,
                                              const int wave) {
  if (flags & 4) {
    const int item0 = 2 * unit;
    if (item0 < ncomb) {
      double* lds = sm + wave * 2 * SV_COMB_LDS;
      if (flags & 1) spike_comb_pair<true>(ev, item0, ncomb, lds, (flags >> 1) & 1);
      else spike_comb_pair<false>(ev, item0, ncomb, lds, 0);
    }
  } else if (unit < ncomb) {
    double* lds = sm + wave * SV_COMB_LDS;
    if (flags & 1) spike_comb_item<true>(ev, unit, lds, (flags >> 1) & 1);
    else spike_comb_item<false>(ev, unit, lds, 0);
  }
}
