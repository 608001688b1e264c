// Device functions of the spike-time pass's PAIR-BY-PAIR form (kernels_spike.h has the description and the kernel): the
// per-item bodies and the work loop of one single-wave worker (pair_pass).  A header of its own because the pass has two
// hosts: the k_spike kernel (svgpfa_api.hip) and, when the class path runs too, the workers of the class kernels
// (kernels_spike_cls.h, a translation unit of its own).
#pragma once
#include "engine_common.h"
#include "sv_math.h"

#ifndef SV_SPIKE_JG
#define SV_SPIKE_JG 8     // inducing points per LDS reduction group of the RBF / value-only spike pass
#endif

// Per-neuron sums of v over the chunk when lane l owns the SPL CONSECUTIVE spikes [l SPL, (l + 1) SPL) (keys = neuron
// ids, non-decreasing over the chunk, < 0 = padding at the end): a serial segmented sum inside the lane, ONE wave scan
// over the lanes' tail partials (sorted keys: the lanes before l whose tail key equals l's head key are exactly those
// that continue its head segment), then every segment end stores its total -- each neuron has exactly one writer.
// Two planes per (trial, neuron, latent): outA takes a segment that lies inside the chunk (outB is then cleared) and the
// first part of a segment cut by the chunk's end (tailcut: the chunk's LAST segment continues in the next chunk, which
// writes outB); headcont: the chunk's FIRST segment continues one of the previous chunk and goes to outB.  Plane A of every
// existing segment is written on every pass; plane B only where a chunk boundary cuts the segment -- ev.seg_cut (static)
// tells the reader where it counts, so the planes stay valid when an item moves between this pass and the class path.
template <int SPL>
__device__ __forceinline__ void seg_emit_lane(const int (&key)[SPL], const double (&v)[SPL], double* outA, double* outB,
                                              bool headcont, bool tailcut, int K, int lane) {
  double run[SPL];
  run[0] = v[0];
#pragma unroll
  for (int i = 1; i < SPL; ++i) run[i] = (key[i] == key[i - 1]) ? run[i - 1] + v[i] : v[i];
  const int kt = key[SPL - 1];
  double t = run[SPL - 1];
#pragma unroll
  for (int dd = 1; dd < 64; dd <<= 1) {
    const double tu = __shfl_up(t, dd, 64);
    const int ku = __shfl_up(kt, dd, 64);
    if (lane >= dd && ku == kt) t += tu;
  }
  const double tprev = __shfl_up(t, 1, 64);
  const int kprev = __shfl_up(kt, 1, 64);
  const double carry = (lane > 0 && kprev == key[0]) ? tprev : 0.0;
  const int knext = __shfl_down(key[0], 1, 64);
  const int kfirst = headcont ? __shfl(key[0], 0, 64) : -2;
#pragma unroll
  for (int i = 0; i < SPL; ++i) {
    const int kn = (i + 1 < SPL) ? key[i + 1] : ((lane < 63) ? knext : -1);
    if (key[i] >= 0 && key[i] != kn) {
      const double tot = run[i] + ((key[i] == key[0]) ? carry : 0.0);
      const long long o = (long long)key[i] * K;
      if (key[i] == kfirst) {
        outB[o] = tot;                       // continuation: plane A was written by the previous chunk
      } else {
        outA[o] = tot;                       // (plane B of a segment no chunk boundary cuts is never read: ev.seg_cut)
      }
    }
  }
}

// The pair-by-pair bodies below are run by ONE wavefront with an LDS area of its own -- the single-wave block of k_spike, or
// a worker wave of a class kernel's block when the pass rides in that launch (pair_pass) -- so their synchronisation is a
// wave-level LDS fence, never a block barrier.
__device__ __forceinline__ void pair_sync() { lds_wave_fence(); }

// Scalar type of the pair loop: double (default) or float (precision mode "mixed", BASELINE config 5: fp32 ELBO terms with
// fp64 Kzz / Cholesky / whitening / KL).  In fp32 the exponential is ONE v_exp_f32 instead of the 14-instruction fp64
// kernel and every FMA issues at twice the fp64 rate: a (spike, inducing point) pair costs ~11 ns of SIMD time instead of
// ~44.  Phases, scaled times and embedding weights are formed in fp64 and rounded once; per-lane partial sums (8 spikes)
// and the 64-lane reduction run in fp32; everything that leaves the wave (P0 / P1 / T partials per chunk, dC per neuron)
// is fp64, and so is every cross-chunk / cross-trial sum downstream.
__device__ __forceinline__ float sv_exp2_neg(float y) { return __builtin_amdgcn_exp2f(y); }   // v_exp_f32; underflows to 0
__device__ __forceinline__ float sv_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double sv_fma(double a, double b, double c) { return fma(a, b, c); }

// Periodic kernel, value + all gradients.  With theta_s = 2 pi t_s / p and phi_j = 2 pi z_j / p every sum over spikes the
// pass needs is linear in sin / cos of theta_s, so the per-pair work is one exponential and seven FMAs:
//   y_sj = hh (cos(theta_s - phi_j) - 1) = ct_s (hh cos phi_j) + st_s (hh sin phi_j) - hh        ck = c_s 2^y
//   P0 = sum ck,  QA = sum ck st,  QB = sum ck ct,  QC = sum ck st t,  QD = sum ck ct t   and then per inducing point
//   sum ck sin(theta - phi)   = cos phi QA - sin phi QB            (-> direct dZ)
//   sum ck cos(theta - phi)   = cos phi QB + sin phi QA            (-> dtheta_lengthscale through sum ck y)
//   sum ck sin(theta - phi) t = cos phi QC - sin phi QD            (-> dtheta_period, minus z_j times the first)
// (no sin(theta - phi) per pair, no separate sum of ck y).  Five reduction vectors, reduced four inducing points at a time.
template <int SPL, typename T>
__device__ __forceinline__ void spike_body_periodic_full(const EngineView& ev, const double* __restrict__ C, int ch, int k, double* red_d) {
  constexpr int NV = 5, JG = 4;   // 20 rows: 13 KB of LDS per wave (fp64), so that three waves per SIMD fit
  const int lane = threadIdx.x & 63;
  const SpikeChunk c = ev.chunks[ch];
  const LatentMeta m = ev.meta[k];
  const int M = m.M, K = ev.K, r = c.trial;
  const KPar p = make_kpar(m, ev.theta + 2 * k);
  const double* __restrict__ z = ev.Zs + m.offM + (long long)r * M;
  const double* __restrict__ a = ev.a + m.offM + (long long)r * M;
  const double L2E = 1.4426950408889634074;
  const double sc = 4.0 * p.ip, hh = 0.5 * p.h * L2E;      // phase in quarter turns: (pi/2) (t 4/p)
  const double w2 = -1.0 / (2.0 * hh), w3 = 1.0 / sc;
  const T hhT = (T)hh;
  T* red = (T*)red_d;

  // (the neuron ids are re-read in the epilogue instead of living in VGPRs through the pair loop: 168 VGPRs = 3 waves / SIMD)
  T cw[SPL], mu[SPL], st[SPL], ct[SPL], stt[SPL], ctt[SPL];
#pragma unroll
  for (int i = 0; i < SPL; ++i) {
    int idx = lane * SPL + i;          // lane owns SPL consecutive spikes
    bool ok = idx < c.count;
    long long gs = c.start + (ok ? idx : 0);
    const double tsc = ev.spk_t[gs] * sc;
    cw[i] = ok ? (T)C[(long long)ev.spk_n[gs] * K + k] : (T)0;
    mu[i] = (T)0;
    double sd, cd;
    sv_sincos_quarter(tsc, sd, cd);
    st[i] = (T)sd;
    ct[i] = (T)cd;
    stt[i] = (T)(sd * tsc);
    ctt[i] = (T)(cd * tsc);
  }
  double* zsc = (double*)(red + NV * JG * 65 + ((NV * JG * 65) & 1));   // per inducing point (fp64): scaled z | sin | cos
  T* hsc = (T*)(zsc + 3 * 64);        // hh sin phi | hh cos phi | a_j   (pair-loop operands, scalar type T)
  T* rs = hsc + 3 * 64;               // NV * JG reduced sums of the current group
  if (lane < M) {
    const double zq = z[lane] * sc;
    double sz, cz;
    sv_sincos_quarter(zq, sz, cz);
    zsc[lane] = zq;
    zsc[64 + lane] = sz;
    zsc[128 + lane] = cz;
    hsc[lane] = (T)(hh * sz);
    hsc[64 + lane] = (T)(hh * cz);
    hsc[128 + lane] = (T)a[lane];
  }
  pair_sync();
  double t0acc = 0.0, t1acc = 0.0;

  for (int jb = 0; jb < M; jb += JG) {
#pragma unroll
    for (int jj = 0; jj < JG; ++jj) {
      const int j = jb + jj;
      T p0 = 0, qa = 0, qb = 0, qc = 0, qd = 0;
      if (j < M) {
        const T hs = hsc[j], hc = hsc[64 + j], aj = hsc[128 + j];
#pragma unroll
        for (int i = 0; i < SPL; ++i) {
          const T y = sv_fma(ct[i], hc, sv_fma(st[i], hs, -hhT));
          const T kt = sv_exp2_neg(y);               // y <= ~1e-16: the kernel only needs |y - rint(y)| <= 1/2
          const T ck = cw[i] * kt;
          p0 += ck;
          qa = sv_fma(ck, st[i], qa);
          qb = sv_fma(ck, ct[i], qb);
          qc = sv_fma(ck, stt[i], qc);
          qd = sv_fma(ck, ctt[i], qd);
          mu[i] = sv_fma(kt, aj, mu[i]);
        }
      }
      red[(0 * JG + jj) * 65 + lane] = p0;
      red[(1 * JG + jj) * 65 + lane] = qa;
      red[(2 * JG + jj) * 65 + lane] = qb;
      red[(3 * JG + jj) * 65 + lane] = qc;
      red[(4 * JG + jj) * 65 + lane] = qd;
    }
    pair_sync();
    // transposed reduction: lane (v, quarter) sums 16 of the 64 lane-partials of row v, then the quarters are added
#pragma unroll
    for (int v0 = 0; v0 < NV * JG; v0 += 16) {
      const int v = v0 + (lane & 15), qd4 = lane >> 4;
      T s = 0;
      if (v < NV * JG) {
#pragma unroll
        for (int e = 0; e < 16; ++e) s += red[v * 65 + qd4 * 16 + e];
      }
      s += __shfl_xor(s, 16, 64);
      s += __shfl_xor(s, 32, 64);
      if (lane < 16 && v < NV * JG) rs[v] = s;
    }
    pair_sync();
    if (lane < JG && jb + lane < M) {
      const int j = jb + lane;
      const double P0 = rs[lane], QA = rs[JG + lane], QB = rs[2 * JG + lane], QC = rs[3 * JG + lane],
                   QD = rs[4 * JG + lane];
      const double sz = zsc[64 + j], cz = zsc[128 + j];
      const double p1 = fma(cz, QA, -(sz * QB));              // sum ck sin(theta - phi_j)
      const double pcs = fma(cz, QB, sz * QA);                // sum ck cos(theta - phi_j)
      const double p3 = fma(cz, QC, -(sz * QD));              // sum ck sin(theta - phi_j) t (scaled)
      const long long o = (long long)ch * ev.sumM + m.colM + j;
      ev.P0[o] = P0;
      ev.P1[o] = p1;
      t0acc = fma(a[j], hh * (pcs - P0), t0acc);              // sum ck y
      t1acc = fma(a[j], p3 - zsc[j] * p1, t1acc);             // sum ck sin(.) (t - z_j)   (scaled units)
    }
    pair_sync();
  }

  const double t0 = wave_sum(t0acc) * w2;
  const double t1 = wave_sum(t1acc) * w3;
  if (lane == 0) {
    ev.Tpart[((long long)ch * K + k) * 2 + 0] = t0;
    ev.Tpart[((long long)ch * K + k) * 2 + 1] = t1;
  }
  // spike part of dC[n, k]: per-neuron sums of mu_s
  int nid[SPL];
#pragma unroll
  for (int i = 0; i < SPL; ++i) {
    const int idx = lane * SPL + i;
    nid[i] = (idx < c.count) ? ev.spk_n[c.start + idx] : -1;
  }
  double mud[SPL];
#pragma unroll
  for (int i = 0; i < SPL; ++i) mud[i] = (double)mu[i];
  if (c.split & 1) {
    double tot = 0.0;
#pragma unroll
    for (int i = 0; i < SPL; ++i) tot += (nid[i] >= 0) ? mud[i] : 0.0;
    tot = wave_sum(tot);
    if (lane == 0) ev.splitPart[(long long)ch * K + k] = p.s2 * tot;
  } else {
    const long long ob = (long long)r * ev.N * K + k;
#pragma unroll
    for (int i = 0; i < SPL; ++i) mud[i] *= p.s2;
    seg_emit_lane<SPL>(nid, mud, ev.gCspk + ob, ev.gCspkB + ob, (c.split & 2) != 0, (c.split & 4) != 0, K, lane);
  }
}

// Kernels as cosine series (used by the class path, kernels_spike_cls.h): ktil(delta) = sum_n c_n cos(n w delta).
//   periodic (engine_common.h periodic_series): w = 2 pi / p, c_n = (2) e^-beta I_n(beta), beta = 1 / l^2 -- exact
//       (Jacobi-Anger), truncated where n c_n < 1e-18; table per latent made at kernel-build time.
//   RBF: exp(-h delta^2), h = 1 / (2 l^2), on the bounded window W of the (trial, class) (all spike times of the trial and
//       all inducing points of the class's latents) equals its periodisation with period L >= |W| + sqrt(41.5 / h) up to
//       the wrapped copies exp(-h (L - |W|)^2) <= 1e-18, and the periodised Gaussian has the series (Poisson summation)
//       c_n = (2) sqrt(pi / h) / L exp(-pi^2 n^2 / (h L^2)),  w = 2 pi / L,  truncated where n c_n < 1e-18.
// With theta_s = w (t_s - t0), phi_j = w (z_j - t0) every sum over spikes the pass needs follows from the trigonometric
// moments of the spikes
//     A_n = sum_s c_s cos(n theta_s)   B_n = sum_s c_s sin(n theta_s)   [periodic: At_n, Bt_n = the same weighted by t_s]
// (with Cc_nj = cos(n phi_j) A_n + sin(n phi_j) B_n,  Ss_nj = cos(n phi_j) B_n - sin(n phi_j) A_n):
//     P0_j = sum_s c_s ktil        = sum_n c_n Cc_nj
//     P1_j = sum_s c_s ktil bz     = sum_n e_n Ss_nj     periodic bz = sin x:  e_n = n c_n / beta   (ktil sin x = -(1/beta) dktil/dx)
//                                                         RBF      bz = delta:  e_n = n w c_n / (2 h) (ktil delta = -(1/2h) dktil/ddelta)
//     T0   = sum_j a_j sum_s c_s ktil b0 = sum_j a_j sum_n g_n Cc_nj
//                                                         periodic b0 = sin^2(x/2): g_n = -1/2 dc_n/dbeta
//                                                         RBF      b0 = delta^2:    g_n = -dc_n/dh = c_n (1/(2h) - pi^2 n^2 / (h^2 L^2))
//     T1   (periodic only) = sum_j a_j sum_n e_n [cos(n phi_j) Bt_n - sin(n phi_j) At_n - z_j Ss_nj]
//     mu_s = s2 sum_j a_j ktil_sj  = s2 sum_n c_n [cos(n theta_s) Abar_n + sin(n theta_s) Bbar_n],  Abar_n = sum_j a_j cos(n phi_j)
// cos / sin of the multiples come from rotations by theta (error ~ n ulp), not from the three-term recurrence (~ n^2
// ulp).  Series values carry an ABSOLUTE error of ~1e-15 of the kernel's maximum (the terms are O(0.3) and cancel where
// the kernel is small); the sums they feed are dominated by their large terms.

template <int KT, bool FULL, int SPL, typename T, bool WC = true>
__device__ __forceinline__ void spike_body(const EngineView& ev, const double* __restrict__ C, int ch, int k,
                           double* red_d) {
  if constexpr (KT == SV_PERIODIC && FULL) {
    spike_body_periodic_full<SPL, T>(ev, C, ch, k, red_d);
    return;
  }
  // This body: RBF (value-only or full) and periodic value-only.  NV partial-sum vectors per inducing point: P0 | P0 P1 P2
  constexpr int NV = FULL ? 3 : 1;
  constexpr int JG = SV_SPIKE_JG;     // inducing points per reduction group
  const int lane = threadIdx.x & 63;
  const SpikeChunk c = ev.chunks[ch];
  const LatentMeta m = ev.meta[k];
  const int M = m.M, K = ev.K, r = c.trial;
  const KPar p = make_kpar(m, ev.theta + 2 * k);
  const double* __restrict__ z = ev.Zs + m.offM + (long long)r * M;
  const double* __restrict__ a = ev.a + m.offM + (long long)r * M;
  T* red = (T*)red_d;
  // Scaled coordinates so that ktil = 2^y with no further multiplies:
  //   RBF      u = (t - z) sc, sc = sqrt(h log2 e):  y = -u^2          bz = u / sc,  b0 = -y / sc^2
  //   periodic w = (t - z) 4/p:  (sn, cs) = sincos(pi/2 w), y = hh (cs - 1), hh = h log2(e) / 2
  //            bz = sn,  b0 = sin^2 = -y / (2 hh),  b1 = delta sn = (p/4) w sn
  const double L2E = 1.4426950408889634074;
  double sc, hh = 0.0, w1, w2, w3 = 0.0;
  if (KT == SV_RBF) {
    sc = sqrt(p.h * L2E);
    w1 = 1.0 / sc;
    w2 = -1.0 / (sc * sc);
  } else {
    sc = 4.0 * p.ip;
    hh = 0.5 * p.h * L2E;
    w1 = 1.0;
    w2 = -1.0 / (2.0 * hh);
    w3 = 1.0 / sc;
  }
  const T hhT = (T)hh;

  T ts[SPL], cw[SPL], mu[SPL];
  // periodic: sin / cos of the spike phase 2 pi t / p, once per spike; the pair (spike, inducing point)
  // then costs 4 FMAs by angle addition instead of a sincos
  T st[SPL], ct[SPL];
  int nid[SPL];
#pragma unroll
  for (int i = 0; i < SPL; ++i) {
    int idx = lane * SPL + i;          // lane owns SPL consecutive spikes
    bool ok = idx < c.count;
    long long gs = c.start + (ok ? idx : 0);
    const double tsd = ev.spk_t[gs] * sc;
    ts[i] = (T)tsd;
    nid[i] = ok ? ev.spk_n[gs] : -1;
    cw[i] = ok ? (T)C[(long long)nid[i] * K + k] : (T)0;
    mu[i] = (T)0;
    if (KT == SV_PERIODIC) {
      double sd, cd;
      sv_sincos_quarter(tsd, sd, cd);
      st[i] = (T)sd;
      ct[i] = (T)cd;
    }
  }
  T* zsc = red + NV * JG * 65;        // 4 * 64 values behind the reduction area: scaled z, sin / cos of its phase, a_j
  if (lane < M) {
    const double zq = z[lane] * sc;
    zsc[lane] = (T)zq;
    zsc[192 + lane] = (T)a[lane];
    if (KT == SV_PERIODIC) {
      double sz, cz;
      sv_sincos_quarter(zq, sz, cz);
      zsc[64 + lane] = (T)sz;
      zsc[128 + lane] = (T)cz;
    }
  }
  pair_sync();
  double t0acc = 0.0, t1acc = 0.0;

  for (int jb = 0; jb < M; jb += JG) {
#pragma unroll
    for (int jj = 0; jj < JG; ++jj) {
      const int j = jb + jj;
      T p0 = 0, p1 = 0, p2 = 0;
      if (j < M) {
        const T zs = zsc[j], aj = zsc[192 + j];
        T sz = 0, cz = 0;
        if (KT == SV_PERIODIC) {
          sz = zsc[64 + j];
          cz = zsc[128 + j];
        }
#pragma unroll
        for (int i = 0; i < SPL; ++i) {
          T y, u = 0;
          if (KT == SV_RBF) {
            u = ts[i] - zs;
            y = -(u * u);
          } else {
            const T cs = sv_fma(st[i], sz, ct[i] * cz);       // cos(2 pi (t - z) / p)
            y = sv_fma(cs, hhT, -hhT);
          }
          const T kt = sv_exp2_neg(y);
          if (FULL) {                                         // RBF only (periodic: spike_body_periodic_full)
            const T ck = cw[i] * kt;
            p0 += ck;
            p2 = sv_fma(ck, y, p2);
            p1 = sv_fma(ck, u, p1);
            mu[i] = sv_fma(kt, aj, mu[i]);
          } else {
            p0 = sv_fma(cw[i], kt, p0);
          }
        }
      }
      red[jj * 65 + lane] = p0;
      if (FULL) {
        red[(JG + jj) * 65 + lane] = p1;
        red[(2 * JG + jj) * 65 + lane] = p2;
      }
    }
    pair_sync();
    // transposed reduction: lane (v, quarter) sums 16 of the 64 lane-partials of vector v
#pragma unroll
    for (int v0 = 0; v0 < NV * JG; v0 += 16) {
      const int v = v0 + (lane & 15), qd = lane >> 4;
      T sT = 0;
      if (v < NV * JG) {
#pragma unroll
        for (int e = 0; e < 16; ++e) sT += red[v * 65 + qd * 16 + e];
      }
      sT += __shfl_xor(sT, 16, 64);
      sT += __shfl_xor(sT, 32, 64);
      const double s = (double)sT;
      const int j = jb + (v % JG);
      if (lane < 16 && v < NV * JG && j < M) {
        const long long o = (long long)ch * ev.sumM + m.colM + j;
        const int type = v / JG;
        if (type == 0)
          ev.P0[o] = s;
        else if (type == 1)
          ev.P1[o] = s * w1;
        else
          t0acc = fma(a[j], s, t0acc);
      }
    }
    pair_sync();
  }

  if (FULL) {
    const double t0 = wave_sum(t0acc) * w2;
    const double t1 = wave_sum(t1acc) * w3;
    if (lane == 0) {
      ev.Tpart[((long long)ch * K + k) * 2 + 0] = t0;
      ev.Tpart[((long long)ch * K + k) * 2 + 1] = t1;
    }
    // spike part of dC[n, k]: per-neuron sums of mu_s
    double mud[SPL];
#pragma unroll
    for (int i = 0; i < SPL; ++i) mud[i] = (double)mu[i];
    if (c.split & 1) {
      double tot = 0.0;
#pragma unroll
      for (int i = 0; i < SPL; ++i) tot += (nid[i] >= 0) ? mud[i] : 0.0;
      tot = wave_sum(tot);
      if (lane == 0) ev.splitPart[(long long)ch * K + k] = p.s2 * tot;
    } else {
      const long long ob = (long long)r * ev.N * K + k;
#pragma unroll
      for (int i = 0; i < SPL; ++i) mud[i] *= p.s2;
      seg_emit_lane<SPL>(nid, mud, ev.gCspk + ob, ev.gCspkB + ob, (c.split & 2) != 0, (c.split & 4) != 0, K, lane);
    }
  }
}

// LDS of one wavefront of the pass (doubles): RBF full 3 x 8 rows + 4 x 64 (14.5 KB in fp64), periodic full 5 x 4 rows + ~7 x 64
// (14 KB), value-only 8 rows; half of that in fp32
__host__ __device__ constexpr int sv_pair_lds_doubles(bool full) {
  const int rbf = 3 * SV_SPIKE_JG * 65 + 4 * 64, per = 5 * 4 * 65 + 7 * 64 + 32;
  return full ? (rbf > per ? rbf : per) : SV_SPIKE_JG * 65 + 4 * 64;
}

// The pass itself: worker `worker` of `nworkers` single-wave workers strides over the (chunk, latent) items the class path
// does not serve.  red: sv_pair_lds_doubles(FULL) doubles of LDS of this wave's own.
// Work item = (chunk, latent).  Workers b and b + 8 share an XCD (and its L2) under the dispatcher's round-robin placement
// when they are single-wave blocks of a grid that is a multiple of 8 -- a speed assumption only -- so the items are
// re-indexed such that the workers of one XCD walk a contiguous range of chunk-major items: the K workers that read the same
// chunk's spike times / neuron ids run on one XCD, back to back, and all but the first find them in its L2 (the item starts
// with a chain of dependent global loads).
template <bool FULL, typename T, bool WC>
__device__ __forceinline__ void pair_pass(const EngineView& ev, const double* __restrict__ C, double* red, int worker,
                                          int nworkers) {
  constexpr int SPL = SV_SPL;
  const int lane = threadIdx.x & 63;
  {   // every item served by the class path (the usual case): leave at once
    int any = 0;
    for (int r = lane; r < ev.R; r += 64) any |= ev.pair_any[r];
    if (!__any(any)) return;
  }
  const int nper = ev.ntype[SV_PERIODIC], nrbf = ev.ntype[SV_RBF], nlat = nper + nrbf;
  const long long nb = (long long)ev.nchunks * nlat;
  for (long long w = worker; w < nb; w += nworkers) {
    const long long x = w & 7, q = nb >> 3, rem = nb & 7;
    const long long item = ((x < rem) ? x * (q + 1) : rem * (q + 1) + (x - rem) * q) + (w >> 3);   // bijective for any nb
    const int ch = (int)(item / nlat), slot = (int)(item % nlat);
    const int k = (slot < nper) ? ev.klist[SV_PERIODIC][slot] : ev.klist[SV_RBF][slot - nper];
    if (ev.ser[ev.chunks[ch].trial * ev.K + k]) continue;        // served by the class path
    if (slot < nper) spike_body<SV_PERIODIC, FULL, SPL, T, WC>(ev, C, ch, k, red);
    else spike_body<SV_RBF, FULL, SPL, T, WC>(ev, C, ch, k, red);
    pair_sync();                                                 // LDS is reused by the next item
  }
}

