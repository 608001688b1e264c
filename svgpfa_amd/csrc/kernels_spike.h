// Spike-time pass: the dominant kernel (sum_k R M_k S_r kernel evaluations per ELBO).
//
// One wavefront per (spike chunk, latent).  Each lane owns SPL (= 8) consecutive spikes of the chunk (times and
// embedding weights c_s = C[n_s, k] in registers); the inducing points z_j and a_j = (Kzz^-1 m)_j
// are wave-uniform.  For the exponential link the spike term of the ELL needs only the posterior
// MEAN, whose adjoint is known a priori (d ELL / d mu_k(t_s) = C[n_s, k]), so value and all
// gradients come out of ONE pass over kappa(t_s, z_j):
//   P0_j = sum_s c_s ktil_sj            -> g_a_j = s2 P0_j, ELL_spk = a . g_a
//   P1_j = sum_s c_s ktil_sj bz_sj      -> direct dZ_j = a_j cz P1_j
//   T_t  = sum_s c_s sum_j a_j ktil_sj bt_sj -> direct dtheta_t = c_t T_t
//   mu_s = s2 sum_j ktil_sj a_j         -> dC[n, k] += sum_{s in n} mu_s, emitted per neuron by an
//                                          in-wave segmented scan (spikes are neuron-sorted)
// Replaces the reference's K*R Python loop of kernelsMatricesStore.py:208-221,
// svPosteriorOnLatents.py:265-300 and svEmbedding.py:137-144; Ktz is never materialised.
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
// out2 != nullptr: the chunk's FIRST segment continues a segment of the previous chunk; its sum goes to out2.
template <int SPL>
__device__ __forceinline__ void seg_emit_lane(const int (&key)[SPL], const double (&v)[SPL], double* out, double* out2,
                                              int K, int lane) {
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
  const int kfirst = out2 ? __shfl(key[0], 0, 64) : -2;
#pragma unroll
  for (int i = 0; i < SPL; ++i) {
    const int kn = (i + 1 < SPL) ? key[i + 1] : ((lane < 63) ? knext : -1);
    if (key[i] >= 0 && key[i] != kn)
      ((key[i] == kfirst) ? out2 : out)[(long long)key[i] * K] = run[i] + ((key[i] == key[0]) ? carry : 0.0);
  }
}

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
__device__ void spike_body_periodic_full(const EngineView& ev, const double* __restrict__ C, int ch, int k, double* red_d) {
  constexpr int NV = 5, JG = 4;   // 20 rows: 13 KB of LDS per wave (fp64), so that three waves per SIMD fit
  const int lane = threadIdx.x;
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
  __syncthreads();
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
    __syncthreads();
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
    __syncthreads();
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
    __syncthreads();
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
    seg_emit_lane<SPL>(nid, mud, ev.gCspk + ob, (c.split & 2) ? ev.gCspkB + ob : nullptr, K, lane);
  }
}

// Kernels as cosine series: ktil(delta) = sum_n c_n cos(n w delta).
//   periodic (engine_common.h periodic_series): w = 2 pi / p, c_n = (2) e^-beta I_n(beta), beta = 1 / l^2 -- exact
//       (Jacobi-Anger), truncated where n c_n < 1e-18; table per latent made at kernel-build time.
//   RBF: exp(-h delta^2), h = 1 / (2 l^2), on the bounded window W of the item (all spike times of the trial and all its
//       inducing points) equals its periodisation with period L = |W| + sqrt(41.5 / h) up to the wrapped copies
//       exp(-h (L - |W|)^2) <= 1e-18, and the periodised Gaussian has the series (Poisson summation)
//       c_n = (2) sqrt(pi / h) / L exp(-pi^2 n^2 / (h L^2)),  w = 2 pi / L,  truncated where n c_n < 1e-18.
//       L and the number of harmonics are chosen per (trial, latent) at run time from the data and the current
//       lengthscale (H ~ 4.3 L / l: 19 at the north-star shape); short lengthscales fall back to the pair loop.
// With theta_s = w (t_s - t0), phi_j = w (z_j - t0) every sum over spikes the pass needs follows from the trigonometric
// moments of the chunk's spikes
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
// A spike costs 8 (RBF) / 10 (periodic) fp64 operations per harmonic (2 / 4 for the moments, 2 for mu, 4 to rotate
// (cos, sin) to the next harmonic) instead of M exponentials: 14 and 20 harmonics at the north-star shape against
// 32 x 21..23 instructions.  cos / sin of the multiples come from rotations by theta (error ~ n ulp), not from the
// three-term recurrence (~ n^2 ulp).  Series values carry an ABSOLUTE error of ~1e-15 of the kernel's maximum (the
// terms are O(0.3) and cancel where the kernel is small); the sums they feed are dominated by their large terms.
struct SeriesPar {
  int H;               // harmonics 0 .. H
  double sc, t0;       // phase in quarter turns = (t - t0) sc
  const double* c;     // c_n
  const double* g;     // T0 weights
  const double* e;     // P1 weights
};

// WC = false (only with FULL): the caller wants no embedding gradient (kernel-parameter / inducing-point closures of the
// EM loop): the posterior mean at the spikes and its per-neuron sums are left out -- the projections Abar / Bbar, two of
// the 8..10 operations per (spike, harmonic) and the segmented emit.
template <int KT, bool FULL, int SPL, bool WC = true>
__device__ void spike_body_series(const EngineView& ev, const double* __restrict__ C, int ch, int k, double* red,
                                  const SeriesPar sp) {
  constexpr bool TW = FULL && KT == SV_PERIODIC;      // time-weighted moments (period gradient)
  constexpr bool MU = FULL && WC;                     // posterior mean at the spikes -> dC
  constexpr int ROWS = FULL ? 16 : 8;                 // LDS reduction rows per group (value-only: half the LDS, more waves)
  constexpr int NV = TW ? 4 : 2, JG = ROWS / NV;
  const int lane = threadIdx.x, H = sp.H;
  const SpikeChunk c = ev.chunks[ch];
  const LatentMeta m = ev.meta[k];
  const int M = m.M, K = ev.K, r = c.trial;
  const KPar p = make_kpar(m, ev.theta + 2 * k);
  const double* __restrict__ z = ev.Zs + m.offM + (long long)r * M;
  const double* __restrict__ a = ev.a + m.offM + (long long)r * M;
  const double sc = sp.sc;
  double* Fs = red + ROWS * 65;                  // [SV_FH + 1][NV] moments of the chunk
  double* AB = Fs + (SV_FH + 1) * 4;             // [SV_FH + 1][2] c_n Abar_n, c_n Bbar_n   (FULL)

  double cw[SPL], tw[TW ? SPL : 1], c1[SPL], s1[SPL], cn[SPL], sn[SPL], mu[MU ? SPL : 1];
#pragma unroll
  for (int i = 0; i < SPL; ++i) {
    const int idx = lane * SPL + i;              // lane owns SPL consecutive spikes
    const bool ok = idx < c.count;
    const long long gs = c.start + (ok ? idx : 0);
    const double t = ev.spk_t[gs];
    cw[i] = ok ? C[(long long)ev.spk_n[gs] * K + k] : 0.0;
    if constexpr (TW) tw[i] = cw[i] * t;
    sv_sincos_quarter((t - sp.t0) * sc, s1[i], c1[i]);
    cn[i] = 1.0;
    sn[i] = 0.0;
    if constexpr (MU) mu[i] = 0.0;
  }
  double zj = 0.0, aj = 0.0, sphi = 0.0, cphi = 1.0;
  if (lane < M) {
    zj = z[lane];
    aj = a[lane];
    sv_sincos_quarter((zj - sp.t0) * sc, sphi, cphi);
  }
  // transposed reduction of ROWS LDS rows of 64 lane-partials: lane (v, quarter) sums 16 entries, the quarters are
  // added by shuffles, lanes < ROWS hold the totals
  auto reduce_rows = [&]() -> double {
    const int v = lane & 15, qd = lane >> 4;
    double sum = 0.0;
    if (v < ROWS) {
#pragma unroll
      for (int e = 0; e < 16; ++e) sum += red[v * 65 + qd * 16 + e];
    }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    return sum;
  };
  if constexpr (MU) {
    // Abar_n, Bbar_n (times c_n): the posterior-mean weights a_j projected on the harmonics of the inducing points
    double cj = 1.0, sj = 0.0;
    for (int nb = 0; nb <= H; nb += 8) {
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) {
        red[(2 * jj + 0) * 65 + lane] = aj * cj;
        red[(2 * jj + 1) * 65 + lane] = aj * sj;
        const double tq = cj * cphi - sj * sphi;
        sj = fma(sj, cphi, cj * sphi);
        cj = tq;
      }
      __syncthreads();
      const double tot = reduce_rows();
      const int n = nb + (lane >> 1);
      if (lane < 16 && n <= H) AB[2 * n + (lane & 1)] = sp.c[n] * tot;
      __syncthreads();
    }
  }

  for (int nb = 0; nb <= H; nb += JG) {
#pragma unroll
    for (int jj = 0; jj < JG; ++jj) {
      const int n = nb + jj;
      double sa = 0.0, sb = 0.0, sat = 0.0, sbt = 0.0;
      if (n <= H) {
        double cA = 0.0, cB = 0.0;
        if constexpr (MU) {
          cA = AB[2 * n];
          cB = AB[2 * n + 1];
        }
#pragma unroll
        for (int i = 0; i < SPL; ++i) {
          sa = fma(cw[i], cn[i], sa);
          sb = fma(cw[i], sn[i], sb);
          if constexpr (TW) {
            sat = fma(tw[i], cn[i], sat);
            sbt = fma(tw[i], sn[i], sbt);
          }
          if constexpr (MU) mu[i] = fma(cn[i], cA, fma(sn[i], cB, mu[i]));
          const double tq = cn[i] * c1[i] - sn[i] * s1[i];     // (cos, sin) of (n + 1) theta: rotation by theta
          sn[i] = fma(sn[i], c1[i], cn[i] * s1[i]);
          cn[i] = tq;
        }
      }
      red[(NV * jj + 0) * 65 + lane] = sa;
      red[(NV * jj + 1) * 65 + lane] = sb;
      if constexpr (TW) {
        red[(NV * jj + 2) * 65 + lane] = sat;
        red[(NV * jj + 3) * 65 + lane] = sbt;
      }
    }
    __syncthreads();
    const double tot = reduce_rows();
    if (lane < ROWS && nb + lane / NV <= H) Fs[nb * NV + lane] = tot;
    __syncthreads();
  }

  // per inducing point: combine the moments with the harmonics of its own phase
  double t0acc = 0.0, t1acc = 0.0;
  if (lane < M) {
    double cj = 1.0, sj = 0.0, P0 = 0.0, P1 = 0.0, d0 = 0.0, d1 = 0.0;
    for (int n = 0; n <= H; ++n) {
      const double An = Fs[n * NV + 0], Bn = Fs[n * NV + 1];
      const double Cc = fma(cj, An, sj * Bn), Ss = fma(cj, Bn, -(sj * An));
      P0 = fma(sp.c[n], Cc, P0);
      if constexpr (FULL) {
        const double en = sp.e[n];
        P1 = fma(en, Ss, P1);
        d0 = fma(sp.g[n], Cc, d0);
        if constexpr (TW) d1 = fma(en, fma(cj, Fs[n * NV + 3], -(sj * Fs[n * NV + 2])) - zj * Ss, d1);
      }
      const double tq = cj * cphi - sj * sphi;
      sj = fma(sj, cphi, cj * sphi);
      cj = tq;
    }
    const long long o = (long long)ch * ev.sumM + m.colM + lane;
    ev.P0[o] = P0;
    if constexpr (FULL) {
      ev.P1[o] = P1;
      t0acc = aj * d0;
      t1acc = aj * d1;
    }
  }
  if constexpr (FULL) {
    const double t0 = wave_sum(t0acc), t1 = wave_sum(t1acc);
    if (lane == 0) {
      ev.Tpart[((long long)ch * K + k) * 2 + 0] = t0;
      ev.Tpart[((long long)ch * K + k) * 2 + 1] = t1;
    }
  }
  if constexpr (MU) {
    // spike part of dC[n, k]: per-neuron sums of mu_s
    int nid[SPL];
#pragma unroll
    for (int i = 0; i < SPL; ++i) {
      const int idx = lane * SPL + i;
      nid[i] = (idx < c.count) ? ev.spk_n[c.start + idx] : -1;
    }
    if (c.split & 1) {
      double tot = 0.0;
#pragma unroll
      for (int i = 0; i < SPL; ++i) tot += (nid[i] >= 0) ? mu[i] : 0.0;
      tot = wave_sum(tot);
      if (lane == 0) ev.splitPart[(long long)ch * K + k] = p.s2 * tot;
    } else {
      const long long ob = (long long)r * ev.N * K + k;
#pragma unroll
      for (int i = 0; i < SPL; ++i) mu[i] *= p.s2;
      seg_emit_lane<SPL>(nid, mu, ev.gCspk + ob, (c.split & 2) ? ev.gCspkB + ob : nullptr, K, lane);
    }
  }
}

// RBF: window, period and series of this (trial, latent) item (see above).  Returns H, or -1 for the pair loop.
// cf: LDS, 3 x (SV_FH + 1) doubles (c_n | g_n | e_n).
__device__ __forceinline__ int rbf_series_setup(const EngineView& ev, const LatentMeta& m, const KPar& p, int r,
                                                double* cf, double& sc, double& t0) {
  const int lane = threadIdx.x, M = m.M;
  const double* z = ev.Zs + m.offM + (long long)r * M;
  double lo = ev.spk_lo[r], hi = ev.spk_hi[r];
  if (lane < M) {
    const double zj = z[lane];
    lo = fmin(lo, zj);
    hi = fmax(hi, zj);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    lo = fmin(lo, __shfl_xor(lo, o, 64));
    hi = fmax(hi, __shfl_xor(hi, o, 64));
  }
  const double PI = 3.14159265358979323846;
  const double h = p.h;                                  // exp(-h delta^2)
  const double L = (hi - lo) + sqrt(41.5 / h);           // wrapped copies below e^-41.5 = 1e-18
  const double q = PI * PI / (h * L * L);                // c_n ~ exp(-q n^2)
  const double nmax = sqrt(46.0 / q);                    // n c_n / c_0 < 64 e^-46 = 7e-19 beyond
  if (!(nmax < (double)SV_FH)) return -1;
  const int H = (int)nmax + 1;
  if (H > SV_FH || 10 * (H + 2) > 21 * M) return -1;     // per-spike cost of the series against M kernel evaluations
  const double w = 2.0 * PI / L, c0 = sqrt(PI / h) / L;
  __syncthreads();                                       // the previous user of this LDS region is done
  for (int n = lane; n <= H; n += 64) {
    const double cn = (n ? 2.0 : 1.0) * c0 * exp(-q * (double)n * (double)n);
    cf[n] = cn;
    cf[(SV_FH + 1) + n] = cn * (0.5 / h - q * (double)n * (double)n / h);
    cf[2 * (SV_FH + 1) + n] = (double)n * w * cn * (0.5 / h);
  }
  __syncthreads();
  sc = 4.0 / L;
  t0 = lo;
  return H;
}

template <int KT, bool FULL, int SPL, typename T, bool WC = true>
__device__ void spike_body(const EngineView& ev, const double* __restrict__ C, int ch, int k,
                           double* red_d) {
  {   // the kernel through its cosine series where that is short enough.  The series bodies are fp64 in both precision
      // modes: rotations accumulate ~n ulp, too much in fp32, and they beat the fp32 pair loop anyway; precision mode
      // "mixed" uses fp32 in the pair loop, i.e. where the series does not apply
    constexpr int NF = SV_FH + 2;
    if constexpr (KT == SV_PERIODIC) {
      const int H = ev.fourH[k];             // made by k_build_reg (periodic_series)
      if (H >= 0) {
        const double* tab = ev.four + (size_t)k * SV_FROWS * NF;
        const KPar p = make_kpar(ev.meta[k], ev.theta + 2 * k);
        spike_body_series<SV_PERIODIC, FULL, SPL, WC>(ev, C, ch, k, red_d,
                                                  SeriesPar{H, 4.0 * p.ip, 0.0, tab, tab + NF, tab + 2 * NF});
        return;
      }
    } else if (ev.rbf_series) {
      const LatentMeta ms = ev.meta[k];
      const KPar p = make_kpar(ms, ev.theta + 2 * k);
      double* cf = red_d + (FULL ? 16 : 8) * 65 + (SV_FH + 1) * 6;   // behind the series body's own LDS
      double sc, t0;
      const int H = rbf_series_setup(ev, ms, p, ev.chunks[ch].trial, cf, sc, t0);
      if (H >= 0) {
        spike_body_series<SV_RBF, FULL, SPL, WC>(ev, C, ch, k, red_d,
                                             SeriesPar{H, sc, t0, cf, cf + (SV_FH + 1), cf + 2 * (SV_FH + 1)});
        return;
      }
    }
  }
  if constexpr (KT == SV_PERIODIC && FULL) {
    spike_body_periodic_full<SPL, T>(ev, C, ch, k, red_d);
    return;
  }
  // This body: RBF (value-only or full) and periodic value-only.  NV partial-sum vectors per inducing point: P0 | P0 P1 P2
  constexpr int NV = FULL ? 3 : 1;
  constexpr int JG = SV_SPIKE_JG;     // inducing points per reduction group
  const int lane = threadIdx.x;
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
  __syncthreads();
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
    __syncthreads();
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
    __syncthreads();
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
      seg_emit_lane<SPL>(nid, mud, ev.gCspk + ob, (c.split & 2) ? ev.gCspkB + ob : nullptr, K, lane);
    }
  }
}

// One launch over all (chunk, latent) work items; the covariance type is resolved per block.  Registers are capped so
// that 3 waves / SIMD stay resident.
// WC = false: no embedding gradient wanted (series bodies skip the spike means; the pair-loop fallbacks compute them anyway).
template <bool FULL, int SPL, typename T = double, bool WC = true>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3, 8))) void k_spike(EngineView ev,
                                                                                         const double* __restrict__ C) {
  // LDS bounds the waves per CU: RBF full 3 x 8 rows + 4 x 64 (14.5 KB in fp64), periodic full 5 x 4 rows + ~7 x 64 (14 KB),
  // value-only 8 rows; half of that in fp32
  constexpr int LDS_RBF = 3 * SV_SPIKE_JG * 65 + 4 * 64, LDS_PER = 5 * 4 * 65 + 7 * 64 + 32;
  constexpr int LDS_SER = (FULL ? 16 : 8) * 65 + (SV_FH + 1) * 9 + 8;   // cosine-series body: rows, moments + Abar / Bbar, RBF coefficients
  constexpr int LDS_FULL = LDS_RBF > LDS_PER ? (LDS_RBF > LDS_SER ? LDS_RBF : LDS_SER) : (LDS_PER > LDS_SER ? LDS_PER : LDS_SER);
  constexpr int LDS_VAL = SV_SPIKE_JG * 65 + 4 * 64 > LDS_SER ? SV_SPIKE_JG * 65 + 4 * 64 : LDS_SER;
  __shared__ double red[FULL ? LDS_FULL : LDS_VAL];
  // Work item = (chunk, latent).  Blocks b and b + 8 share an XCD (and its L2) under the dispatcher's round-robin
  // placement -- a speed assumption only -- so the grid is re-indexed such that the blocks of one XCD walk a contiguous
  // range of chunk-major items: the K blocks that read the same chunk's spike times / neuron ids run on one XCD, back to
  // back, and all but the first find them in its L2 (the item starts with a chain of dependent global loads).
  const int nper = ev.ntype[SV_PERIODIC], nrbf = ev.ntype[SV_RBF], nlat = nper + nrbf;
  const int nb = gridDim.x, b = blockIdx.x, x = b & 7, q = nb >> 3, rem = nb & 7;
  const int item = ((x < rem) ? x * (q + 1) : rem * (q + 1) + (x - rem) * q) + (b >> 3);   // bijective for any nb
  const int ch = item / nlat, slot = item % nlat;
  if (slot < nper) {
    spike_body<SV_PERIODIC, FULL, SPL, T, WC>(ev, C, ch, ev.klist[SV_PERIODIC][slot], red);
  } else {
    spike_body<SV_RBF, FULL, SPL, T, WC>(ev, C, ch, ev.klist[SV_RBF][slot - nper], red);
  }
}

#define SV_FS_OUT 8   // outputs per block of the two fixed-order cross-trial reductions below

// Mseg[n][k] = sum over all local spikes of neuron n of mu_k(t_s)   (embedding M-step path, once per
// computeSVPosteriorOnLatentsStats).  Block = SV_FS_OUT consecutive (n, k) outputs x 32 lanes per output, the lanes laid
// out as (32 / gs) trial groups x gs segment lanes (a segment = the consecutive spikes of neuron n in trial r; gs = a
// power of two chosen by the host from the mean segment length): lane (tr, sl) adds the elements sl, sl + gs, ... of the
// segments of trials tr, tr + 32 / gs, ...; the 32 partial sums are added in order through LDS.  (One thread per output
// walking every spike of its neuron: 4.7 ms at the north-star size.)
__global__ __launch_bounds__(256) void k_mseg(EngineView ev, int gs) {
  __shared__ double red[256];
  const int tid = threadIdx.x, o = tid % SV_FS_OUT, pg = tid / SV_FS_OUT, NG = 256 / SV_FS_OUT;
  const int id = blockIdx.x * SV_FS_OUT + o, NK = ev.N * ev.K;
  const int tr = pg / gs, sl = pg % gs, ntr = NG / gs;
  double s = 0.0;
  if (id < NK) {
    const int n = id / ev.K, k = id % ev.K;
    const double* mu = ev.muspk + (long long)k * ev.S;
    for (int r = tr; r < ev.R; r += ntr) {
      const long long rn = (long long)r * ev.N + n;
      const long long i0 = ev.seg_off[rn], i1 = ev.seg_off[rn + 1];
      double sr = 0.0;
      for (long long i = i0 + sl; i < i1; i += gs) sr += mu[i];
      s += sr;
    }
  }
  red[pg * SV_FS_OUT + o] = s;
  __syncthreads();
  if (pg == 0 && id < NK) {
    double t = red[o];
    for (int q = 1; q < NG; ++q) t += red[q * SV_FS_OUT + o];
    ev.Mseg[id] = t;
  }
}

// The same sums from the spike part of dC that a FULL spike pass has just left per trial (svgpfa_freeze_stats): Mseg[n][k]
// = sum_r (gCspk + gCspkB)[r][n][k] + the split chunks of neuron n, trials four at a time, fixed order.
__global__ __launch_bounds__(256) void k_mseg_from_dc(EngineView ev) {
  __shared__ double red[256];
  const int tid = threadIdx.x, o = tid % SV_FS_OUT, pg = tid / SV_FS_OUT, NG = 256 / SV_FS_OUT;
  const int id = blockIdx.x * SV_FS_OUT + o, NK = ev.N * ev.K, R = ev.R;
  double s = 0.0;
  if (id < NK) {
    for (int r0 = pg; r0 < R; r0 += 4 * NG) {
      double v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int r = r0 + u * NG;
        v[u] = (r < R) ? ev.gCspk[(long long)r * NK + id] + ev.gCspkB[(long long)r * NK + id] : 0.0;
      }
      s += (v[0] + v[1]) + (v[2] + v[3]);
    }
    if (pg == 0) {
      const int n = id / ev.K, k = id % ev.K;
      for (int l = 0; l < ev.nsplit; ++l) {
        const int ch = ev.split_chunks[l];
        if (ev.chunks[ch].neuron == n) s += ev.splitPart[(long long)ch * ev.K + k];
      }
    }
  }
  red[pg * SV_FS_OUT + o] = s;
  __syncthreads();
  if (pg == 0 && id < NK) {
    double t = red[o];
    for (int q = 1; q < NG; ++q) t += red[q * SV_FS_OUT + o];
    ev.Mseg[id] = t;
  }
}

// ELL with frozen latent stats (embedding M-step closure): out = [ELL, dC, dd].
// Block g < nbo sums SV_FS_OUT consecutive outputs over the per-trial partial rows: thread (o, pg) adds the trials pg,
// pg + NG, ... four at a time (eight independent loads in flight; one thread per output walking all R rows left ten
// blocks waiting on memory 500 times in a row), the NG partial sums are added in order through LDS.  Last block: ELL.
__global__ __launch_bounds__(256) void k_finalize_stats(EngineView ev, const double* C, const double* d,
                                                        int want_grad, double* out) {
  __shared__ double red[256];
  const int tid = threadIdx.x;
  const int NK = ev.N * ev.K, N = ev.N, K = ev.K;
  if (blockIdx.x == gridDim.x - 1) {
    double e = 0.0;
    for (int i = tid; i < ev.nell; i += 256) e += ev.ellq[i];
    for (int i = tid; i < NK; i += 256) e += C[i] * ev.Mseg[i];
    for (int n = tid; n < N; n += 256) e += ev.spk_cnt[n] * d[n];
    e = block_sum(e, red);
    if (tid == 0) out[0] = e;
    return;
  }
  if (!want_grad) return;
  const int o = tid % SV_FS_OUT, pg = tid / SV_FS_OUT, NG = 256 / SV_FS_OUT;
  const int id = blockIdx.x * SV_FS_OUT + o;
  double s = 0.0;
  if (id < NK) {
    const int n = id / K, k = id % K;
    const double cnk = C[id];
    for (int b0 = pg; b0 < ev.ncd; b0 += 4 * NG) {
      double v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int b = b0 + u * NG;
        double t = 0.0;
        if (b < ev.ncd) {
          const long long o2 = ((long long)b * N + n) * 2 * K + k;
          t = ev.gCpart[o2] + cnk * ev.gCpart[o2 + K];
        }
        v[u] = t;
      }
      s += (v[0] + v[1]) + (v[2] + v[3]);
    }
    s = -s;
  } else if (id < NK + N) {
    const int n = id - NK;
    for (int b0 = pg; b0 < ev.ncd; b0 += 4 * NG) {
      double v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int b = b0 + u * NG;
        v[u] = (b < ev.ncd) ? ev.gdpart[(long long)b * N + n] : 0.0;
      }
      s += (v[0] + v[1]) + (v[2] + v[3]);
    }
  }
  red[pg * SV_FS_OUT + o] = s;
  __syncthreads();
  if (pg == 0 && id < NK + N) {
    double t = red[o];
    for (int q = 1; q < NG; ++q) t += red[q * SV_FS_OUT + o];
    out[1 + id] = t + (id < NK ? ev.Mseg[id] : ev.spk_cnt[id - NK]);
  }
}

