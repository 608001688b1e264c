// Spike-time pass, class path: the kernels' cosine series (kernels_spike.h header: periodic = Jacobi-Anger, RBF = the
// periodised Gaussian over the trial's window) turn every sum over spikes the ELBO needs into linear functions of
// trigonometric moments of the spikes.  The weight of a spike in those moments is c_s = C[n_s, k] -- a function of the
// spike's NEURON only -- so the moments factor through the UNWEIGHTED per-(trial, neuron) sums
//     U_c[n][nu] = sum_{s in (trial, neuron nu)} cos(n theta_s),   U_s[n][nu] = ... sin,   (periodic: t_s cos, t_s sin too)
//     A_n^(k) = sum_nu C[nu, k] U_c[n][nu]                      (a [K_c x N] . [N x NV (H + 1)] product per trial: f64 MFMA)
//     dC[nu, k] += s2 sum_n c_n (Abar_n U_c[n][nu] + Bbar_n U_s[n][nu])
// and U depends on the data and the angular frequency only: all latents that share a frequency (a CLASS) share it.
//   * periodic latents with the same period form a class;
//   * RBF latents form classes by lengthscale (within a factor 2 of the class's longest): the periodisation identity holds
//     for every period L >= |W| + sqrt(41.5 / h), so the class uses ONE L per trial (from its longest lengthscale and the
//     window over all its members' inducing points); a member's series then runs to its own H_k (~ 1.5 L / l_k).
// A spike then costs 6 fp64 instructions per harmonic and CLASS (8 with the time-weighted moments) instead of 8..10 per
// harmonic and LATENT (round 2's spike_body_series), and there is no per-spike embedding gather at all.
//
// Steps (in this order, on the evaluation's one stream):
//   class table   class table; per (trial, class) window / period / harmonics; ser[r][k] -- rider blocks of k_build_reg
//                 (spike_cls_common.h; k_spike_classes when the spikes arrive after the kernel build)
//   W             (r, k): W = s2 c_n (Abar_n, Bbar_n): the posterior-mean weights a projected on the harmonics (only when
//                 dC is wanted; needs a = Kzz^-1 m) -- a rider of k_prep's wave 0 (k_spike_pre in the statistics calls)
//   k_spike_cls   (tile, class): U for the tile's <= 64 rows (lane = row = one (trial, neuron) segment or half of one,
//                 harmonics in registers), A-moment partials by f64 MFMA through an LDS transpose, dC rows; both kinds of
//                 class (with / without time-weighted moments) in one kernel, k_spike_cls_all (round 5: at every size;
//                 SVGPFA_CLS_FUSED=0: one kernel per kind, k_spike_cls<4> then k_spike_cls<2>)
//   k_spike_comb  (r, k): sums the tiles' partials (fixed order) and combines them per inducing point: P0, P1, T of the
//                 item -- what k_item_bwd's gather reads.  On small shards, where every launch is one under-filled round,
//                 the gather does this combine itself instead (ev.comb_fold; kernels_item_bwd.h gather_item_inputs): one
//                 launch less, at the price of ~2.5 us per k_item_bwd block (which is why the full-size problem keeps the
//                 separate kernel: 262 -> 288 us against 11.5 us measured)
// (trial, latent) items the series does not serve (short lengthscales, very few inducing points) keep the pair-by-pair
// pass of kernels_spike.h; ser[r][k] says which.
// Replaces the reference's K*R Python loop of kernelsMatricesStore.py:208-221, svPosteriorOnLatents.py:265-300 and
// svEmbedding.py:137-144; Ktz at the spikes is never materialised.
#pragma once
#include "engine_common.h"
#include "kernels_mfma.h"
#include "spike_cls_common.h"
#include "spike_pair_common.h"

// The pair-by-pair pass as a RIDER of a class kernel's workers (fp64): behind its class items worker `worker` strides over
// the (chunk, latent) items the class path does not serve (pair_pass; it leaves at once when there are none, the usual
// case).  The pass used to be a launch of its own behind the class kernels: 4.5-5 us per evaluation for finding nothing.
// DC / full select the variant exactly as launch_pair_pass does for k_spike; lds: sv_pair_lds_doubles(full) doubles.
template <int DC>
__device__ __forceinline__ void cls_pair_rider(const EngineView& ev, const double* __restrict__ C, int full, double* lds,
                                               int worker, int nworkers) {
  full &= 1;
  if (DC == 1) pair_pass<true, double, true>(ev, C, lds, worker, nworkers);        // (DC == 1 implies full and dC wanted)
  else if (full) pair_pass<true, double, false>(ev, C, lds, worker, nworkers);
  else pair_pass<false, double, true>(ev, C, lds, worker, nworkers);
}
// doubles of LDS of a single-wave class block: the transpose tile and the dC rows, or the rider's area
template <int NV, int DC>
__host__ __device__ constexpr int sv_cls_block_lds() {
  const int own = 64 * 16 + (DC == 1 ? SV_CLS_MAXMEM * 64 : 1), ride = (NV == 2) ? sv_pair_lds_doubles(true) : 0;
  return own > ride ? own : ride;
}

// ------------------------------------------------------------------------------------------------ k_spike_classes
// The class table and the per-trial entries for the engine's kernel SNAPSHOT, when the spikes arrived after the kernel
// build (else the class blocks of k_build_reg made them): grid = ceil(R / 4) blocks of 256 threads, every block rebuilds
// the (tiny) table in LDS, one wavefront per trial then fills the per-trial entries.
__global__ __launch_bounds__(256) void k_spike_classes(EngineView ev) {
  __shared__ SpikeClassScratch cs;
  const int wave = threadIdx.x >> 6, R = ev.R;
  const int r = blockIdx.x * 4 + wave;
  if (wave == 0) spike_class_table(ev, cs, nullptr, nullptr);
  __syncthreads();
  if (blockIdx.x == 0 && wave == 1) spike_class_store(ev, cs);
  if (r < R) spike_class_trial(ev, cs, nullptr, r);
}

// ------------------------------------------------------------------------------------------------ k_spike_pre
// W of every (trial, latent) item (spike_pre_item) when k_prep_mfma did not make it (statistics calls).  grid = R K blocks
// of 64 threads.
__global__ __launch_bounds__(64) void k_spike_pre(EngineView ev) {
  __shared__ double scratch[SV_PRE_LDS];
  const int item = blockIdx.x, lane = threadIdx.x, K = ev.K;
  const int r = item / K, k = item % K;
  const LatentMeta m = ev.meta[k];
  const double aj = (lane < m.M) ? ev.a[m.offM + (long long)r * m.M + lane] : 0.0;
  spike_pre_item(ev, item, r, k, m, aj, scratch);
}

// ------------------------------------------------------------------------------------------------ k_spike_cls
// One register pass over the rows' spikes: harmonics h0 .. h0 + HB - 1 of every row (lane) of the tile.
//   acc[NV n + 0] = sum_s cos((h0 + n) theta_s)   [+1] = sin   ([+2], [+3]: the same weighted by t_s, NV = 4)
// (cos, sin) of the multiples by the three-term recurrence; a later pass starts from direct sincos of (h0 - 1) theta, h0 theta.
template <int NV, int HB>
__device__ __forceinline__ void cls_pass(const double* __restrict__ tt, int maxlen, int len, int lane, double t0,
                                         double sc, int h0, double (&acc)[NV * HB]) {
#pragma unroll
  for (int c = 0; c < NV * HB; ++c) acc[c] = 0.0;
  // transposed times: row = lane, spike i at [i][lane]; four spikes' times are fetched a group ahead (one spike costs
  // ~0.4 us of issue time: a single load in flight did not cover the memory latency at two waves per SIMD).  The loads
  // are unconditional -- a read past the tile's end lands in the next tile or the host's pad (SV_CLS_TT_PAD) --
  // so that the four are issued back to back and waited for once (behind a bounds check the compiler waited per load).
  constexpr int PF = SV_CLS_PF;
  double tn[PF];
#pragma unroll
  for (int u = 0; u < PF; ++u) tn[u] = tt[(long long)u * 64 + lane];
#if defined(SV_CLS_DIAG) && SV_CLS_DIAG == 2     // diagnosis build: no pass over the spikes (results are wrong)
  maxlen = 0;
#endif
  for (int i0 = 0; i0 < maxlen; i0 += PF) {
    double tc[PF];
#pragma unroll
    for (int u = 0; u < PF; ++u) tc[u] = tn[u];
#pragma unroll
    for (int u = 0; u < PF; ++u) tn[u] = tt[(long long)(i0 + PF + u) * 64 + lane];
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      if (i0 + u < len) {
        const double t = tc[u];
        const double ph = (t - t0) * sc;
        // (cos, sin) of the multiples of theta by the three-term recurrence  f((n + 1) theta) = 2 cos(theta) f(n theta) -
        // f((n - 1) theta): ONE FMA per value (the rotation by theta it replaces: two multiplies and two FMAs per pair --
        // with the two accumulations 4 instead of 6 instructions per spike and harmonic).  Its rounding error grows like
        // n^2 u in the worst case (theta near a multiple of pi) against n u for the rotation: <= 64^2 u = 5e-13 at the
        // largest harmonic this path admits, on terms that enter the sums with the series' fast-decaying coefficients.
        double s1, c1, cn = 1.0, sn = 0.0, cp, sp;
        sv_sincos_quarter_sc(ph, s1, c1);
        cp = c1;                                         // cos(-theta), sin(-theta): the values "before" harmonic 0
        sp = -s1;
        if (h0 > 0) {
          // (a real branch on the wave-uniform h0: left to itself the compiler evaluates both sincos for every spike and
          // selects -- 60 of the ~210 instructions a spike costs in the first pass, which at the north-star shape is the
          // only pass; round 4, read off the device assembly.  The empty asm cannot be speculated, and the arguments hang on it.)
          double phh = ph;
          asm volatile("" : "+v"(phh));
          sv_sincos_quarter_sc(phh * (double)h0, sn, cn);
          sv_sincos_quarter_sc(phh * (double)(h0 - 1), sp, cp);
        }
        const double k2 = c1 + c1;
#pragma unroll
        for (int n = 0; n < HB; ++n) {
          acc[NV * n + 0] += cn;
          acc[NV * n + 1] += sn;
          if (NV == 4) {
            acc[NV * n + 2] = fma(t, cn, acc[NV * n + 2]);
            acc[NV * n + 3] = fma(t, sn, acc[NV * n + 3]);
          }
          const double cq = fma(k2, cn, -cp), sq = fma(k2, sn, -sp);
          cp = cn;
          sp = sn;
          cn = cq;
          sn = sq;
        }
      }
    }
  }
}

// What follows a pass.  The rows' sums go, 16 columns at a time, through a (swizzled, conflict-free) LDS tile Ut[row][16]
// so that the tile's partial A-moments  Apart[member][col] = sum_rows C[neuron(row), member] U[row][col]  come out of 16
// v_mfma_f64_16x16x4_f64 (A operand: the gathered embedding weights cg, B operand: U); and, DC, the rows' own dC partials
// dcl[member][row] += sum_cols U[row][col] W[member][col]  on the VALU (W is wave-uniform).
// Order (round 4; diagnosis builds had shown the flush to take as long as the pass over the spikes, 16 of an item's 36 us):
// every LOAD of the flush -- W, then the embedding weights -- is issued and consumed before its first STORE to global
// memory, and the A-moment tiles stay in registers until the end: a wait for a load is then never a wait for an earlier
// store as well (one counter serves both on this chip).  The dC rows come first, while all accumulators are live anyway;
// their sums run as two independent FMA chains per member (cos / sin columns) instead of one of 2 HB links.
template <int NV, int HB, int DC>
__device__ __forceinline__ void cls_flush(const EngineView& ev, const double* __restrict__ C, const double (&acc)[NV * HB],
                                          int neuron, int nrows, int memk, int nmem, int tile, int g, int r, int h0, int H,
                                          double* Ut, double* dcl) {
  const int lane = threadIdx.x & 63, li = lane & 15, lg = lane >> 4, K = ev.K;
  const int ncol = NV * (H + 1);           // valid columns of the class's moment rows
#if defined(SV_CLS_DIAG) && SV_CLS_DIAG == 1     // diagnosis build: no flush (results are wrong), the pass alone is timed
#pragma unroll
  for (int c = 0; c < NV * HB; ++c) asm volatile("" ::"v"(acc[c]));
  return;
#endif
  if (DC == 1) {
    // W of the members for this pass's harmonics, staged through the (idle) transpose tile: [member][2 HB], entries
    // beyond the last harmonic are zeros (spike_pre_item).  Coalesced loads, all in flight, then broadcast reads.
    constexpr int WP = 2 * HB, NWV = (SV_CLS_MAXMEM * WP + 63) / 64;      // doubles per member and pass; per lane at most
    const int tot = nmem * WP;
    double wv[NWV];
#pragma unroll
    for (int u = 0; u < NWV; ++u) {
      wv[u] = 0.0;
      if (u * 64 < tot) {                  // wave-uniform: only the loads the class's members need
        const int e = u * 64 + lane, sm = (e < tot) ? e / WP : 0, c = (e < tot) ? e % WP : 0;
        const int mk = __shfl(memk, sm, 64);
        const int col = 2 * h0 + c;
        wv[u] = ev.Wb[((long long)r * K + mk) * SV_WCOLS + (col < SV_WCOLS ? col : 0)];
        if (col >= SV_WCOLS) wv[u] = 0.0;
      }
    }
#pragma unroll
    for (int u = 0; u < NWV; ++u)
      if (u * 64 + lane < tot) Ut[u * 64 + lane] = wv[u];
    lds_wave_fence();
    for (int s = 0; s < nmem; ++s) {
      const double* Wk = Ut + s * WP;
      double v0 = 0.0, v1 = 0.0;
#pragma unroll
      for (int n = 0; n < HB; ++n) {
        v0 = fma(acc[NV * n + 0], Wk[2 * n + 0], v0);
        v1 = fma(acc[NV * n + 1], Wk[2 * n + 1], v1);
      }
      const double v = v0 + v1;
      dcl[s * 64 + lane] = (h0 == 0) ? v : dcl[s * 64 + lane] + v;
    }
    lds_wave_fence();                      // the tile is free for the transposes
  }
  // embedding weights in A-operand layout: cg[i] = C[neuron(row 4 i + lg)][member li]  (unconditional loads from valid
  // addresses, then masked: padding rows carry neuron 0).  Fetched behind the pass, so that they do not occupy 32
  // registers through it -- and behind the FIRST chunk's transposed store (round 4: the store frees that chunk's 32
  // accumulator registers, which the 16 weights then take; requested in front of it, accumulators, weights and the products'
  // tiles were live together and 10-25 registers went to the stack); C (N K doubles) stays in L1 / L2.
  double cg[16];
  auto load_cg = [&]() {
    const int mk = __shfl(memk, li < nmem ? li : 0, 64);
    int nu[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) nu[i] = __shfl(neuron, 4 * i + lg, 64);
#pragma unroll
    for (int i = 0; i < 16; ++i) cg[i] = C[(long long)nu[i] * K + mk];          // sixteen independent loads in flight
#pragma unroll
    for (int i = 0; i < 16; ++i) cg[i] = (4 * i + lg < nrows && li < nmem) ? cg[i] : 0.0;
  };
  constexpr int NCH = NV * HB / 16;
  d4 D[NCH];
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch) {
    const int c0 = 16 * ch;
    D[ch] = (d4){0.0, 0.0, 0.0, 0.0};
    if (ch == 0 || NV * h0 + c0 < ncol) {  // wave-uniform: whole chunks beyond the last harmonic are skipped (chunk 0 never is)
#pragma unroll
      for (int c = 0; c < 16; ++c) Ut[lane * 16 + (c ^ li)] = acc[c0 + c];
      if (ch == 0) {
        __builtin_amdgcn_sched_barrier(0);
        load_cg();
      }
      lds_wave_fence();
      d4 Dc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int row = 4 * i + lg;
        Dc = __builtin_amdgcn_mfma_f64_16x16x4f64(cg[i], Ut[row * 16 + (li ^ (row & 15))], Dc, 0, 0, 0);
      }
      D[ch] = Dc;
      lds_wave_fence();                    // the tile is free for the next chunk
    }
  }
  // ---- stores: the tile's A-moment partials (f64 accumulator layout: register q holds row (lane >> 4) + 4 q = member)
  int mkq[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) mkq[q] = __shfl(memk, (lg + 4 * q < nmem) ? lg + 4 * q : 0, 64);
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch) {
    const int col = NV * h0 + 16 * ch + li;
    if (col < ncol) {
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (lg + 4 * q < nmem) ev.Apart[((long long)tile * K + mkq[q]) * SV_ACOLS + col] = D[ch][q];
    }
  }
}

// the rows' dC partials: a whole segment writes plane A, the halves of a cut segment one plane each (ev.seg_cut, static,
// tells the reader which segments have a second half: engine_common.h sv_dc_spike); pieces of an oversize segment go to
// their slot
__device__ __forceinline__ void cls_emit_rows(const EngineView& ev, const SpikeRow& row, int r, int memk, int nmem,
                                              const double* dcl) {
  const int lane = threadIdx.x & 63, K = ev.K;
  if (row.len <= 0) return;
  for (int s = 0; s < nmem; ++s) {
    const int k = __shfl(memk, s, 64);
    const double v = dcl[s * 64 + lane];
    const long long o = ((long long)r * ev.N + row.neuron) * K + k;
    if (row.kind == 0) {
      ev.gCspk[o] = v;                     // (plane B of an uncut segment is never read: ev.seg_cut)
    } else if (row.kind == 1) {
      ev.gCspk[o] = v;
    } else if (row.kind == 2) {
      ev.gCspkB[o] = v;
    } else {
      ev.splitS[(long long)row.slot * K + k] = v;
    }
  }
}

template <int NV, int DC>
__device__ __forceinline__ void cls_item(const EngineView& ev, const double* __restrict__ C, const SpikeClassTab* cls,
                                         int g, int ord, double* Ut, double* dcl) {
  constexpr int HBM = (NV == 2) ? 24 : 16;          // largest register pass: 48 / 64 accumulators
  const int lane = threadIdx.x & 63, K = ev.K, R = ev.R;      // (a wave of a larger block in k_ell_cls)
  const SpikeTile tl = ev.tiles_ord[ord];           // tiles in cost order (a copy: one dependent load less than through an index)
  const int r = tl.trial, tile = tl.id;
  // everything the item needs that hangs off (tile, class), requested together
  const int H = ev.cls_H[(long long)g * R + r];
  const double sc = ev.cls_sc[(long long)g * R + r], t0 = ev.cls_t0[(long long)g * R + r];
  const int nmem = cls->nmem[g];
  const int memk = cls->mem[g][lane & (SV_CLS_MAXMEM - 1)];     // member list, one entry per lane (read by shuffles)
  SpikeRow row = ev.rows[tl.row0 + (lane < tl.nrows ? lane : 0)];
  if (lane >= tl.nrows) row = SpikeRow{0, 0, 0, 0};
  if (H < 0) return;                                 // this (trial, class) takes the pair loop
  const double* __restrict__ tt = ev.tt + tl.tt_off;
  for (int h0 = 0; h0 <= H; h0 += HBM) {
    const int left = H + 1 - h0;
    if (left > 16 && NV == 2) {
      double acc[NV * 24];
      cls_pass<NV, 24>(tt, tl.maxlen, row.len, lane, t0, sc, h0, acc);
      cls_flush<NV, 24, DC>(ev, C, acc, row.neuron, tl.nrows, memk, nmem, tile, g, r, h0, H, Ut, dcl);
    } else if (left > 8) {
      double acc[NV * 16];
      cls_pass<NV, 16>(tt, tl.maxlen, row.len, lane, t0, sc, h0, acc);
      cls_flush<NV, 16, DC>(ev, C, acc, row.neuron, tl.nrows, memk, nmem, tile, g, r, h0, H, Ut, dcl);
    } else {
      double acc[NV * 8];
      cls_pass<NV, 8>(tt, tl.maxlen, row.len, lane, t0, sc, h0, acc);
      cls_flush<NV, 8, DC>(ev, C, acc, row.neuron, tl.nrows, memk, nmem, tile, g, r, h0, H, Ut, dcl);
    }
  }
  if (DC == 1) cls_emit_rows(ev, row, r, memk, nmem, dcl);
}

// Work item = (tile, class); the number of classes is only known on the device, so the grid is a fixed number of
// single-wave blocks that stride over the items (tiles in the host's cost order, longest first).
// NV = 4: the classes that carry the time-weighted moments (periodic classes when Z / theta gradients are wanted: `full`);
// NV = 2: all others.  Two kernels rather than one: the NV = 2 pass needs a third fewer registers.
// DC: 1 = the rows' dC partials wanted (needs W).
// pair: the pair-by-pair pass rides behind the class items (NV = 2 launches only: that kernel runs in every evaluation).
template <int NV, int DC>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 8))) void k_spike_cls(EngineView ev,
                                                                                             const double* __restrict__ C,
                                                                                             int full, int pair) {
  __shared__ double lds[sv_cls_block_lds<NV, DC>()];
  double* Ut = lds;
  double* dcl = lds + 64 * 16;
  const SpikeClassTab* cls = ev.cls;
  const int ncls = cls->ncls;
  int mine = 0;                                  // classes this kernel serves (scalar loop, ncls <= 32)
  // full: bit 0 = Z / theta / C gradient sums wanted, bit 1 = time-weighted moments wanted (kernel-parameter gradient): only
  // then do periodic classes take the NV = 4 pass
  const int tw = (full >> 1) & 1;
  for (int g = 0; g < ncls; ++g) mine += ((tw && cls->type[g] == SV_PERIODIC) == (NV == 4)) ? 1 : 0;
  const int nitems = ev.ntiles * mine;           // ntiles K < 2^31 is checked by the host
  for (int w = blockIdx.x; w < nitems; w += gridDim.x) {
    int j = w % mine, g = 0;
    for (;; ++g) {
      if ((tw && cls->type[g] == SV_PERIODIC) == (NV == 4)) {
        if (j == 0) break;
        --j;
      }
    }
    cls_item<NV, DC>(ev, C, cls, g, w / mine, Ut, dcl);
  }
  if (NV == 2 && pair) {
    lds_wave_fence();
    cls_pair_rider<DC>(ev, C, full, lds, blockIdx.x, gridDim.x);
  }
}

// Both kinds of class in ONE launch (the default at every size, svgpfa_engine::cls_fused): work item = (tile, class)
// over ALL classes, the item's class decides the pass it takes.  One launch boundary (a drained chip) less.
template <int DC>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 8))) void k_spike_cls_all(EngineView ev,
                                                                                                 const double* __restrict__ C,
                                                                                                 int full, int pair) {
  __shared__ double lds[sv_cls_block_lds<2, DC>()];
  double* Ut = lds;
  double* dcl = lds + 64 * 16;
  const SpikeClassTab* cls = ev.cls;
  const int ncls = cls->ncls;
  const int nitems = ev.ntiles * ncls;
  for (int w = blockIdx.x; w < nitems; w += gridDim.x) {
    const int g = w % ncls;
    if ((full & 2) && cls->type[g] == SV_PERIODIC) cls_item<4, DC>(ev, C, cls, g, w / ncls, Ut, dcl);
    else cls_item<2, DC>(ev, C, cls, g, w / ncls, Ut, dcl);
  }
  if (pair) {
    lds_wave_fence();
    cls_pair_rider<DC>(ev, C, full, lds, blockIdx.x, gridDim.x);
  }
}

// ------------------------------------------------------------------------------------------------ k_ell_cls
// Small shards (up to 4 096 items, fp64): the quadrature term's two roles (kernels_mfma.h quad_ell_mfma_body) AND the class
// items of the spike pass in ONE launch -- blocks [0, nell) are the ELL blocks of k_quad_ell_mfma<KS, 2>, every wave of the
// blocks behind them is a class worker of k_spike_cls_all (its own slice of the block's LDS; workers stride over the
// (tile, class) items).  Neither depends on the other (both follow k_front, both precede k_item_bwd), each is ONE
// under-filled round of the chip on a shard (8-way shard of cfg3: 504 blocks of four waves for 24 us, ~1 000 single-wave
// items for 25 us), and two kernels share the chip best inside one launch: 49 -> ~30 us.
// grid = nell + ceil(workers / 4) blocks of 256 threads; dynamic LDS = max(ELL roles, 4 worker slices).
#define SV_CLS_SLICE (64 * 16 + SV_CLS_MAXMEM * 64)       // doubles of LDS per class worker: transpose tile + dC rows
template <int KS, int DC>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_ell_cls(EngineView ev,
                                                                                             const double* __restrict__ C,
                                                                                             const double* d, int want_sums,
                                                                                             int want_lat, int parts, int lc0,
                                                                                             int lc1, int full, int nell, int pair) {
  extern __shared__ double sm[];
  // (ELL blocks first.  The class workers' blocks first -- they are the long ones -- was measured in round 4: 8-way shard
  // 42.4 against 43.7 us, 4-way 68 against 55: the ELL blocks then start behind the workers on the same SIMDs.)
  const int nwb = (int)gridDim.x - nell, bid = (int)blockIdx.x - nell;
  if (bid < 0) {
    // (latent adjoints without dC / dd -- the E-step closure: role-1 blocks only, they write the ELL value: kernels_mfma.h)
    const int nb = (want_lat && !want_sums) ? 0 : ev.R * parts, blk = (int)blockIdx.x;
    if (blk < nb) quad_ell_mfma_body<KS, 0, double>(ev, C, d, want_sums, 1, parts, lc0, blk, sm);
    else quad_ell_mfma_body<KS, 1, double>(ev, C, d, 1, want_sums ? 0 : 1, parts, lc1, blk - nb, sm);
    return;
  }
  const int wave = threadIdx.x >> 6;
  double* Ut = sm + wave * SV_CLS_SLICE;
  double* dcl = Ut + 64 * 16;
  const SpikeClassTab* cls = ev.cls;
  const int ncls = cls->ncls;
  const int nitems = ev.ntiles * ncls;
  const int nwork = nwb * 4;
  for (int w = bid * 4 + wave; w < nitems; w += nwork) {
    const int g = w % ncls;
    if ((full & 2) && cls->type[g] == SV_PERIODIC) cls_item<4, DC>(ev, C, cls, g, w / ncls, Ut, dcl);
    else cls_item<2, DC>(ev, C, cls, g, w / ncls, Ut, dcl);
  }
  if (pair) {                  // (the worker's slice holds sv_pair_lds_doubles(true) doubles: static_assert below)
    lds_wave_fence();
    cls_pair_rider<DC>(ev, C, full, Ut, bid * 4 + wave, nwork);
  }
}
static_assert(SV_CLS_SLICE >= sv_pair_lds_doubles(true), "a class worker's LDS slice must hold the pair-by-pair rider's area");

// ------------------------------------------------------------------------------------------------ k_spike_comb
// Per (trial, latent): the class's moments A_n, B_n (At_n, Bt_n) = the trial's tile partials added in tile order, then per
// inducing point j (phase phi_j, harmonics by rotation), with Cc = cos(n phi_j) A_n + sin(n phi_j) B_n and
// Ss = cos(n phi_j) B_n - sin(n phi_j) A_n (kernels_spike.h header):
//   P0_j = sum_n c_n Cc,  P1_j = sum_n e_n Ss,  T0 = sum_j a_j sum_n g_n Cc,
//   T1 (periodic) = sum_j a_j sum_n e_n [cos(n phi_j) Bt_n - sin(n phi_j) At_n - z_j Ss]
// -> P0t / P1t [r][sum M], Tt [r][k][2]: what k_item_bwd's gather reads for this item.  grid = R K blocks of 64 threads.
template <bool FULL>
__global__ __launch_bounds__(64) void k_spike_comb(EngineView ev, int tw, int pairs) {
  __shared__ double lds[2 * SV_COMB_LDS];
  if (pairs) spike_comb_pair<FULL>(ev, 2 * (int)blockIdx.x, ev.R * ev.K, lds, tw);      // (M <= 32: two items per wave)
  else spike_comb_item<FULL>(ev, blockIdx.x, lds, tw);
}
