// Spike-time pass, PAIR-BY-PAIR form: the path of the (trial, latent) items whose kernel the class path
// (kernels_spike_cls.h) cannot serve through a cosine series -- short lengthscales against the trial's window, periodic
// kernels with beta = 1 / l^2 > 45, very few inducing points -- and of every RBF latent under SVGPFA_RBF_SERIES=0.
// ev.ser[trial][latent] says which items belong to the class path; they are skipped here.
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
#include "spike_pair_common.h"

// One launch over all (chunk, latent) work items; the covariance type is resolved per item.  Two waves / SIMD: at the cap
// of three (168 registers) the fp64 FULL variants spilled 42 registers and ran a third slower (0.97 against 0.71 ms for
// the whole cfg3 pass pair by pair).  Which items remain for this pass is only known on the device (ev.ser), so the grid
// is a fixed number of single-wave blocks (a multiple of 8) that stride over the items and skip the class path's.
// WC = false: no embedding gradient wanted (the pair loops compute the spike means anyway).
// When the class path runs as well (fp64) the pass does not get a launch of its own: it rides behind the class items of the
// class kernel's workers (kernels_spike_cls.h) -- an evaluation whose items are all served by the class path, the usual
// case, used to pay 4.5-5 us for a launch that found nothing to do.
template <bool FULL, typename T = double, bool WC = true>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 8))) void k_spike(EngineView ev,
                                                                                         const double* __restrict__ C) {
  __shared__ double red[sv_pair_lds_doubles(FULL)];
  pair_pass<FULL, T, WC>(ev, C, red, blockIdx.x, gridDim.x);
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
// = sum_r (gCspk [+ gCspkB where the segment is cut: sv_dc_spike])[r][n][k] + the split chunks of neuron n, trials four at a time, fixed order.
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
        v[u] = (r < R) ? sv_dc_spike(ev, r, id / ev.K, id % ev.K) : 0.0;
      }
      s += (v[0] + v[1]) + (v[2] + v[3]);
    }
    if (pg == 0) s += split_sums(ev, id / ev.K, id % ev.K);
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

