// Vector kernels of the EM loop's optimiser (svgpfa_amd/stats/shardedLBFGS.py, flat mode on the GPU).
//
// torch.optim.LBFGS -- the optimiser the reference drives, stats/svEM.py:266-294 -- spends an iteration in ~60 small
// tensor operations (clone, add_, mul, dot per history pair, abs().max(), ...).  For the per-trial parameter groups
// of the EM loop that is either launch-latency bound (inducing-point locations) or a stream of whole-vector passes
// over HBM (E-step group: 22 MB per vector at the north-star size, ~100 passes per iteration).  The three entry
// points below do an iteration's vector work in the minimum number of passes:
//   svgpfa_lbfgs_grad_stats  after a closure evaluation: g = sign * g_raw (the copy the optimiser needs anyway),
//                            g.d, g.g, |g|_1, max|g| in the same pass                      (2 reads, 1 write)
//   svgpfa_lbfgs_pair_dots   start of an iteration: y = g - g_prev and s = t d written into the history, and all
//                            5 m + 5 inner products of the vector-free two-loop recursion   (2 m + 3 reads, 2 writes)
//   svgpfa_lbfgs_direction   d = -H g + Y^T cY + S^T cS, x0 = x, max|d|                     (2 m + 2 reads, 2 writes)
// Every sum has a fixed order (block-contiguous slices, partials added in block order by a second small launch): results
// are bit-reproducible.  Entries [0, ns) of a vector carry weight w0 in the sums (replicated parameters count once
// across ranks: w0 = 0 on every rank but the first).  Stateless: plain device pointers, a stream, a caller workspace.
#include "../../include/svgpfa.h"

#include <hip/hip_runtime.h>

#include "engine_common.h"

#define SV_LB_T 256   // threads per block
#define SV_LB_E 8     // elements per thread: a block owns SV_LB_T * SV_LB_E consecutive elements

namespace {

__device__ __forceinline__ double nan_max(double a, double b) { return (a != a || b != b) ? (a + b) : fmax(a, b); }

__device__ __forceinline__ double wave_nan_max(double x) {
  for (int o = 32; o > 0; o >>= 1) x = nan_max(x, __shfl_xor(x, o, 64));
  return x;
}

// sums of NS values and maxima of NM values over the block; thread 0 writes them to part[o * nb + block]
template <int NS, int NM>
__device__ __forceinline__ void block_out(const double (&s)[NS], const double (&mx)[NM > 0 ? NM : 1], double* part, int o0,
                                          int nb, double* red) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = SV_LB_T >> 6;
  __syncthreads();
#pragma unroll
  for (int i = 0; i < NS; ++i) {
    const double v = wave_sum(s[i]);
    if (lane == 0) red[w * (NS + NM) + i] = v;
  }
#pragma unroll
  for (int i = 0; i < NM; ++i) {
    const double v = wave_nan_max(mx[i]);
    if (lane == 0) red[w * (NS + NM) + NS + i] = v;
  }
  __syncthreads();
  if ((int)threadIdx.x < NS + NM) {
    const int i = threadIdx.x;
    double v = red[i];
    for (int q = 1; q < nw; ++q) v = (i < NS) ? v + red[q * (NS + NM) + i] : nan_max(v, red[q * (NS + NM) + i]);
    part[(long long)(o0 + i) * nb + blockIdx.x] = v;
  }
}

// g = sign * graw; partial [g.d, g.g, |g|_1] (weighted) and max|g|
__global__ __launch_bounds__(SV_LB_T) void k_lbfgs_grad_stats(long long n, const double* __restrict__ graw, double sign,
                                                              const double* __restrict__ d, double* __restrict__ g,
                                                              long long ns, double w0, double* part, int nb) {
  __shared__ double red[4 * 4];
  const long long base = (long long)blockIdx.x * (SV_LB_T * SV_LB_E) + threadIdx.x;
  double s[3] = {0.0, 0.0, 0.0}, mx[1] = {0.0};
  double v[SV_LB_E], dv[SV_LB_E];
#pragma unroll
  for (int u = 0; u < SV_LB_E; ++u) {
    const long long i = base + (long long)u * SV_LB_T;
    v[u] = (i < n) ? graw[i] : 0.0;
    dv[u] = (d != nullptr && i < n) ? d[i] : 0.0;
  }
#pragma unroll
  for (int u = 0; u < SV_LB_E; ++u) {
    const long long i = base + (long long)u * SV_LB_T;
    if (i < n) {
      const double x = sign * v[u];
      g[i] = x;
      const double xw = (i < ns) ? x * w0 : x;
      s[0] = fma(xw, dv[u], s[0]);
      s[1] = fma(xw, x, s[1]);
      s[2] += fabs(xw);
      mx[0] = nan_max(mx[0], fabs(x));
    }
  }
  block_out<3, 1>(s, mx, part, 0, nb, red);
}

// y = g - pg -> Y[m], s = t d -> S[m]; partials of [y.s, y.y, g.g, s.g, y.g] and, for every history row i < m,
// [S_i.y, Y_i.s, Y_i.y, S_i.g, Y_i.g] (weighted), laid out as the optimiser's coefficient code expects:
// output 5 + j m + i for j = 0..4
__global__ __launch_bounds__(SV_LB_T) void k_lbfgs_pair_dots(long long n, const double* __restrict__ g,
                                                             const double* __restrict__ pg, const double* __restrict__ d,
                                                             double t, double* S, double* Y, long long ld, int m, long long ns,
                                                             double w0, double* part, int nb) {
  __shared__ double red[4 * 5];
  const long long base = (long long)blockIdx.x * (SV_LB_T * SV_LB_E) + threadIdx.x;
  double yv[SV_LB_E], sv[SV_LB_E], gv[SV_LB_E];   // y, s unweighted; g weighted below
  double wv[SV_LB_E];
#pragma unroll
  for (int u = 0; u < SV_LB_E; ++u) {
    const long long i = base + (long long)u * SV_LB_T;
    const bool in = i < n;
    const double gi = in ? g[i] : 0.0, pi = in ? pg[i] : 0.0, di = in ? d[i] : 0.0;
    yv[u] = gi - pi;
    sv[u] = di * t;
    gv[u] = gi;
    wv[u] = (i < ns) ? w0 : 1.0;
    if (in) {
      Y[(long long)m * ld + i] = yv[u];
      S[(long long)m * ld + i] = sv[u];
    }
  }
  {
    double h[5] = {0.0, 0.0, 0.0, 0.0, 0.0}, none[1] = {0.0};
#pragma unroll
    for (int u = 0; u < SV_LB_E; ++u) {
      const double yw = yv[u] * wv[u], gw = gv[u] * wv[u], sw = sv[u] * wv[u];
      h[0] = fma(yw, sv[u], h[0]);
      h[1] = fma(yw, yv[u], h[1]);
      h[2] = fma(gw, gv[u], h[2]);
      h[3] = fma(sw, gv[u], h[3]);
      h[4] = fma(yw, gv[u], h[4]);
    }
    block_out<5, 0>(h, none, part, 0, nb, red);
  }
  for (int r = 0; r < m; ++r) {
    double Sv[SV_LB_E], Yv[SV_LB_E];
#pragma unroll
    for (int u = 0; u < SV_LB_E; ++u) {
      const long long i = base + (long long)u * SV_LB_T;
      const bool in = i < n;
      Sv[u] = in ? S[(long long)r * ld + i] : 0.0;
      Yv[u] = in ? Y[(long long)r * ld + i] : 0.0;
    }
    double h[5] = {0.0, 0.0, 0.0, 0.0, 0.0}, none[1] = {0.0};
#pragma unroll
    for (int u = 0; u < SV_LB_E; ++u) {
      const double yw = yv[u] * wv[u], gw = gv[u] * wv[u], sw = sv[u] * wv[u];
      h[0] = fma(Sv[u], yw, h[0]);
      h[1] = fma(Yv[u], sw, h[1]);
      h[2] = fma(Yv[u], yw, h[2]);
      h[3] = fma(Sv[u], gw, h[3]);
      h[4] = fma(Yv[u], gw, h[4]);
    }
    // outputs 5 + j m + r: written one by one (block_out's layout is consecutive outputs)
    {
      const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = SV_LB_T >> 6;
      __syncthreads();
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        const double v = wave_sum(h[j]);
        if (lane == 0) red[w * 5 + j] = v;
      }
      __syncthreads();
      if (threadIdx.x < 5) {
        const int j = threadIdx.x;
        double v = red[j];
        for (int q = 1; q < nw; ++q) v += red[q * 5 + j];
        part[(long long)(5 + j * m + r) * nb + blockIdx.x] = v;
      }
    }
  }
}

// d = -H g + sum_i cY_i Y_i + sum_i cS_i S_i (coef = [cY (m), cS (m), H]); x0 = x; partial max|d|
__global__ __launch_bounds__(SV_LB_T) void k_lbfgs_direction(long long n, const double* __restrict__ g,
                                                             const double* __restrict__ S, const double* __restrict__ Y,
                                                             long long ld, int m, const double* __restrict__ coef,
                                                             double* __restrict__ d, const double* __restrict__ x,
                                                             double* __restrict__ x0, double* part, int nb) {
  __shared__ double red[4];
  const long long base = (long long)blockIdx.x * (SV_LB_T * SV_LB_E) + threadIdx.x;
  const double H = coef[2 * m];
  double acc[SV_LB_E];
#pragma unroll
  for (int u = 0; u < SV_LB_E; ++u) {
    const long long i = base + (long long)u * SV_LB_T;
    acc[u] = (i < n) ? -H * g[i] : 0.0;
    if (i < n && x0 != nullptr) x0[i] = x[i];
  }
  for (int r = 0; r < m; ++r) {
    const double cy = coef[r], cs = coef[m + r];
    double Sv[SV_LB_E], Yv[SV_LB_E];
#pragma unroll
    for (int u = 0; u < SV_LB_E; ++u) {
      const long long i = base + (long long)u * SV_LB_T;
      const bool in = i < n;
      Sv[u] = in ? S[(long long)r * ld + i] : 0.0;
      Yv[u] = in ? Y[(long long)r * ld + i] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < SV_LB_E; ++u) acc[u] = fma(cs, Sv[u], fma(cy, Yv[u], acc[u]));
  }
  double none[1] = {0.0}, mx[1] = {0.0};
#pragma unroll
  for (int u = 0; u < SV_LB_E; ++u) {
    const long long i = base + (long long)u * SV_LB_T;
    if (i < n) {
      d[i] = acc[u];
      mx[0] = nan_max(mx[0], fabs(acc[u]));
    }
  }
  (void)none;
  {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = SV_LB_T >> 6;
    const double v = wave_nan_max(mx[0]);
    if (lane == 0) red[w] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
      double r = red[0];
      for (int q = 1; q < nw; ++q) r = nan_max(r, red[q]);
      part[blockIdx.x] = r;
    }
  }
}

// second stage: block o adds (o < nsum) or maximises the nb block partials of output o, in block order;
// block 0 also appends up to two scalars read from device memory (the closure's value, a carried max|d|)
__global__ __launch_bounds__(256) void k_lbfgs_final(const double* __restrict__ part, int nb, int nsum, int nout,
                                                     double* out, const double* app0, double scale0, const double* app1) {
  __shared__ double red[4];
  const int o = blockIdx.x, tid = threadIdx.x;
  const double* p = part + (long long)o * nb;
  if (o < nsum) {
    double s = 0.0;
    for (int b = tid; b < nb; b += 256) s += p[b];
    s = block_sum(s, red);
    if (tid == 0) out[o] = s;
  } else {
    double s = 0.0;
    for (int b = tid; b < nb; b += 256) s = nan_max(s, p[b]);
    s = wave_nan_max(s);
    if ((tid & 63) == 0) red[tid >> 6] = s;
    __syncthreads();
    if (tid == 0) out[o] = nan_max(nan_max(red[0], red[1]), nan_max(red[2], red[3]));
  }
  if (o == 0 && tid == 0) {
    if (app0) out[nout] = scale0 * *app0;
    if (app1) out[nout + 1] = *app1;
  }
}

// The coefficient step of the vector-free two-loop recursion ON THE DEVICE (round 5): what shardedLBFGS.py's _step_flat did
// on the host between pair_dots and direction -- accept the candidate pair when y.s > 1e-10, extend the Gram matrices
// SY[i][j] = s_i.y_j and YY[i][j] = y_i.y_j by its row and column, H = y.s / y.y, then
//     al_i = (-S_i.g - sum_{j > i} al_j SY[i][j]) / SY[i][i]                          i = m1 - 1 .. 0
//     yq_i = -Y_i.g - sum_j YY[i][j] al_j
//     be_i = (H yq_i + sum_{j < i} (al_j - be_j) SY[j][i]) / SY[i][i]                  i = 0 .. m1 - 1
//     g.d  = -H g.g - H sum al_i Y_i.g + sum (al_i - be_i) S_i.g
// and the coefficients of  d = -H g + sum cY_i y_i + cS_i s_i :  cY = -H al, cS = al - be  -- so that the host need not
// wait for the inner products before it can enqueue the direction, the trial point and the next evaluation: ONE read-back
// per iteration (with the evaluation's) instead of two.  One wavefront; the sums over j run over the lanes and are added in
// a fixed order.  The direction is launched over m + 1 rows: a rejected candidate gets coefficients 0.
// gram: [hs1 hs1] SY | [hs1 hs1] YY | H (state that lives on the device from one iteration to the next).
__global__ __launch_bounds__(64) void k_lbfgs_coef(int m, int hs1, const double* __restrict__ tot, double* gram, double* coef,
                                                  double* out4) {
  __shared__ double Sg[128], Yg[128], al[128], be[128], yq[128];
  const int lane = threadIdx.x;
  double* SY = gram;
  double* YY = gram + (long long)hs1 * hs1;
  double* Hs = gram + 2LL * hs1 * hs1;
  const double ys = tot[0], yy = tot[1], gg = tot[2], sg = tot[3], yg = tot[4];
  const double* S_y = tot + 5;
  const double* s_Y = tot + 5 + m;
  const double* Y_y = tot + 5 + 2 * m;
  const double* S_g = tot + 5 + 3 * m;
  const double* Y_g = tot + 5 + 4 * m;
  const bool acc = ys > 1e-10;
  for (int i = lane; i < m; i += 64) {
    Sg[i] = S_g[i];
    Yg[i] = Y_g[i];
    if (acc) {
      SY[(long long)i * hs1 + m] = S_y[i];
      SY[(long long)m * hs1 + i] = s_Y[i];
      YY[(long long)i * hs1 + m] = Y_y[i];
      YY[(long long)m * hs1 + i] = Y_y[i];
    }
  }
  if (acc && lane == 0) {
    SY[(long long)m * hs1 + m] = ys;
    YY[(long long)m * hs1 + m] = yy;
    Hs[0] = ys / yy;
    Sg[m] = sg;
    Yg[m] = yg;
  }
  __syncthreads();
  __threadfence_block();
  const int m1 = acc ? m + 1 : m;
  const double H = Hs[0];                      // (the caller seeds it; an accepted pair has just replaced it)
  for (int i = m1 - 1; i >= 0; --i) {
    double s = 0.0;
    for (int j = i + 1 + lane; j < m1; j += 64) s = fma(al[j], SY[(long long)i * hs1 + j], s);
    s = wave_sum(s);
    if (lane == 0) al[i] = (-Sg[i] - s) / SY[(long long)i * hs1 + i];
    __syncthreads();
  }
  for (int i = lane; i < m1; i += 64) {
    double s = 0.0;
    for (int j = 0; j < m1; ++j) s = fma(YY[(long long)i * hs1 + j], al[j], s);
    yq[i] = -Yg[i] - s;
  }
  __syncthreads();
  for (int i = 0; i < m1; ++i) {
    double s = 0.0;
    for (int j = lane; j < i; j += 64) s = fma(al[j] - be[j], SY[(long long)j * hs1 + i], s);
    s = wave_sum(s);
    if (lane == 0) be[i] = (H * yq[i] + s) / SY[(long long)i * hs1 + i];
    __syncthreads();
  }
  double a = 0.0, b = 0.0;
  for (int i = lane; i < m1; i += 64) {
    a = fma(al[i], Yg[i], a);
    b = fma(al[i] - be[i], Sg[i], b);
  }
  a = wave_sum(a);
  b = wave_sum(b);
  // coefficients over m + 1 rows (the candidate's row m is part of the direction's launch whether accepted or not)
  for (int i = lane; i <= m; i += 64) {
    const bool in = i < m1;
    coef[i] = in ? -H * al[i] : 0.0;
    coef[(m + 1) + i] = in ? al[i] - be[i] : 0.0;
  }
  if (lane == 0) {
    coef[2 * (m + 1)] = H;
    out4[0] = -H * gg + (-H * a + b);
    out4[1] = (double)m1;
    out4[2] = acc ? 1.0 : 0.0;
    out4[3] = H;
  }
}

inline int lb_blocks(long long n) { return (int)((n + (long long)SV_LB_T * SV_LB_E - 1) / ((long long)SV_LB_T * SV_LB_E)); }

inline int lb_status() { return hipGetLastError() == hipSuccess ? SVGPFA_OK : SVGPFA_ERR_HIP; }

}   // namespace

extern "C" {

int64_t svgpfa_lbfgs_workspace_doubles(int64_t n, int32_t max_history) {
  if (n <= 0 || max_history < 0) return -1;
  return (int64_t)lb_blocks(n) * (5 * (int64_t)max_history + 5);
}

int svgpfa_lbfgs_grad_stats(void* stream, int64_t n, const double* g_raw, double sign, const double* d, double* g,
                            int64_t n_weighted, double weight, const double* value, const double* carry, double* work,
                            double* out6) {
  if (n <= 0 || !g_raw || !g || !work || !out6) return SVGPFA_ERR_INVALID;
  const int nb = lb_blocks(n);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_lbfgs_grad_stats, dim3(nb), dim3(SV_LB_T), 0, st, (long long)n, g_raw, sign, d, g,
                     (long long)n_weighted, weight, work, nb);
  hipLaunchKernelGGL(k_lbfgs_final, dim3(4), dim3(256), 0, st, (const double*)work, nb, 3, 4, out6, value, sign, carry);
  return lb_status();
}

int svgpfa_lbfgs_pair_dots(void* stream, int64_t n, const double* g, const double* g_prev, const double* d, double t,
                           double* S, double* Y, int64_t row_stride, int32_t m, int64_t n_weighted, double weight,
                           double* work, double* out) {
  if (n <= 0 || m < 0 || !g || !g_prev || !d || !S || !Y || row_stride < n || !work || !out) return SVGPFA_ERR_INVALID;
  const int nb = lb_blocks(n);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_lbfgs_pair_dots, dim3(nb), dim3(SV_LB_T), 0, st, (long long)n, g, g_prev, d, t, S, Y,
                     (long long)row_stride, (int)m, (long long)n_weighted, weight, work, nb);
  const int no = 5 * m + 5;
  hipLaunchKernelGGL(k_lbfgs_final, dim3(no), dim3(256), 0, st, (const double*)work, nb, no, no, out,
                     (const double*)nullptr, 1.0, (const double*)nullptr);
  return lb_status();
}

int svgpfa_lbfgs_direction(void* stream, int64_t n, const double* g, const double* S, const double* Y, int64_t row_stride,
                           int32_t m, const double* coef, double* d, const double* x, double* x0, double* work,
                           double* out_dmax) {
  if (n <= 0 || m < 0 || !g || !coef || !d || !work || !out_dmax || (m > 0 && (!S || !Y || row_stride < n)) ||
      ((x0 != nullptr) != (x != nullptr)))
    return SVGPFA_ERR_INVALID;
  const int nb = lb_blocks(n);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_lbfgs_direction, dim3(nb), dim3(SV_LB_T), 0, st, (long long)n, g, S, Y, (long long)row_stride,
                     (int)m, coef, d, x, x0, work, nb);
  hipLaunchKernelGGL(k_lbfgs_final, dim3(1), dim3(256), 0, st, (const double*)work, nb, 0, 1, out_dmax,
                     (const double*)nullptr, 1.0, (const double*)nullptr);
  return lb_status();
}

int svgpfa_lbfgs_coef(void* stream, int32_t m, int32_t history_rows, const double* dots, double* gram, double* coef,
                      double* out4) {
  if (m < 0 || history_rows <= m || history_rows > 128 || !dots || !gram || !coef || !out4) return SVGPFA_ERR_INVALID;
  hipLaunchKernelGGL(k_lbfgs_coef, dim3(1), dim3(64), 0, (hipStream_t)stream, (int)m, (int)history_rows, dots, gram, coef, out4);
  return lb_status();
}

}   // extern "C"
