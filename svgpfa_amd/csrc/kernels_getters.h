// Kernels behind the KernelsMatricesStore getters and the statistics API (tests, plots, the embedding M-step): none
// of them is on the ELBO hot path.
//   k_ktz           : materialise a Ktz block (reference kernelsMatricesStore.py:186-195, :208-221)
//   k_transpose_ks  : [K][S] -> [S][K] (spike-time statistics as the reference lays them out)
//   k_transpose_rkq : [R][K][Q] -> [R][Q][K] (quadrature statistics as the reference lays them out)
#pragma once
#include "engine_common.h"

__global__ void k_ktz(EngineView ev, int k, const double* times, const long long* toff, int nper,
                      long long ntot, double* out) {
  const LatentMeta m = ev.meta[k];
  const KPar p = make_kpar(m, ev.theta + 2 * k);
  long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= ntot * m.M) return;
  long long gi = id / m.M;
  int j = (int)(id % m.M);
  int r;
  if (toff) {
    int lo = 0, hi = ev.R;  // find trial with toff[r] <= gi < toff[r+1]
    while (hi - lo > 1) {
      int mid = (lo + hi) >> 1;
      if (toff[mid] <= gi) lo = mid; else hi = mid;
    }
    r = lo;
  } else {
    r = (int)(gi / nper);
  }
  double d = times[gi] - ev.Zs[m.offM + (long long)r * m.M + j];
  out[id] = p.s2 * (m.ktype == SV_RBF ? kern_val<SV_RBF>(d, p) : kern_val<SV_PERIODIC>(d, p));
}

__global__ void k_transpose_ks(const double* in, double* out, long long S, int K) {
  long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= S * K) return;
  long long s = id / K;
  int k = (int)(id % K);
  out[id] = in[(long long)k * S + s];
}

__global__ void k_transpose_rkq(const double* in, double* out, int R, int K, int Q) {
  long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= (long long)R * K * Q) return;
  const int k = (int)(id % K);
  const long long rq = id / K;
  const int q = (int)(rq % Q);
  const long long r = rq / Q;
  out[id] = in[(r * K + k) * Q + q];
}
