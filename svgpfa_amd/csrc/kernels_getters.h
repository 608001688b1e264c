// Kernels behind the KernelsMatricesStore getters and the statistics API (tests, plots, the embedding M-step): none
// of them is on the ELBO hot path.
//   k_ktz           : materialise a Ktz block (reference kernelsMatricesStore.py:186-195, :208-221)
//   k_transpose_b   : batched 2-D transpose ([R][K][Q] <-> [R][Q][K], [K][S] <-> [S][K])
//   k_sum_kl        : the KL term alone
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

// out[b][c][r] = in[b][r][c]: layout conversions between the engine's buffers ([R][K][Q], [K][S]) and the reference's
// ([R][Q][K], [S][K]) for the statistics API.
__global__ void k_transpose_b(const double* in, double* out, long long B, long long rows, long long cols) {
  long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= B * rows * cols) return;
  const long long r = id % rows, bc = id / rows, c = bc % cols, b = bc / cols;
  out[id] = in[(b * rows + r) * cols + c];
}

// KL = sum of the per-item terms k_prep left in klpart (statistics API; fixed order)
__global__ __launch_bounds__(256) void k_sum_kl(EngineView ev, double* out) {
  __shared__ double red[4];
  double s = 0.0;
  for (int i = threadIdx.x; i < ev.R * ev.K; i += 256) s += ev.klpart[i];
  s = block_sum(s, red);
  if (threadIdx.x == 0) out[0] = s;
}
