//   k_ktz       : materialise a Ktz block for the KernelsMatricesStore getters (tests / plots only)
// The posterior, ELL, quadrature-backward and per-item backward kernels live in kernels_mfma.h.
#pragma once
#include "engine_common.h"

// ------------------------------------------------------------------------------------ k_ktz
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
