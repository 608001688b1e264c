// The class path's kernels and their launch wrappers (see spike_cls_launch.h for why they are a translation unit of their own).
#include "spike_cls_launch.h"

#include <algorithm>
#include <cstdlib>

#include "kernels_spike_cls.h"

void sv_launch_spike_classes(const EngineView& v, hipStream_t s) {
  hipLaunchKernelGGL(k_spike_classes, dim3((v.R + 3) / 4), dim3(256), 0, s, v);
}

void sv_launch_spike_pre(const EngineView& v, hipStream_t s) {
  hipLaunchKernelGGL(k_spike_pre, dim3(v.R * v.K), dim3(64), 0, s, v);
}

void sv_launch_spike_cls(const EngineView& v, const double* C, int nv, int dc, int full, int pair, unsigned grid, hipStream_t s) {
  const dim3 gc(grid), bs(64);
  if (nv == 0) {
    if (dc) hipLaunchKernelGGL((k_spike_cls_all<1>), gc, bs, 0, s, v, C, full, pair);
    else hipLaunchKernelGGL((k_spike_cls_all<0>), gc, bs, 0, s, v, C, full, pair);
  } else if (nv == 4) {
    if (dc) hipLaunchKernelGGL((k_spike_cls<4, 1>), gc, bs, 0, s, v, C, full, 0);
    else hipLaunchKernelGGL((k_spike_cls<4, 0>), gc, bs, 0, s, v, C, full, 0);
  } else {
    if (dc) hipLaunchKernelGGL((k_spike_cls<2, 1>), gc, bs, 0, s, v, C, full, pair);
    else hipLaunchKernelGGL((k_spike_cls<2, 0>), gc, bs, 0, s, v, C, full, pair);
  }
}

void sv_launch_spike_comb(const EngineView& v, bool full, hipStream_t s) {
  if (full) hipLaunchKernelGGL(k_spike_comb<true>, dim3(v.R * v.K), dim3(64), 0, s, v);
  else hipLaunchKernelGGL(k_spike_comb<false>, dim3(v.R * v.K), dim3(64), 0, s, v);
}

template <int KS, int DC>
static hipError_t launch_ell_cls_t(const EngineView& v, const double* C, const double* d, int want_sums, int want_lat, int parts,
                                   int lc0, int lc1, int full, int pair, unsigned workers, size_t lds_ell, hipStream_t s) {
  const size_t lds = std::max(lds_ell, (size_t)4 * SV_CLS_SLICE * sizeof(double));
  static bool allowed = false;           // (one attribute call per instantiation and process)
  if (!allowed && lds > 48 * 1024) {
    hipError_t st = hipFuncSetAttribute((const void*)k_ell_cls<KS, DC>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (st != hipSuccess) return st;
    allowed = true;
  }
  int nell = v.R * parts * (want_lat ? 2 : 1);
  {   // diagnosis only (results are wrong): SVGPFA_ELLCLS_DIAG=1 launches the ELL blocks alone, =2 the class workers alone
    static const int diag = getenv("SVGPFA_ELLCLS_DIAG") ? atoi(getenv("SVGPFA_ELLCLS_DIAG")) : 0;
    if (diag == 1) workers = 0;
    if (diag == 2) nell = 0;
  }
  const unsigned grid = (unsigned)nell + (workers + 3) / 4;
  hipLaunchKernelGGL((k_ell_cls<KS, DC>), dim3(grid), dim3(256), lds, s, v, C, d, want_sums, want_lat, parts, lc0, lc1, full, nell, pair);
  return hipGetLastError();
}

hipError_t sv_launch_ell_cls(const EngineView& v, const double* C, const double* d, int ks, int dc, int want_sums, int want_lat,
                             int parts, int lc0, int lc1, int full, int pair, unsigned workers, size_t lds_ell, hipStream_t s) {
#define SV_ELL_CLS_CASE(K_)                                                                                              \
  case K_:                                                                                                               \
    return dc ? launch_ell_cls_t<K_, 1>(v, C, d, want_sums, want_lat, parts, lc0, lc1, full, pair, workers, lds_ell, s)       \
              : launch_ell_cls_t<K_, 0>(v, C, d, want_sums, want_lat, parts, lc0, lc1, full, pair, workers, lds_ell, s);
  switch (ks) {
    SV_ELL_CLS_CASE(1) SV_ELL_CLS_CASE(2) SV_ELL_CLS_CASE(3) SV_ELL_CLS_CASE(4) SV_ELL_CLS_CASE(5) SV_ELL_CLS_CASE(6)
    SV_ELL_CLS_CASE(7) SV_ELL_CLS_CASE(8) SV_ELL_CLS_CASE(9) SV_ELL_CLS_CASE(10) SV_ELL_CLS_CASE(11) SV_ELL_CLS_CASE(12)
    SV_ELL_CLS_CASE(13) SV_ELL_CLS_CASE(14) SV_ELL_CLS_CASE(15)
    default:
      return dc ? launch_ell_cls_t<16, 1>(v, C, d, want_sums, want_lat, parts, lc0, lc1, full, pair, workers, lds_ell, s)
                : launch_ell_cls_t<16, 0>(v, C, d, want_sums, want_lat, parts, lc0, lc1, full, pair, workers, lds_ell, s);
  }
#undef SV_ELL_CLS_CASE
}

int sv_spike_cls_blocks_per_cu() {
  int bc = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&bc, (const void*)k_spike_cls<4, 1>, 64, 0) != hipSuccess) return 0;
  return bc;
}
