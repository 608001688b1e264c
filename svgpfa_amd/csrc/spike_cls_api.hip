// The class path's kernels and their launch wrappers (see spike_cls_launch.h for why they are a translation unit of their own).
#include "spike_cls_launch.h"

#include <algorithm>
#include <cstdlib>

#include "kernels_spike_cls.h"

// Every wrapper returns the launch status (hipGetLastError right behind the launch): a refused launch -- a grid of zero
// blocks, an LDS request the device cannot serve -- must reach svgpfa_last_error, not vanish.
hipError_t sv_launch_spike_classes(const EngineView& v, hipStream_t s) {
  hipLaunchKernelGGL(k_spike_classes, dim3((v.R + 3) / 4), dim3(256), 0, s, v);
  return hipGetLastError();
}

hipError_t sv_launch_spike_pre(const EngineView& v, hipStream_t s) {
  hipLaunchKernelGGL(k_spike_pre, dim3(v.R * v.K), dim3(64), 0, s, v);
  return hipGetLastError();
}

hipError_t sv_launch_spike_cls(const EngineView& v, const double* C, int nv, int dc, int full, int pair, unsigned grid, hipStream_t s) {
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
  return hipGetLastError();
}

hipError_t sv_launch_spike_comb(const EngineView& v, int full, hipStream_t s) {      // full: bit 0 full, bit 1 time-weighted moments, bit 2 two items per wave
  const int pairs = (full >> 2) & 1;
  const dim3 grid(pairs ? (v.R * v.K + 1) / 2 : v.R * v.K);
  if (full & 1) hipLaunchKernelGGL(k_spike_comb<true>, grid, dim3(64), 0, s, v, (full >> 1) & 1, pairs);
  else hipLaunchKernelGGL(k_spike_comb<false>, grid, dim3(64), 0, s, v, 0, pairs);
  return hipGetLastError();
}

template <int KS, int DC>
static hipError_t launch_ell_cls_t(const EngineView& v, const double* C, const double* d, int want_sums, int want_lat, int parts,
                                   int lc0, int lc1, int full, int pair, unsigned workers, size_t lds_ell, hipStream_t s) {
  const size_t lds = std::max(lds_ell, (size_t)4 * SV_CLS_SLICE * sizeof(double));
  int nell = v.R * parts * ((want_lat && want_sums) ? 2 : 1);      // (latent adjoints alone: role-1 blocks only)
#ifdef SV_ELLCLS_DIAG     // diagnosis BUILDS only (results are wrong; tests/tools/ellcls_diag.sh): 1 launches the ELL blocks alone, 2 the class workers alone
  if (SV_ELLCLS_DIAG == 1) workers = 0;
  if (SV_ELLCLS_DIAG == 2) nell = 0;
#endif
  const unsigned grid = (unsigned)nell + (workers + 3) / 4;
  hipLaunchKernelGGL((k_ell_cls<KS, DC>), dim3(grid), dim3(256), lds, s, v, C, d, want_sums, want_lat, parts, lc0, lc1, full, nell, pair);
  return hipGetLastError();
}

// The dynamic-LDS allowance of k_ell_cls<ks, *> on the CURRENT device (svgpfa_create calls this once per engine, as it does
// for every other kernel above 48 KB: the attribute is per device, and an engine on a second device of the process must
// set it there too).
template <int KS>
static hipError_t allow_ell_cls_t() {
  hipError_t st = hipFuncSetAttribute((const void*)k_ell_cls<KS, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (st != hipSuccess) return st;
  return hipFuncSetAttribute((const void*)k_ell_cls<KS, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}
hipError_t sv_allow_ell_cls(int ks) {
  switch (ks) {
    case 1: return allow_ell_cls_t<1>();   case 2: return allow_ell_cls_t<2>();   case 3: return allow_ell_cls_t<3>();
    case 4: return allow_ell_cls_t<4>();   case 5: return allow_ell_cls_t<5>();   case 6: return allow_ell_cls_t<6>();
    case 7: return allow_ell_cls_t<7>();   case 8: return allow_ell_cls_t<8>();   case 9: return allow_ell_cls_t<9>();
    case 10: return allow_ell_cls_t<10>(); case 11: return allow_ell_cls_t<11>(); case 12: return allow_ell_cls_t<12>();
    case 13: return allow_ell_cls_t<13>(); case 14: return allow_ell_cls_t<14>(); case 15: return allow_ell_cls_t<15>();
    default: return allow_ell_cls_t<16>();
  }
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
