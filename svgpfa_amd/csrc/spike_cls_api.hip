// The class path's kernels and their launch wrappers (see spike_cls_launch.h for why they are a translation unit of their own).
#include "spike_cls_launch.h"

#include "kernels_spike_cls.h"

void sv_launch_spike_classes(const EngineView& v, hipStream_t s) {
  hipLaunchKernelGGL(k_spike_classes, dim3((v.R + 3) / 4), dim3(256), 0, s, v);
}

void sv_launch_spike_pre(const EngineView& v, hipStream_t s) {
  hipLaunchKernelGGL(k_spike_pre, dim3(v.R * v.K), dim3(64), 0, s, v);
}

void sv_launch_spike_cls(const EngineView& v, const double* C, int nv, int dc, int full, unsigned grid, hipStream_t s) {
  const dim3 gc(grid), bs(64);
  if (nv == 0) {
    if (dc) hipLaunchKernelGGL((k_spike_cls_all<1>), gc, bs, 0, s, v, C, full);
    else hipLaunchKernelGGL((k_spike_cls_all<0>), gc, bs, 0, s, v, C, full);
  } else if (nv == 4) {
    if (dc) hipLaunchKernelGGL((k_spike_cls<4, 1>), gc, bs, 0, s, v, C, full);
    else hipLaunchKernelGGL((k_spike_cls<4, 0>), gc, bs, 0, s, v, C, full);
  } else {
    if (dc) hipLaunchKernelGGL((k_spike_cls<2, 1>), gc, bs, 0, s, v, C, full);
    else hipLaunchKernelGGL((k_spike_cls<2, 0>), gc, bs, 0, s, v, C, full);
  }
}

void sv_launch_spike_comb(const EngineView& v, bool full, hipStream_t s) {
  if (full) hipLaunchKernelGGL(k_spike_comb<true>, dim3(v.R * v.K), dim3(64), 0, s, v);
  else hipLaunchKernelGGL(k_spike_comb<false>, dim3(v.R * v.K), dim3(64), 0, s, v);
}

int sv_spike_cls_blocks_per_cu() {
  int bc = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&bc, (const void*)k_spike_cls<4, 1>, 64, 0) != hipSuccess) return 0;
  return bc;
}
