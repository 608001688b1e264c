// Host-side launch wrappers of the class path's kernels (kernels_spike_cls.h).  Those kernels live in a translation unit
// of their own, spike_cls_api.hip, compiled with -mllvm -disable-machine-licm: their work loops are persistent (a block
// strides over (tile, class) items), and the compiler's machine-level loop-invariant code motion hoists every fp64
// polynomial constant of the loop body (sincos: 14 doubles) into registers that stay live across the whole loop --
// with it the kernels spill 60-97 registers, without it 32-65, and run 38 + 38 us instead of 47 + 43 at the north-star
// size (round 4; the same flag slows k_quad_ell_mfma down by 8 %, hence not the whole library).
#pragma once
#include <hip/hip_runtime.h>

#include "engine_common.h"

// (every wrapper returns the launch status)
hipError_t sv_launch_spike_classes(const EngineView& v, hipStream_t s);
hipError_t sv_launch_spike_pre(const EngineView& v, hipStream_t s);
// nv: 4 / 2 = the kernel of the classes with / without time-weighted moments, 0 = both kinds in one launch; dc: the rows'
// dC partials wanted (needs W); pair: the pair-by-pair pass (fp64) rides behind the class items (nv = 2 or 0 only)
hipError_t sv_launch_spike_cls(const EngineView& v, const double* C, int nv, int dc, int full, int pair, unsigned grid, hipStream_t s);
hipError_t sv_launch_spike_comb(const EngineView& v, int full, hipStream_t s);
// `full` of the class kernels and the combine is a bit set: bit 0 = gradient sums wanted (Z / theta / C), bit 1 = the
// time-weighted moments of periodic classes wanted (kernel-parameter gradient: the period's T1 is their only consumer)
// k_ell_cls: the quadrature ELL blocks and the class items in one launch (small shards, fp64); ks = k-steps of the ELL
// product (1 .. 16), lds_ell = dynamic LDS the ELL roles need.  Returns hipSuccess or the launch / attribute error.
hipError_t sv_launch_ell_cls(const EngineView& v, const double* C, const double* d, int ks, int dc, int want_sums, int want_lat,
                             int parts, int lc0, int lc1, int full, int pair, unsigned workers, size_t lds_ell, hipStream_t s);
hipError_t sv_allow_ell_cls(int ks);   // dynamic-LDS allowance of k_ell_cls<ks, *> on the current device (svgpfa_create)
int sv_spike_cls_blocks_per_cu();      // resident single-wave blocks per CU of the largest class kernel (0 on error)
