// libsvgpfa_hip.so -- C ABI of the MI355X-native svGPFA ELBO engine (see include/svgpfa.h).
// Host side: owns HBM buffers, validates shapes, enqueues the kernel pipeline on one HIP stream.
#include "../../include/svgpfa.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "engine_common.h"
#include "kernels_getters.h"
#include "kernels_item.h"
#include "kernels_item_bwd.h"
#include "kernels_mfma.h"
#include "kernels_spike.h"
#include "spike_cls_launch.h"

typedef LatentPtrTable PtrTable;

// k_prep / k_item_bwd read the caller's per-latent tensors: the pointer table travels by value in
// the kernarg segment, so nothing has to be staged through HBM when the caller's tensors move.

struct svgpfa_engine {
  EngineView v{};
  std::vector<LatentMeta> meta;
  std::vector<void*> allocs;
  hipStream_t stream = nullptr;
  int device = 0;
  int Mmax = 0, NT = 1;
  int precision = SVGPFA_PRECISION_FP64;   // MIXED: fp32 spike pass and quadrature ELL (BASELINE config 5)
  int ell_parts = 0, KKT = 1;  // ell_parts > 0: MFMA GEMM form of the quadrature term
  int ell_lc[2] = {0, 0};      // looped tiles resident in LDS per role (fewer than all: streamed in chunks)
  int ell_nw = 4;              // waves per block of the ELL kernels
  int ellf_groups = 0;         // > 0: evaluations that want dC / dd AND latent adjoints run k_quad_ell_fused with this many neuron groups per trial
  int times_nw = 4;            // waves per block of k_times_fwd (SVGPFA_TIMES_NW, up to SV_TIMES_MAXTH / 64)
  int tile_nw = 4;             // waves per block of k_item_bwd = NT x teams (a multiple of NT, at most 4)
  long long totM = 0, totMM = 0, totTT = 0;  // R*sum M, R*sum M^2, R*sum M(M+1)/2
  bool have_quad = false, have_spikes = false, have_kernels = false, have_stats = false;
  bool aux_valid = false;   // Kzz / L buffers hold the matrices of the current kernel snapshot (getters only)
  bool kw_valid = false;    // EngineView::Kwc holds the whitened quadrature tiles of the current kernels and nodes
  bool kw_cache = true;     // keep them across E-step evaluations (SVGPFA_KW_CACHE=0: off)
  bool timing = false;
  int timing_period = 1;                 // bracket the kernels of every timing_period-th evaluation (svgpfa_enable_timing)
  unsigned long long timing_eval = 0;    // evaluations enqueued since timing was enabled
  uint64_t emb_token = 0, p0_token = 0;   // svgpfa_set_embedding_token: content id of C now / when P0 was computed
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool;  // event pairs, reused
  std::vector<int> ev_slot;                                // which kernel a pair brackets: 0 k_spike, 1 k_item_bwd
  size_t ev_used = 0;
  double ev_tot[2] = {0.0, 0.0};                           // folded totals per slot (ms) and launches
  long long ev_cnt[2] = {0, 0};
  std::string err;
  int* status_host = nullptr;  // pinned
  // Launch strategy.  Every kernel is launched ONCE over all latents with the covariance type resolved per block
  // (wave-uniform branch): measured faster than one launch per type for the per-item kernels (+3 % on cfg3: fewer
  // launches, shorter grid tails) and, since both types of the spike pass fit 3 waves / SIMD, for the spike pass too
  // (+1.5 %; its grid lists the longer periodic work items first).
  // ONE stream at every size.  (Rounds 2-3 measured the quadrature chain or the second class kernel on side streams and
  // the whole evaluation replayed as a HIP graph: an edge that crosses streams costs 12-20 us at the end of a chain, the
  // class kernels fill every wave slot of the chip, kernels side by side slow each other down, and a graph's branches run
  // through the same queues and barrier packets -- all removed in round 4)
  // Both kinds of class (with / without time-weighted moments) in ONE launch: default up to R K = 4096 items (one launch
  // boundary -- a drained chip -- less: one rank's shard of cfg3, ms per evaluation, two kernels -> one: 2-way 0.405 -> 0.386,
  // 4-way 0.266 -> 0.249, 8-way 0.198 -> 0.176; at the full 5 000 items 0.655 / 0.659 in round 3, 62 -> 52 us of class kernels
  // in round 5: cls_single_kernel).  -1: one kernel at every size, and up to 4 096 items the ELL blocks in the same launch
  // (k_ell_cls); SVGPFA_CLS_FUSED=1: both at every size; 0: two class kernels, launches apart.
  int cls_fused = -1;
  int n_cu = 256;                               // compute units of the device
  bool comb_pairs = true;                       // the spike moments' combine with two items per wave where M <= 32 (SVGPFA_COMB_PAIRS=0: one)
  bool ell_cls = true;                          // small shards: ELL blocks and class items in one launch (SVGPFA_ELL_CLS=0: off)
  bool comb_ride = true;                        // the moments' combine rides in the ELL kernel's launch at full size (SVGPFA_COMB_RIDE=0: own launch)
  bool pair_ride = true;                        // the pair-by-pair pass rides in the class kernels' launch (SVGPFA_PAIR_RIDE=0: own launch)
  int build_wpe = -1;                           // k_build_reg<32>: -1 by grid size, 2 / 3 waves per SIMD forced (SVGPFA_BUILD_WPE)
  int build_mode = -1;                          // kernel build at 16 < M <= 32: -1 / 0 k_build_reg, 1 k_build_mfma (SVGPFA_BUILD=mfma); above 32 always k_build_mfma
  bool front = true;                            // k_front instead of k_prep_mfma + k_times_fwd_mfma in svgpfa_elbo (SVGPFA_FRONT=0: off)
  bool cls_on = true;                           // false: problem too small for the class path's extra launches (set_spikes)
  double series_min_pairs = 1e7;                // SVGPFA_SERIES_MIN_PAIRS: fewest (spike, inducing point) pairs for the class path
  bool cls_valid = false;                       // the class table of the spike pass belongs to the current kernel build
  int grid_cls = 2048, grid_pair = 2048;        // blocks of the two persistent-style spike kernels (resident waves of the chip)
};

#define HIPCHK(e, call)                                                                  \
  do {                                                                                   \
    hipError_t _s = (call);                                                              \
    if (_s != hipSuccess) {                                                              \
      (e)->err = std::string(#call) + ": " + hipGetErrorString(_s);                      \
      return SVGPFA_ERR_HIP;                                                             \
    }                                                                                    \
  } while (0)

#define FAIL(e, code, msg) \
  do {                     \
    (e)->err = (msg);      \
    return (code);         \
  } while (0)

template <typename T>
static int dalloc(svgpfa_engine* e, T** p, size_t n) {
  void* q = nullptr;
  if (n == 0) n = 1;
  hipError_t s = hipMalloc(&q, n * sizeof(T));
  if (s != hipSuccess) {
    e->err = std::string("hipMalloc: ") + hipGetErrorString(s);
    return SVGPFA_ERR_HIP;
  }
  e->allocs.push_back(q);
  *p = (T*)q;
  return SVGPFA_OK;
}
#define DALLOC(e, p, n)                      \
  do {                                       \
    int _r = dalloc((e), &(p), (size_t)(n)); \
    if (_r) return _r;                       \
  } while (0)

static size_t lds_times_mfma(int NT) {
  size_t MP = 16 * NT, LD = MP + 2;
  return (2 * MP * LD + 4 * MP) * 8;
}
static size_t lds_item_bwd(int NT, int NWB, bool full) { return sv_item_bwd_lds_doubles(NT, NWB, full) * 8; }
#define DISPATCH_NT(nt, ...)                                 \
  switch (nt) {                                              \
    case 1: { constexpr int NT_ = 1; __VA_ARGS__; } break;   \
    case 2: { constexpr int NT_ = 2; __VA_ARGS__; } break;   \
    case 3: { constexpr int NT_ = 3; __VA_ARGS__; } break;   \
    default: { constexpr int NT_ = 4; __VA_ARGS__; } break;  \
  }
// LDS of k_quad_ell_mfma with lc looped tiles resident; qell_chunk = the largest lc within the budget (0: none fits)
static size_t lds_qell_mfma_lc(const svgpfa_engine* e, int role, int lc) {
  size_t K2 = 2 * e->v.K, KKP = (K2 + 3) & ~(size_t)3, LDK = KKP + 2;
  size_t Lp = (((role == 0 ? e->v.Q : e->v.N) + 15) / 16) * 16;
  const size_t elt = e->precision == SVGPFA_PRECISION_MIXED ? 4 : 8;      // operands in LDS: float in mixed precision
  return (((size_t)lc * 16 + 1) * LDK + Lp + (size_t)e->ell_nw * 16 * LDK + 1) * elt + 9 * 8;
}
static int qell_chunk(const svgpfa_engine* e, int role) {
  const int LT = ((role == 0 ? e->v.Q : e->v.N) + 15) / 16;
  int lc = LT;
  while (lc > 0 && lds_qell_mfma_lc(e, role, lc) > 64 * 1024) --lc;
  return lc;
}
static size_t lds_qell_mfma(const svgpfa_engine* e, int role) { return lds_qell_mfma_lc(e, role, e->ell_lc[role]); }
#define DISPATCH_KS(ks, ...)                                                                            \
  switch (ks) {                                                                                         \
    case 1: { constexpr int KS_ = 1; __VA_ARGS__; } break;   case 2: { constexpr int KS_ = 2; __VA_ARGS__; } break;   \
    case 3: { constexpr int KS_ = 3; __VA_ARGS__; } break;   case 4: { constexpr int KS_ = 4; __VA_ARGS__; } break;   \
    case 5: { constexpr int KS_ = 5; __VA_ARGS__; } break;   case 6: { constexpr int KS_ = 6; __VA_ARGS__; } break;   \
    case 7: { constexpr int KS_ = 7; __VA_ARGS__; } break;   case 8: { constexpr int KS_ = 8; __VA_ARGS__; } break;   \
    case 9: { constexpr int KS_ = 9; __VA_ARGS__; } break;   case 10: { constexpr int KS_ = 10; __VA_ARGS__; } break; \
    case 11: { constexpr int KS_ = 11; __VA_ARGS__; } break; case 12: { constexpr int KS_ = 12; __VA_ARGS__; } break; \
    case 13: { constexpr int KS_ = 13; __VA_ARGS__; } break; case 14: { constexpr int KS_ = 14; __VA_ARGS__; } break; \
    case 15: { constexpr int KS_ = 15; __VA_ARGS__; } break; default: { constexpr int KS_ = 16; __VA_ARGS__; } break; \
  }
// the fused ELL kernel exists for 2K <= 20 (SV_ELLF_KSMAX)
#define DISPATCH_KSF(ks, ...)                                                                           \
  switch (ks) {                                                                                         \
    case 1: { constexpr int KS_ = 1; __VA_ARGS__; } break;   case 2: { constexpr int KS_ = 2; __VA_ARGS__; } break;   \
    case 3: { constexpr int KS_ = 3; __VA_ARGS__; } break;   case 4: { constexpr int KS_ = 4; __VA_ARGS__; } break;   \
    default: { constexpr int KS_ = 5; __VA_ARGS__; } break;  \
  }
static size_t lds_qell_fused(int KS, int Q) {      // (the rider blocks' combine area fits below it)
  return std::max((size_t)sv_ellf_lds_doubles(KS, (Q + 15) / 16), (size_t)SV_ELLF_NW * 2 * SV_COMB_LDS) * sizeof(double);
}
static size_t lds_prep_mfma(int NT) {
  size_t MP = 16 * NT, LD = MP + 2;
  return (2 * MP * LD + NT + 64 + SV_PRE_LDS) * 8;
}

template <typename F>
static int allow_lds(svgpfa_engine* e, F f, size_t bytes) {
  if (bytes > 160 * 1024) FAIL(e, SVGPFA_ERR_UNSUPPORTED, "problem needs more than 160 KiB of LDS per workgroup");
  if (bytes > 48 * 1024)
    HIPCHK(e, hipFuncSetAttribute((const void*)f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
  return SVGPFA_OK;
}

extern "C" {

int svgpfa_create(const svgpfa_dims* d, int device, svgpfa_engine** out) {
  if (!d || !out) return SVGPFA_ERR_INVALID;
  *out = nullptr;
  svgpfa_engine* e = new svgpfa_engine();
  *out = e;  // returned even on failure so that the caller can read last_error, then destroy
  e->device = device;
  if (d->n_trials <= 0 || d->n_neurons <= 0 || d->n_latents <= 0 || d->n_quad <= 0 || !d->n_ind ||
      !d->kernel_type)
    FAIL(e, SVGPFA_ERR_INVALID, "svgpfa_create: non-positive dimension or NULL descriptor array");
  if (d->n_latents > SV_MAX_K) FAIL(e, SVGPFA_ERR_UNSUPPORTED, "n_latents > 32");
  if (d->precision != SVGPFA_PRECISION_FP64 && d->precision != SVGPFA_PRECISION_MIXED)
    FAIL(e, SVGPFA_ERR_INVALID, "svgpfa_create: unknown precision mode");
  e->precision = d->precision;
  HIPCHK(e, hipSetDevice(device));
  EngineView& v = e->v;
  v.R = d->n_trials; v.N = d->n_neurons; v.K = d->n_latents; v.Q = d->n_quad;
  v.reg = 0.0;
  e->meta.resize(v.K);
  long long offM = 0, offMM = 0, offTT = 0;
  int colM = 0, thoff = 0;
  for (int k = 0; k < v.K; ++k) {
    LatentMeta& m = e->meta[k];
    m.M = d->n_ind[k];
    if (m.M <= 0 || m.M > SV_MAX_M) FAIL(e, SVGPFA_ERR_UNSUPPORTED, "inducing points per latent must be in 1..64");
    m.P = m.M * (m.M + 1) / 2;
    m.PT = sv_tri_off(m.M);
    m.ktype = d->kernel_type[k];
    if (m.ktype != SV_RBF && m.ktype != SV_PERIODIC) FAIL(e, SVGPFA_ERR_INVALID, "unknown kernel type");
    m.nth = (m.ktype == SV_RBF) ? 1 : 2;
    m.offM = offM; m.offMM = offMM; m.offTT = offTT; m.colM = colM; m.thoff = thoff;
    double s = d->scale ? d->scale[k] : 1.0;
    m.scale2 = s * s;
    m.inv_ls_scale = 1.0 / (d->lengthscale_scale ? d->lengthscale_scale[k] : 1.0);
    m.inv_p_scale = 1.0 / (d->period_scale ? d->period_scale[k] : 1.0);
    offM += (long long)v.R * m.M;
    offMM += (long long)v.R * m.M * m.M;
    offTT += (long long)v.R * m.PT;
    colM += m.M;
    thoff += m.nth;
    e->Mmax = std::max(e->Mmax, m.M);
  }
  v.ntype[0] = v.ntype[1] = 0;
  for (int k = 0; k < v.K; ++k) {
    int t = e->meta[k].ktype;
    v.klist[t][v.ntype[t]++] = k;
  }
  e->totM = offM; e->totMM = offMM; e->totTT = offTT;
  v.sumM = colM; v.nth_total = thoff;
  e->NT = (e->Mmax + 15) / 16;

  LatentMeta* dmeta; DALLOC(e, dmeta, v.K);
  HIPCHK(e, hipMemcpy(dmeta, e->meta.data(), sizeof(LatentMeta) * v.K, hipMemcpyHostToDevice));
  v.meta = dmeta;
  double* tq; double* wq; DALLOC(e, tq, (size_t)v.R * v.Q); DALLOC(e, wq, (size_t)v.R * v.Q);
  v.tq = tq; v.wq = wq;
  DALLOC(e, v.Zs, e->totM); DALLOC(e, v.theta, 2 * v.K);
  DALLOC(e, v.four, (size_t)v.K * SV_FROWS * (SV_FH + 2)); DALLOC(e, v.fourH, v.K);
  HIPCHK(e, hipMemset(v.fourH, 0xff, sizeof(int) * v.K));
  DALLOC(e, v.Kzz, e->totMM); DALLOC(e, v.L, e->totMM); DALLOC(e, v.Linv, e->totTT);
  DALLOC(e, v.a, e->totM); DALLOC(e, v.ut, e->totM); DALLOC(e, v.Vt, e->totTT); DALLOC(e, v.klpart, (size_t)v.R * v.K);
  size_t rqk = (size_t)v.R * v.Q * v.K;
  e->KKT = (2 * v.K + 3) / 4;   // KS: k-steps of the h product
  // (the fused ELL kernel's neuron groups each write a plane of the latent adjoints: kernels_mfma.h k_quad_ell_fused)
  const int ellf_groups_max = ((v.N + 15) / 16 + SV_ELLF_NG - 1) / SV_ELLF_NG;
  // (shape limits: 2K <= 20, at most two groups of SV_ELLF_NG neuron tiles, at least one node tile per wave, both operand
  // matrices of a block within half of a CU's LDS)
  const bool ellf_can = e->precision == SVGPFA_PRECISION_FP64 && e->KKT <= SV_ELLF_KSMAX && ellf_groups_max <= 2 &&
                        (v.Q + 15) / 16 >= SV_ELLF_NW && lds_qell_fused(e->KKT, v.Q) <= 80 * 1024;
  const size_t gplanes = (ellf_can && ellf_groups_max == 2) ? 2 : 1;
  DALLOC(e, v.muq, rqk); DALLOC(e, v.varq, rqk); DALLOC(e, v.gmuq, gplanes * rqk); DALLOC(e, v.gvarq, gplanes * rqk);
  v.glat2 = 0;
  // blocks per trial of the ELL kernels: every block re-stages the trial's looped operand in LDS, so fewer blocks
  // per trial are cheaper as long as the grid still fills the chip (measured on cfg3: 1 -> 151 us, 4 -> 168 us)
  e->ell_parts = std::min(4, std::max(1, (383 + v.R) / v.R));
  if (getenv("SVGPFA_ELL_PARTS")) e->ell_parts = std::max(1, atoi(getenv("SVGPFA_ELL_PARTS")));
  // waves per block: 4.  More waves per block would divide the 13 own tiles of cfg3 more evenly (7 waves: 2 rounds)
  // but at ~165 VGPRs only 12 waves fit a CU, i.e. ONE such block instead of three: measured 152 us vs 128 us.
  e->ell_nw = 4;
  if (getenv("SVGPFA_ELL_NW")) e->ell_nw = std::min(8, std::max(1, atoi(getenv("SVGPFA_ELL_NW"))));
  // k_item_bwd: G teams of NT waves (kernels_item_bwd.h).  M <= 32: one team fills a large grid best (more blocks per CU),
  // two or four teams shorten the chain of node tiles of a block on a grid that is one under-filled round anyway
  e->tile_nw = (e->NT >= 3) ? e->NT : (((long long)v.R * v.K >= 512) ? 2 : 4);
  if (getenv("SVGPFA_TIMES_NW")) e->times_nw = std::min(SV_TIMES_MAXTH / 64, std::max(1, atoi(getenv("SVGPFA_TIMES_NW"))));
  if (getenv("SVGPFA_TILE_NW")) {
    const int nw = atoi(getenv("SVGPFA_TILE_NW"));
    if (nw >= 1 && nw <= 4 && (sv_bwd_po(e->NT) ? nw % e->NT == 0 : nw >= 2)) e->tile_nw = nw;
  }
  e->ell_lc[0] = qell_chunk(e, 0);
  e->ell_lc[1] = qell_chunk(e, 1);
  if (e->ell_lc[0] <= 0 || e->ell_lc[1] <= 0)
    FAIL(e, SVGPFA_ERR_UNSUPPORTED, "n_neurons / n_quad too large for the LDS-resident ELL kernels");
  v.ncd = v.R;
  v.nell = v.R * e->ell_parts;
  DALLOC(e, v.gCpart, (size_t)v.ncd * v.N * 2 * v.K); DALLOC(e, v.gdpart, (size_t)v.ncd * v.N);
  DALLOC(e, v.ellq, (size_t)std::max(v.nell, 2 * v.R));
  DALLOC(e, v.ells, (size_t)v.R * v.K); DALLOC(e, v.gthitem, (size_t)v.R * v.K * 2);
  DALLOC(e, v.Mseg, (size_t)v.N * v.K);
  // class path of the spike pass: class table, per-(class, trial) parameters, per-(trial, latent) switch and sums
  DALLOC(e, v.cls, 1);
  HIPCHK(e, hipMemset(v.cls, 0, sizeof(SpikeClassTab)));
  DALLOC(e, v.cls_t0, (size_t)v.K * v.R); DALLOC(e, v.cls_sc, (size_t)v.K * v.R); DALLOC(e, v.cls_H, (size_t)v.K * v.R);
  DALLOC(e, v.ser, (size_t)v.R * v.K);
  HIPCHK(e, hipMemset(v.ser, 0, sizeof(int) * (size_t)v.R * v.K));
  DALLOC(e, v.sitem, (size_t)v.R * v.K);
  HIPCHK(e, hipMemset(v.sitem, 0xff, sizeof(SpikeItem) * (size_t)v.R * v.K));     // H = -1
  DALLOC(e, v.pair_any, v.R);
  HIPCHK(e, hipMemset(v.pair_any, 0xff, sizeof(int) * v.R));
  DALLOC(e, v.Wb, (size_t)v.R * v.K * SV_WCOLS);
  DALLOC(e, v.P0t, (size_t)v.R * v.sumM); DALLOC(e, v.P1t, (size_t)v.R * v.sumM); DALLOC(e, v.Tt, (size_t)v.R * v.K * 2);
  HIPCHK(e, hipMemset(v.P0t, 0, sizeof(double) * (size_t)v.R * v.sumM));   // a recording without spikes never runs the pass
  HIPCHK(e, hipMemset(v.P1t, 0, sizeof(double) * (size_t)v.R * v.sumM));
  HIPCHK(e, hipMemset(v.Tt, 0, sizeof(double) * (size_t)v.R * v.K * 2));
  // small shards (every launch one under-filled round): the per-item combine of the spike moments inside k_item_bwd
  v.comb_fold = ((long long)v.R * v.K <= 2048) ? 1 : 0;
  if (getenv("SVGPFA_COMB_FOLD")) v.comb_fold = atoi(getenv("SVGPFA_COMB_FOLD")) != 0 ? 1 : 0;
  DALLOC(e, v.status, 1);
  HIPCHK(e, hipMemset(v.status, 0, sizeof(int)));
  DALLOC(e, v.fin_ticket, 1);
  HIPCHK(e, hipMemset(v.fin_ticket, 0, sizeof(unsigned)));
  DALLOC(e, v.fin_part, 2 + SV_FIN_NS);
  HIPCHK(e, hipHostMalloc((void**)&e->status_host, sizeof(int)));
  if (getenv("SVGPFA_CLS_FUSED")) e->cls_fused = atoi(getenv("SVGPFA_CLS_FUSED")) != 0 ? 1 : 0;
  {
    // diagnosis / test switches: kernels of a type pair by pair in the spike pass instead of through their cosine series
    const char* rs = getenv("SVGPFA_RBF_SERIES");
    v.rbf_series = (rs && atoi(rs) == 0) ? 0 : 1;
    const char* mp = getenv("SVGPFA_SERIES_MIN_PAIRS");
    if (mp) e->series_min_pairs = atof(mp);
    const char* cp = getenv("SVGPFA_COMB_PAIRS");
    e->comb_pairs = !(cp && atoi(cp) == 0);
    const char* ec = getenv("SVGPFA_ELL_CLS");
    e->ell_cls = !(ec && atoi(ec) == 0);
    const char* br = getenv("SVGPFA_BUILD");
    if (br) e->build_mode = !strcmp(br, "mfma") ? 1 : -1;
    const char* cr = getenv("SVGPFA_COMB_RIDE");
    e->comb_ride = !(cr && atoi(cr) == 0);
    const char* pr = getenv("SVGPFA_PAIR_RIDE");
    e->pair_ride = !(pr && atoi(pr) == 0);
    const char* bw = getenv("SVGPFA_BUILD_WPE");
    if (bw) e->build_wpe = (atoi(bw) == 3) ? 3 : 2;
    const char* fr = getenv("SVGPFA_FRONT");
    e->front = !(fr && atoi(fr) == 0);
    const char* kc = getenv("SVGPFA_KW_CACHE");
    e->kw_cache = !(kc && atoi(kc) == 0);
    const char* ps = getenv("SVGPFA_PERIODIC_SERIES");
    v.per_series = (ps && atoi(ps) == 0) ? 0 : 1;
  }
  {   // the two spike kernels stride a fixed grid over work whose amount is only known on the device: one block per
      // resident wave slot of the chip (a multiple of 8 for the pair pass's XCD-aware item order)
    hipDeviceProp_t prop;
    HIPCHK(e, hipGetDeviceProperties(&prop, device));
    int bc = 0, bp = 0;
    bc = sv_spike_cls_blocks_per_cu();
    HIPCHK(e, hipOccupancyMaxActiveBlocksPerMultiprocessor(&bp, (const void*)k_spike<true, double, true>, 64, 0));
    e->n_cu = prop.multiProcessorCount;
    e->grid_cls = std::max(8, prop.multiProcessorCount * std::max(1, bc));
    e->grid_pair = std::max(8, (prop.multiProcessorCount * std::max(1, bp)) & ~7);
  }

  {
    // The fused ELL kernel (one evaluation of E for both sums) where its grid fills the chip: below that the two roles' finer
    // blocks (ell_parts per trial and role) do.  SVGPFA_ELL_FUSED=0|1 overrides the size rule (never the shape limits).
    const char* ef = getenv("SVGPFA_ELL_FUSED");
    const bool want = ef ? atoi(ef) != 0 : ((long long)v.R * ellf_groups_max >= e->n_cu);
    e->ellf_groups = (want && ellf_can) ? ellf_groups_max : 0;
  }

  int rc = 0;
  if (e->NT == 2) rc = allow_lds(e, (k_build_mfma<2>), sv_build_mfma_lds(2, v.K, true));
  else if (e->NT == 3) rc = allow_lds(e, (k_build_mfma<3>), sv_build_mfma_lds(3, v.K, true));
  else if (e->NT == 4) rc = allow_lds(e, (k_build_mfma<4>), sv_build_mfma_lds(4, v.K, true));
  if (rc) return rc;
  DISPATCH_NT(e->NT, rc = allow_lds(e, (k_prep_mfma<NT_>), lds_prep_mfma(NT_));
              if (!rc) rc = allow_lds(e, (k_front<NT_>), sv_front_lds(NT_));
              if (!rc) rc = allow_lds(e, (k_times_fwd_mfma<NT_>), lds_times_mfma(NT_));
              if (!rc) rc = allow_lds(e, (k_item_bwd<NT_, true>), lds_item_bwd(NT_, e->tile_nw, true));
              if (!rc) rc = allow_lds(e, (k_item_bwd<NT_, false>), lds_item_bwd(NT_, e->tile_nw, false)));
  if (rc) return rc;
  if (e->precision == SVGPFA_PRECISION_MIXED) {
    DISPATCH_KS(e->KKT, rc = allow_lds(e, (k_quad_ell_mfma<KS_, 0, float>), lds_qell_mfma(e, 0));
                if (!rc) rc = allow_lds(e, (k_quad_ell_mfma<KS_, 2, float>), std::max(lds_qell_mfma(e, 0), lds_qell_mfma(e, 1))));
  } else {
    DISPATCH_KS(e->KKT, rc = allow_lds(e, (k_quad_ell_mfma<KS_, 0, double>), lds_qell_mfma(e, 0));
                if (!rc) rc = allow_lds(e, (k_quad_ell_mfma<KS_, 2, double>), std::max(lds_qell_mfma(e, 0), lds_qell_mfma(e, 1))));
  }
  if (rc) return rc;
  if (e->ellf_groups > 0) {
    DISPATCH_KSF(e->KKT, rc = allow_lds(e, (k_quad_ell_fused<KS_>), lds_qell_fused(e->KKT, e->v.Q)));
    if (rc) return rc;
  }
  HIPCHK(e, sv_allow_ell_cls(e->KKT));     // k_ell_cls (small shards) on THIS device
  return SVGPFA_OK;
}

void svgpfa_destroy(svgpfa_engine* e) {
  if (!e) return;
  (void)hipSetDevice(e->device);
  (void)hipDeviceSynchronize();
  for (void* p : e->allocs) (void)hipFree(p);
  if (e->status_host) (void)hipHostFree(e->status_host);
  for (auto& pr : e->ev_pool) {
    (void)hipEventDestroy(pr.first);
    (void)hipEventDestroy(pr.second);
  }
  delete e;
}

const char* svgpfa_last_error(const svgpfa_engine* e) { return e ? e->err.c_str() : "null engine"; }

int svgpfa_set_stream(svgpfa_engine* e, void* s) {
  if (!e) return SVGPFA_ERR_INVALID;
  e->stream = (hipStream_t)s;
  return SVGPFA_OK;
}

int svgpfa_sync(svgpfa_engine* e) {
  if (!e) return SVGPFA_ERR_INVALID;
  HIPCHK(e, hipSetDevice(e->device));   // the caller's thread may have another device current
  HIPCHK(e, hipMemcpyAsync(e->status_host, e->v.status, sizeof(int), hipMemcpyDeviceToHost, e->stream));
  HIPCHK(e, hipStreamSynchronize(e->stream));
  int st = *e->status_host;
  if (st != 0) {
    HIPCHK(e, hipMemsetAsync(e->v.status, 0, sizeof(int), e->stream));
    int item = st - 1;
    char buf[160];
    snprintf(buf, sizeof buf,
             "Kzz + eps*I is not positive definite (local trial %d, latent %d): Cholesky failed",
             item / e->v.K, item % e->v.K);
    e->have_kernels = false;
    FAIL(e, SVGPFA_ERR_NOT_PD, buf);
  }
  return SVGPFA_OK;
}

int svgpfa_set_reg(svgpfa_engine* e, double eps) {
  if (!e) return SVGPFA_ERR_INVALID;
  e->v.reg = eps;
  return SVGPFA_OK;
}

int svgpfa_set_quad(svgpfa_engine* e, const double* t, const double* w) {
  if (!e || !t || !w) return SVGPFA_ERR_INVALID;
  HIPCHK(e, hipSetDevice(e->device));   // the caller's thread may have another device current
  size_t n = sizeof(double) * e->v.R * e->v.Q;
  HIPCHK(e, hipMemcpyAsync((void*)e->v.tq, t, n, hipMemcpyDeviceToDevice, e->stream));
  HIPCHK(e, hipMemcpyAsync((void*)e->v.wq, w, n, hipMemcpyDeviceToDevice, e->stream));
  e->kw_valid = false;
  e->have_quad = true;
  e->have_stats = false;
  return SVGPFA_OK;
}

int svgpfa_set_spikes(svgpfa_engine* e, const int64_t* off, const double* times, const int32_t* neuron) {
  if (!e || !off) return SVGPFA_ERR_INVALID;
  HIPCHK(e, hipSetDevice(e->device));   // the caller's thread may have another device current
  EngineView& v = e->v;
  if (e->have_spikes) FAIL(e, SVGPFA_ERR_STATE, "svgpfa_set_spikes may be called once per engine");
  if (off[0] != 0) FAIL(e, SVGPFA_ERR_INVALID, "spike offsets must start at 0");
  for (int r = 0; r < v.R; ++r)
    if (off[r + 1] < off[r]) FAIL(e, SVGPFA_ERR_INVALID, "spike offsets must be non-decreasing");
  const long long S = off[v.R];
  if (S > 0 && (!times || !neuron)) return SVGPFA_ERR_INVALID;
  if (S > 2000000000LL) FAIL(e, SVGPFA_ERR_UNSUPPORTED, "more than 2e9 spikes on one GPU");
  v.S = (int)S;
  const long long CHUNK = SV_SPIKE_CHUNK;
  std::vector<int32_t> nh((size_t)S);
  if (S) HIPCHK(e, hipMemcpy(nh.data(), neuron, sizeof(int32_t) * S, hipMemcpyDeviceToHost));
  std::vector<double> th((size_t)S);
  if (S) HIPCHK(e, hipMemcpy(th.data(), times, sizeof(double) * S, hipMemcpyDeviceToHost));
  std::vector<double> tlo((size_t)v.R, 0.0), thi((size_t)v.R, 0.0);   // spike-time window per trial (RBF cosine series)
  for (int r = 0; r < v.R; ++r) {
    if (off[r + 1] > off[r]) {
      auto mm = std::minmax_element(th.begin() + off[r], th.begin() + off[r + 1]);
      tlo[r] = *mm.first;
      thi[r] = *mm.second;
    }
  }
  std::vector<long long> seg((size_t)v.R * v.N + 1);
  std::vector<double> cnt(v.N, 0.0);
  std::vector<SpikeChunk> chunks;
  std::vector<unsigned char> seg_cut((size_t)v.R * v.N, 0);   // bit 0: cut by a chunk boundary; bit 1: cut to fill a tile (below)
  std::vector<int> cfirst(v.R + 1), splits;
  for (int r = 0; r < v.R; ++r) {
    long long s0 = off[r], s1 = off[r + 1];
    long long i = s0;
    cfirst[r] = (int)chunks.size();
    SpikeChunk cur{r, 0, s0, 0, -1};
    auto flush = [&]() {
      if (cur.count > 0) chunks.push_back(cur);
      cur.count = 0;
      cur.split = 0;
    };
    for (int n = 0; n < v.N; ++n) {
      seg[(size_t)r * v.N + n] = i;
      long long b = i;
      while (i < s1 && nh[i] == n) { ++i; cnt[n] += 1.0; }
      long long len = i - b;
      if (len == 0) continue;
      if (len > CHUNK) {             // oversize segment: dedicated split chunks
        flush();
        for (long long c = b; c < i; c += CHUNK) {
          SpikeChunk sc{r, (int)std::min<long long>(CHUNK, i - c), c, 1, n};
          splits.push_back((int)chunks.size());
          chunks.push_back(sc);
        }
        cur.start = i;
      } else {
        // dense packing: chunks are filled to CHUNK spikes; a segment that does not fit is CUT -- its tail opens the next
        // chunk, marked as a head continuation (bit 1 of `split`): that chunk's first per-neuron sum goes to gCspkB, so
        // every dC partial still has exactly one writer.  A segment of <= CHUNK spikes is cut at most once.
        if (cur.count == 0) cur.start = b;
        const long long room = CHUNK - cur.count;
        if (len <= room) {
          cur.count += (int)len;
          if (cur.count == CHUNK) flush();
        } else {
          cur.count += (int)room;
          cur.split |= 4;              // this chunk's last segment continues in the next chunk
          seg_cut[(size_t)r * v.N + n] |= 1;
          flush();
          cur = SpikeChunk{r, (int)(len - room), b + room, 2, -1};
        }
      }
    }
    flush();
    if (i != s1)
      FAIL(e, SVGPFA_ERR_INVALID,
           "spikes of a trial must be grouped by non-decreasing neuron index in [0, n_neurons)");
  }
  seg[(size_t)v.R * v.N] = S;
  cfirst[v.R] = (int)chunks.size();
  v.nchunks = (int)chunks.size();

  // ---- class path (kernels_spike_cls.h): rows = (trial, neuron) segments, tiles of <= 64 rows of one trial.  To fill the
  //      trial's last tile, the longest segments are cut in two (each at most once: a cut segment's halves go to the two
  //      dC planes); the rows are then sorted by length so that the lanes of a tile run loops of similar length.
  //      Segments longer than a chunk of the pair pass are oversize in both passes: pieces with slots of their own.
  std::vector<SpikeRow> rows;
  std::vector<long long> row_start;
  std::vector<SpikeTile> tiles;
  std::vector<int> tfirst(v.R + 1), sS_neuron, sS_trial;
  long long tt_len = 0;
  for (int r = 0; r < v.R; ++r) {
    tfirst[r] = (int)tiles.size();
    std::vector<std::pair<SpikeRow, long long>> tr;       // rows of this trial with their first spike
    std::vector<size_t> whole;                            // indices of uncut, cuttable rows
    for (int n = 0; n < v.N; ++n) {
      const long long b = seg[(size_t)r * v.N + n], len = seg[(size_t)r * v.N + n + 1] - b;
      if (len <= 0) continue;
      if (len > CHUNK) {
        for (long long c = b; c < b + len; c += CHUNK) {
          tr.push_back({SpikeRow{(int)std::min<long long>(CHUNK, b + len - c), n, 3, (int)sS_neuron.size()}, c});
          sS_neuron.push_back(n);
          sS_trial.push_back(r);
        }
      } else {
        if (len >= 2) whole.push_back(tr.size());
        tr.push_back({SpikeRow{(int)len, n, 0, 0}, b});
      }
    }
    if (tr.empty()) continue;
    const size_t target = ((tr.size() + 63) / 64) * 64;
    std::sort(whole.begin(), whole.end(), [&](size_t a, size_t b2) {
      return tr[a].first.len != tr[b2].first.len ? tr[a].first.len > tr[b2].first.len : a < b2;
    });
    for (size_t w = 0; w < whole.size() && tr.size() < target; ++w) {
      const size_t i = whole[w];
      const int len = tr[i].first.len, h1 = (len + 1) / 2;
      const int n = tr[i].first.neuron;
      const long long b = tr[i].second;
      tr[i] = {SpikeRow{h1, n, 1, 0}, b};
      tr.push_back({SpikeRow{len - h1, n, 2, 0}, b + h1});
      seg_cut[(size_t)r * v.N + n] |= 2;
    }
    std::stable_sort(tr.begin(), tr.end(), [](const std::pair<SpikeRow, long long>& a, const std::pair<SpikeRow, long long>& b2) {
      return a.first.len > b2.first.len;
    });
    for (size_t i0 = 0; i0 < tr.size(); i0 += 64) {
      const int nr = (int)std::min<size_t>(64, tr.size() - i0);
      SpikeTile tl{r, nr, tr[i0].first.len, (int)rows.size(), tt_len, (int)tiles.size(), 0};
      for (int i = 0; i < nr; ++i) {
        rows.push_back(tr[i0 + i].first);
        row_start.push_back(tr[i0 + i].second);
      }
      tt_len += 64LL * tl.maxlen;
      tiles.push_back(tl);
    }
  }
  tfirst[v.R] = (int)tiles.size();
  v.nrows = (int)rows.size();
  v.ntiles = (int)tiles.size();
  v.nsplitS = (int)sS_neuron.size();
  // k_spike_cls's pass loads rows 0 .. 3 unconditionally and, in every iteration i0 < maxlen, rows i0 + 4 .. i0 + 7: the
  // highest row read of a tile is 4 ceil(maxlen / 4) + 3 <= maxlen + 6, i.e. up to 7 rows past the tile (maxlen = 1), so
  // two prefetch groups of padding behind the last tile (SV_CLS_TT_PAD; engine_common.h proves at compile time, for every
  // row length, that a tile's read-ahead ends within that many doubles behind the tile)
  std::vector<double> tt((size_t)(tt_len + SV_CLS_TT_PAD), 0.0);
  for (const SpikeTile& tl : tiles)
    for (int l = 0; l < tl.nrows; ++l) {
      const SpikeRow& rw = rows[tl.row0 + l];
      const long long b = row_start[tl.row0 + l];
      for (int i = 0; i < rw.len; ++i) tt[(size_t)(tl.tt_off + 64LL * i + l)] = th[(size_t)(b + i)];
    }
  std::vector<SpikeTile> tord(tiles);
  std::stable_sort(tord.begin(), tord.end(), [](const SpikeTile& a, const SpikeTile& b2) { return a.maxlen > b2.maxlen; });
  {
    SpikeRow* drow; SpikeTile* dtile; int* dtf; SpikeTile* dto; double* dtt; int* dsn; int* dst;
    DALLOC(e, drow, rows.size()); DALLOC(e, dtile, tiles.size()); DALLOC(e, dtf, v.R + 1); DALLOC(e, dto, tiles.size());
    DALLOC(e, dtt, tt.size()); DALLOC(e, dsn, sS_neuron.size()); DALLOC(e, dst, sS_trial.size());
    if (!rows.empty()) HIPCHK(e, hipMemcpy(drow, rows.data(), sizeof(SpikeRow) * rows.size(), hipMemcpyHostToDevice));
    if (!tiles.empty()) {
      HIPCHK(e, hipMemcpy(dtile, tiles.data(), sizeof(SpikeTile) * tiles.size(), hipMemcpyHostToDevice));
      HIPCHK(e, hipMemcpy(dto, tord.data(), sizeof(SpikeTile) * tord.size(), hipMemcpyHostToDevice));
    }
    HIPCHK(e, hipMemcpy(dtf, tfirst.data(), sizeof(int) * (v.R + 1), hipMemcpyHostToDevice));
    HIPCHK(e, hipMemcpy(dtt, tt.data(), sizeof(double) * tt.size(), hipMemcpyHostToDevice));
    if (!sS_neuron.empty()) {
      HIPCHK(e, hipMemcpy(dsn, sS_neuron.data(), sizeof(int) * sS_neuron.size(), hipMemcpyHostToDevice));
      HIPCHK(e, hipMemcpy(dst, sS_trial.data(), sizeof(int) * sS_trial.size(), hipMemcpyHostToDevice));
    }
    v.rows = drow; v.tiles = dtile; v.tile_first = dtf; v.tiles_ord = dto; v.tt = dtt;
    v.splitS_neuron = dsn; v.splitS_trial = dst;
    DALLOC(e, v.splitS, (size_t)std::max(v.nsplitS, 1) * v.K);
    HIPCHK(e, hipMemset(v.splitS, 0, sizeof(double) * (size_t)std::max(v.nsplitS, 1) * v.K));
    DALLOC(e, v.Apart, (size_t)std::max(v.ntiles, 1) * v.K * SV_ACOLS);
  }

  double* st; int* sn; long long* so; long long* sg; double* sc; SpikeChunk* dc; int* cf; int* sp;
  DALLOC(e, st, S); DALLOC(e, sn, S); DALLOC(e, so, v.R + 1); DALLOC(e, sg, seg.size());
  DALLOC(e, sc, v.N); DALLOC(e, dc, chunks.size()); DALLOC(e, cf, v.R + 1); DALLOC(e, sp, splits.size());
  if (!splits.empty())
    HIPCHK(e, hipMemcpy(sp, splits.data(), sizeof(int) * splits.size(), hipMemcpyHostToDevice));
  v.split_chunks = sp;
  v.nsplit = (int)splits.size();
  if (S) {
    HIPCHK(e, hipMemcpy(st, times, sizeof(double) * S, hipMemcpyDeviceToDevice));
    HIPCHK(e, hipMemcpy(sn, neuron, sizeof(int32_t) * S, hipMemcpyDeviceToDevice));
  }
  std::vector<long long> off64(off, off + v.R + 1);
  HIPCHK(e, hipMemcpy(so, off64.data(), sizeof(long long) * (v.R + 1), hipMemcpyHostToDevice));
  HIPCHK(e, hipMemcpy(sg, seg.data(), sizeof(long long) * seg.size(), hipMemcpyHostToDevice));
  HIPCHK(e, hipMemcpy(sc, cnt.data(), sizeof(double) * v.N, hipMemcpyHostToDevice));
  if (!chunks.empty())
    HIPCHK(e, hipMemcpy(dc, chunks.data(), sizeof(SpikeChunk) * chunks.size(), hipMemcpyHostToDevice));
  HIPCHK(e, hipMemcpy(cf, cfirst.data(), sizeof(int) * (v.R + 1), hipMemcpyHostToDevice));
  {
    double* dlo; double* dhi;
    DALLOC(e, dlo, v.R); DALLOC(e, dhi, v.R);
    HIPCHK(e, hipMemcpy(dlo, tlo.data(), sizeof(double) * v.R, hipMemcpyHostToDevice));
    HIPCHK(e, hipMemcpy(dhi, thi.data(), sizeof(double) * v.R, hipMemcpyHostToDevice));
    v.spk_lo = dlo; v.spk_hi = dhi;
  }
  {
    unsigned char* dcut;
    DALLOC(e, dcut, seg_cut.size());
    if (!seg_cut.empty()) HIPCHK(e, hipMemcpy(dcut, seg_cut.data(), seg_cut.size(), hipMemcpyHostToDevice));
    v.seg_cut = dcut;
  }
  v.spk_t = st; v.spk_n = sn; v.spk_off = so; v.seg_off = sg; v.spk_cnt = sc; v.chunks = dc; v.chunk_first = cf;
  DALLOC(e, v.P0, (size_t)v.nchunks * v.sumM); DALLOC(e, v.P1, (size_t)v.nchunks * v.sumM);
  DALLOC(e, v.Tpart, (size_t)v.nchunks * v.K * 2);
  DALLOC(e, v.muspk, (size_t)v.K * S); DALLOC(e, v.varspk, (size_t)v.K * S);
  DALLOC(e, v.gCspk, (size_t)v.R * v.N * v.K); DALLOC(e, v.splitPart, (size_t)v.nchunks * v.K);
  DALLOC(e, v.gCspkB, (size_t)v.R * v.N * v.K);
  HIPCHK(e, hipMemset(v.gCspk, 0, sizeof(double) * (size_t)v.R * v.N * v.K));
  HIPCHK(e, hipMemset(v.gCspkB, 0, sizeof(double) * (size_t)v.R * v.N * v.K));
  HIPCHK(e, hipMemset(v.splitPart, 0, sizeof(double) * std::max<size_t>(1, (size_t)v.nchunks * v.K)));
  HIPCHK(e, hipMemset(v.P1, 0, sizeof(double) * std::max<size_t>(1, (size_t)v.nchunks * v.sumM)));
  HIPCHK(e, hipMemset(v.Tpart, 0, sizeof(double) * std::max<size_t>(1, (size_t)v.nchunks * v.K * 2)));
  // Small problems stay with the single pair-by-pair launch: the class path costs three to four more launches per
  // evaluation (class table, W, one or two class kernels), which only pays once the pair loop itself takes tens of
  // microseconds (cfg2, 4e6 pairs: 0.137 ms per evaluation with the class path against 0.11 without)
  e->cls_on = (double)S * (double)v.sumM >= e->series_min_pairs;
  e->have_spikes = true;
  e->have_stats = false;
  e->p0_token = 0;
  e->cls_valid = !e->cls_on;           // nothing to compute when the class path is off: ser stays 0, pair_any stays set
  return SVGPFA_OK;
}

// write_aux: also store Kzz and L (getter path).  Register-resident kernel, 4 / 2 / 1 items per wave.
// snap != nullptr: read Z / theta from the caller's tensors and snapshot them in the same launch.
static void launch_build(svgpfa_engine* e, int write_aux, const SnapTable* snap, bool cls = false) {
  EngineView& v = e->v;
  SnapTable none;
  memset(&none, 0, sizeof none);
  // in front of the item blocks: K blocks compute the periodic kernels' cosine series, R more (cls) the spike pass's class
  // table and per-trial entries
  const int extra = snap ? v.K + (cls ? v.R : 0) : 0;
  const int c = (snap && cls) ? 1 : 0;
  if (e->Mmax <= 16) {
    hipLaunchKernelGGL(k_build_reg<16>, dim3(v.K * ((v.R + 3) / 4) + extra), dim3(64), sv_build_lds(16, v.K, c), e->stream, v,
                       write_aux, snap ? 1 : 0, snap ? *snap : none, c);
  } else if (e->Mmax <= 32 && e->build_mode != 1) {
    // register-resident kernel: the default up to M = 32 (cfg3: 38.6 us against 50-65 us for the blocked kernel at one, two
    // or four items per wave -- the shorter single-wave latency wins).  (Its M = 64 instantiation -- 64 x 64 doubles per
    // item in registers, 1.62 ms at cfg5 against 1.13 -- was removed in round 4.)
    {
      // more blocks than fit the chip at two waves per SIMD: the three-wave instantiation, one round instead of two
      const unsigned nblk = (unsigned)(v.K * ((v.R + 1) / 2) + extra);
      const bool three = e->build_wpe == 3 || (e->build_wpe < 0 && nblk > 8u * (unsigned)e->n_cu && nblk <= 12u * (unsigned)e->n_cu);
      if (three)
        hipLaunchKernelGGL((k_build_reg<32, 3>), dim3(nblk), dim3(64), sv_build_lds(32, v.K, c), e->stream, v, write_aux,
                           snap ? 1 : 0, snap ? *snap : none, c);
      else
        hipLaunchKernelGGL((k_build_reg<32, 1>), dim3(nblk), dim3(64), sv_build_lds(32, v.K, c), e->stream, v, write_aux,
                           snap ? 1 : 0, snap ? *snap : none, c);
    }
  } else {                         // M > 32 (SVGPFA_BUILD=mfma: M > 16): blocked, tile products on the f64 MFMA (k_build_mfma;
                                   // cfg5: 1.13 ms against 1.62 ms for k_build_reg<64>, which holds 64 x 64 doubles per item in registers)
    switch (e->NT) {
      case 2:
        hipLaunchKernelGGL(k_build_mfma<2>, dim3(v.K * ((v.R + sv_build_ipw(2) - 1) / sv_build_ipw(2)) + extra), dim3(64), sv_build_mfma_lds(2, v.K, c), e->stream,
                           v, write_aux, snap ? 1 : 0, snap ? *snap : none, c);
        break;
      case 3:
        hipLaunchKernelGGL(k_build_mfma<3>, dim3(v.K * ((v.R + sv_build_ipw(3) - 1) / sv_build_ipw(3)) + extra), dim3(64), sv_build_mfma_lds(3, v.K, c), e->stream,
                           v, write_aux, snap ? 1 : 0, snap ? *snap : none, c);
        break;
      default:
        hipLaunchKernelGGL(k_build_mfma<4>, dim3(v.K * v.R + extra), dim3(64), sv_build_mfma_lds(4, v.K, c), e->stream, v,
                           write_aux, snap ? 1 : 0, snap ? *snap : none, c);
        break;
    }
  }
}

// Classes of the spike pass's class path for the current kernel snapshot (needs the spikes' windows): they ride in the
// kernel build's launch (launch_build); this separate kernel serves a build that ran before the spikes were set.
static int launch_classes(svgpfa_engine* e) {
  HIPCHK(e, sv_launch_spike_classes(e->v, e->stream));
  e->cls_valid = true;
  return SVGPFA_OK;
}

int svgpfa_build_kernels(svgpfa_engine* e, const double* const* Z, const double* const* theta) {
  if (!e || !Z || !theta) return SVGPFA_ERR_INVALID;
  HIPCHK(e, hipSetDevice(e->device));   // the caller's thread may have another device current
  EngineView& v = e->v;
  SnapTable st;
  memset(&st, 0, sizeof st);
  for (int k = 0; k < v.K; ++k) {
    if (!Z[k] || !theta[k]) FAIL(e, SVGPFA_ERR_INVALID, "NULL Z / theta pointer");
    st.Z[k] = Z[k];
    st.theta[k] = theta[k];
  }
  const bool cls = e->have_spikes && e->cls_on;
  launch_build(e, 0, &st, cls);
  e->aux_valid = false;
  e->kw_valid = false;
  e->p0_token = 0;
  e->cls_valid = cls || !e->cls_on;
  HIPCHK(e, hipGetLastError());
  e->have_kernels = true;
  e->have_stats = false;
  return SVGPFA_OK;
}

int64_t svgpfa_shared_len(const svgpfa_engine* e) {
  if (!e) return 0;
  return 3 + (int64_t)e->v.N * e->v.K + e->v.N + e->v.nth_total;
}

static int fill_ptrs(svgpfa_engine* e, PtrTable& pt, const double* const* mean, const double* const* cholvec,
                     double* const* g_mean, double* const* g_cholvec, double* const* g_Z, unsigned mask) {
  if (!mean || !cholvec) FAIL(e, SVGPFA_ERR_INVALID, "NULL mean / cholvec table");
  if ((mask & 1u) && (!g_mean || !g_cholvec)) FAIL(e, SVGPFA_ERR_INVALID, "variational gradients requested but output tables are NULL");
  if ((mask & 8u) && !g_Z) FAIL(e, SVGPFA_ERR_INVALID, "inducing-point gradients requested but output table is NULL");
  memset(&pt, 0, sizeof pt);
  for (int k = 0; k < e->v.K; ++k) {
    if (!mean[k] || !cholvec[k]) FAIL(e, SVGPFA_ERR_INVALID, "NULL mean / cholvec pointer");
    pt.p[k].mean = mean[k];
    pt.p[k].cholvec = cholvec[k];
    if (mask & 1u) {
      if (!g_mean[k] || !g_cholvec[k]) FAIL(e, SVGPFA_ERR_INVALID, "NULL variational gradient pointer");
      pt.p[k].g_mean = g_mean[k];
      pt.p[k].g_cholvec = g_cholvec[k];
    }
    if (mask & 8u) {
      if (!g_Z[k]) FAIL(e, SVGPFA_ERR_INVALID, "NULL inducing-point gradient pointer");
      pt.p[k].g_Z = g_Z[k];
    }
  }
  return SVGPFA_OK;
}

// ncomb > 0: the combine of the class path's spike moments rides behind the ELL blocks (rider blocks, one item per wave)
static int launch_quad_ell(svgpfa_engine* e, hipStream_t st, const double* C, const double* d, int want_cd, int want_lat,
                           int ncomb = 0, int comb_full = 0) {
  EngineView& v = e->v;
  // role 0 (dC, dd and the ELL value; its column sums give sum E) and role 1 (latent adjoints) share a launch when both are
  // wanted; the latent adjoints alone (E-step closure) are role-1 blocks only, which then write the ELL value themselves
  dim3 gm(v.R * e->ell_parts), bm(64 * e->ell_nw);
  // rider blocks: one item per wave, or two (comb_full bit 2: M <= 32, sv_comb_pairs)
  const int per = (comb_full & 4) ? 2 : 1;
  const unsigned nrb = ncomb > 0 ? (unsigned)((ncomb + per * e->ell_nw - 1) / (per * e->ell_nw)) : 0u;
  const size_t lrid = ncomb > 0 ? (size_t)e->ell_nw * per * SV_COMB_LDS * sizeof(double) : 0;
  const bool f32 = e->precision == SVGPFA_PRECISION_MIXED;
  if (want_lat) {
    const size_t l2 = std::max(std::max(lds_qell_mfma(e, 0), lds_qell_mfma(e, 1)), lrid);
    if (f32) {
      DISPATCH_KS(e->KKT, hipLaunchKernelGGL((k_quad_ell_mfma<KS_, 2, float>), dim3((want_cd ? 2 : 1) * gm.x + nrb), bm, l2, st, v, C, d,
                                             want_cd, 1, e->ell_parts, e->ell_lc[0], e->ell_lc[1], ncomb, comb_full));
    } else {
      DISPATCH_KS(e->KKT, hipLaunchKernelGGL((k_quad_ell_mfma<KS_, 2, double>), dim3((want_cd ? 2 : 1) * gm.x + nrb), bm, l2, st, v, C, d,
                                             want_cd, 1, e->ell_parts, e->ell_lc[0], e->ell_lc[1], ncomb, comb_full));
    }
  } else {
    const size_t l0 = std::max(lds_qell_mfma(e, 0), lrid);
    if (f32) {
      DISPATCH_KS(e->KKT, hipLaunchKernelGGL((k_quad_ell_mfma<KS_, 0, float>), dim3(gm.x + nrb), bm, l0, st, v, C,
                                             d, want_cd, 1, e->ell_parts, e->ell_lc[0], e->ell_lc[1], ncomb, comb_full));
    } else {
      DISPATCH_KS(e->KKT, hipLaunchKernelGGL((k_quad_ell_mfma<KS_, 0, double>), dim3(gm.x + nrb), bm, l0, st, v, C,
                                             d, want_cd, 1, e->ell_parts, e->ell_lc[0], e->ell_lc[1], ncomb, comb_full));
    }
  }
  HIPCHK(e, hipGetLastError());
  return SVGPFA_OK;
}

// dC / dd and the latent adjoints from one evaluation of E (k_quad_ell_fused): R * groups blocks of four waves, riders as above
static int launch_quad_ell_fused(svgpfa_engine* e, hipStream_t st, const EngineView& ve, const double* C, const double* d,
                                 int ncomb, int comb_full) {
  const int per = (comb_full & 4) ? 2 : 1;
  const unsigned nrb = ncomb > 0 ? (unsigned)((ncomb + per * SV_ELLF_NW - 1) / (per * SV_ELLF_NW)) : 0u;
  DISPATCH_KSF(e->KKT, hipLaunchKernelGGL((k_quad_ell_fused<KS_>), dim3((unsigned)(ve.R * e->ellf_groups) + nrb), dim3(64 * SV_ELLF_NW),
                                          lds_qell_fused(e->KKT, e->v.Q), st, ve, C, d, e->ellf_groups, ncomb, comb_full));
  HIPCHK(e, hipGetLastError());
  return SVGPFA_OK;
}

static int launch_times(svgpfa_engine* e, hipStream_t st, const double* times, const long long* toff, int nper, double* mu,
                        double* var, long long os_r, long long os_t, long long os_k, int want_var) {
  EngineView& v = e->v;
  DISPATCH_NT(e->NT, hipLaunchKernelGGL((k_times_fwd_mfma<NT_>), dim3(v.R * v.K), dim3(64 * e->times_nw),
                                        lds_times_mfma(NT_), st, v, times, toff, nper, mu, var, os_r, os_t, os_k,
                                        want_var));
  HIPCHK(e, hipGetLastError());
  return SVGPFA_OK;
}

// one wavefront per 16-column panel; want_w: the class path's W rides along (needs a valid class table)
static void launch_prep(svgpfa_engine* e, const PtrTable& pt, bool want_w = false) {
  EngineView& v = e->v;
  DISPATCH_NT(e->NT, hipLaunchKernelGGL(k_prep_mfma<NT_>, dim3(v.R * v.K), dim3(64 * NT_), lds_prep_mfma(NT_), e->stream, v, pt,
                                        want_w ? 1 : 0));
}

// k_prep_mfma + the quadrature pass of k_times_fwd_mfma in one launch (svgpfa_elbo)
static void launch_front(svgpfa_engine* e, const PtrTable& pt, bool want_w) {
  EngineView& v = e->v;
  DISPATCH_NT(e->NT, hipLaunchKernelGGL(k_front<NT_>, dim3(v.R * v.K), dim3(256), sv_front_lds(NT_), e->stream, v, pt,
                                        want_w ? 1 : 0));
}

// quadrature statistics into the engine's [r][k][q] buffers
static int launch_times_quad(svgpfa_engine* e, hipStream_t st) {
  EngineView& v = e->v;
  return launch_times(e, st, v.tq, nullptr, v.Q, v.muq, v.varq, (long long)v.K * v.Q, 1LL, (long long)v.Q, 1);
}

// svgpfa_enable_timing: bracket a kernel with a hipEvent pair on the stream it is launched on.  Pairs are reused: a query
// (svgpfa_kernel_ms) folds every recorded pair into per-slot running totals and frees the pool, and so does a full pool
// (after waiting for its streams), so a long timed run holds at most SV_EV_POOL_MAX pairs.  Returns the pair's index
// (-1: timing off), never a pointer into the growing vector.
#define SV_EV_POOL_MAX 4096
static int fold_events(svgpfa_engine* e) {
  HIPCHK(e, hipStreamSynchronize(e->stream));
  for (size_t i = 0; i < e->ev_used; ++i) {
    float ms = 0.f;
    HIPCHK(e, hipEventElapsedTime(&ms, e->ev_pool[i].first, e->ev_pool[i].second));
    e->ev_tot[e->ev_slot[i]] += ms;
    e->ev_cnt[e->ev_slot[i]] += 1;
  }
  e->ev_used = 0;
  return SVGPFA_OK;
}

static int timing_begin(svgpfa_engine* e, int slot, hipStream_t st) {
  if (!e->timing) return -1;
  // sampled: an evaluation carries at most ONE pair -- k_item_bwd's on evaluations 0, P, 2P ..., the spike pass's half a period
  // later (an event record is a packet of its own on the stream: two pairs per evaluation cost 12.5 us of a 0.524-ms
  // evaluation, tests/tools/timing_cost.py)
  if (e->timing_period > 1) {
    const unsigned long long ph = (e->timing_eval - 1) % (unsigned long long)e->timing_period;
    if (ph != (slot == 1 ? 0ull : (unsigned long long)(e->timing_period / 2))) return -1;
  }
  if (e->ev_used >= SV_EV_POOL_MAX && fold_events(e) != SVGPFA_OK) return -1;
  if (e->ev_used == e->ev_pool.size()) {
    std::pair<hipEvent_t, hipEvent_t> pr;
    if (hipEventCreate(&pr.first) != hipSuccess || hipEventCreate(&pr.second) != hipSuccess) return -1;
    e->ev_pool.push_back(pr);
    e->ev_slot.push_back(0);
  }
  const int idx = (int)e->ev_used++;
  e->ev_slot[idx] = slot;
  (void)hipEventRecord(e->ev_pool[idx].first, st);
  return idx;
}
static void timing_end(svgpfa_engine* e, int idx, hipStream_t st) {
  if (idx >= 0) (void)hipEventRecord(e->ev_pool[idx].second, st);
}

// pair-by-pair pass over the (chunk, latent) items the class path does not serve (early exit per trial when it serves all)
static void launch_pair_pass(svgpfa_engine* e, hipStream_t sk, bool full, bool want_c, const double* C) {
  EngineView& v = e->v;
  const dim3 bs(64);
  const long long npair = (long long)v.nchunks * v.K;
  const dim3 gp((unsigned)std::min<long long>((npair + 7) & ~7LL, e->grid_pair));
  if (npair <= 0) return;
  if (e->precision == SVGPFA_PRECISION_MIXED) {
    if (full && !want_c) hipLaunchKernelGGL((k_spike<true, float, false>), gp, bs, 0, sk, v, C);
    else if (full) hipLaunchKernelGGL((k_spike<true, float>), gp, bs, 0, sk, v, C);
    else hipLaunchKernelGGL((k_spike<false, float>), gp, bs, 0, sk, v, C);
  } else {
    if (full && !want_c) hipLaunchKernelGGL((k_spike<true, double, false>), gp, bs, 0, sk, v, C);
    else if (full) hipLaunchKernelGGL((k_spike<true>), gp, bs, 0, sk, v, C);
    else hipLaunchKernelGGL((k_spike<false>), gp, bs, 0, sk, v, C);
  }
}

// The spike-time pass on stream sk.  full: Z / theta gradient sums wanted too; want_c: the spike means' per-neuron sums (dC)
// wanted (only with full); have_w: k_prep made the class path's W (else k_spike_pre does, here).  Class path (cosine
// series, kernels_spike_cls.h) for the (trial, latent) items it serves, the pair-by-pair pass (kernels_spike.h) for the
// rest; both stride fixed grids over work lists that live on the device.
// one launch for all classes of the class path (shards), or the classes with time-weighted moments (periodic, when Z / theta
// gradients are wanted) in a kernel of their own
// bit 2 of the combine's flags: two (trial, latent) items per wave (every M <= 32: spike_cls_common.h spike_comb_pair);
// SVGPFA_COMB_PAIRS=0 keeps one
static int sv_comb_pairs(const svgpfa_engine* e) { return (e->NT <= 2 && e->comb_pairs) ? 4 : 0; }
static bool cls_one_launch(const svgpfa_engine* e) {      // ... and, with it, the small-shard launch k_ell_cls
  return e->cls_fused == 1 || (e->cls_fused < 0 && (long long)e->v.R * e->v.K <= 4096);
}
// The class kernels of their own launch: both kinds of class in ONE kernel at every size unless SVGPFA_CLS_FUSED=0 (round 5:
// at the north-star size 52 us against 32 + 30 for the two launches, measured once the time-weighted moments were carried
// for the kernel-parameter gradient only.  Its workers take their (tile, class) items by stride: handed out by ticket --
// one counter 245 us, 60 ns per atomic serialised across the XCDs; 256 counters, one per group of eight workers, 53.5 us --
// they are no faster, so the SIMDs' unequal item counts are not what the kernel waits for.)
static bool cls_single_kernel(const svgpfa_engine* e) { return e->cls_fused != 0; }

// The pair-by-pair pass rides in the class kernel's launch (fp64, class path running: cls_pair_rider) instead of taking a
// launch of its own; SVGPFA_PAIR_RIDE=0 keeps the launch.
static bool pair_rides(const svgpfa_engine* e) {
  return e->pair_ride && e->cls_on && e->v.ntiles > 0 && e->precision == SVGPFA_PRECISION_FP64;
}

static int launch_spike_pass(svgpfa_engine* e, hipStream_t sk, bool full, bool want_c, const double* C, bool have_w,
                              bool comb_rides = false, bool want_tw = true) {
  const int fl = (full ? 1 : 0) | ((full && want_tw) ? 2 : 0);       // the class kernels' / combine's `full` bit set
  EngineView& v = e->v;
  const unsigned gc = (unsigned)std::min<long long>((long long)v.ntiles * v.K, e->grid_cls);
  const bool ride = pair_rides(e);
  if (v.ntiles > 0 && e->cls_on) {
    const bool fused = cls_single_kernel(e);
    const bool tw = !fused && (fl & 2) && v.ntype[SV_PERIODIC] > 0;      // a launch for the classes with time-weighted moments
    const int dc = (full && want_c) ? 1 : 0;
    if (dc && !have_w) HIPCHK(e, sv_launch_spike_pre(v, sk));
    // a rider needs workers for the pair items even when the class items are few: the grid of the pair pass's own launch
    const long long npair = (long long)v.nchunks * v.K;
    const unsigned gr = ride ? std::max(gc, (unsigned)std::min<long long>((npair + 7) & ~7LL, e->grid_cls)) : gc;
    if (fused) {
      HIPCHK(e, sv_launch_spike_cls(v, C, 0, dc, (dc || full) ? fl : 0, ride ? 1 : 0, gr, sk));
    } else {
      if (tw) HIPCHK(e, sv_launch_spike_cls(v, C, 4, dc, fl, 0, gc, sk));
      HIPCHK(e, sv_launch_spike_cls(v, C, 2, dc, (dc || full) ? fl : 0, ride ? 1 : 0, gr, sk));
    }
  }
  if (!v.comb_fold && e->cls_on && !comb_rides) HIPCHK(e, sv_launch_spike_comb(v, fl | sv_comb_pairs(e), sk));
  if (!ride) {
    launch_pair_pass(e, sk, full, want_c, C);
    HIPCHK(e, hipGetLastError());
  }
  return SVGPFA_OK;
}

static int check_ready(svgpfa_engine* e) {
  if (!e->have_quad) FAIL(e, SVGPFA_ERR_STATE, "svgpfa_set_quad has not been called");
  if (!e->have_spikes) FAIL(e, SVGPFA_ERR_STATE, "svgpfa_set_spikes has not been called");
  if (!e->have_kernels) FAIL(e, SVGPFA_ERR_STATE, "svgpfa_build_kernels has not been called (or its Cholesky failed)");
  return SVGPFA_OK;
}

// The launches of one ELBO evaluation, all on e->stream.
struct ElboPlan {
  unsigned mask;
  bool want_lat, full, run_spike;
};
static int enqueue_elbo(svgpfa_engine* e, const PtrTable& pt, const ElboPlan& pl, const double* C, const double* d,
                        double* shared) {
  EngineView& v = e->v;
  const unsigned mask = pl.mask;
  int rc;
  if (!e->cls_valid && (rc = launch_classes(e))) return rc;   // spikes set after the last kernel build
  hipStream_t sm = e->stream;
  if (e->timing) ++e->timing_eval;
  // the class path's W rides in k_prep when this evaluation's spike pass wants dC
  const bool have_w = pl.run_spike && pl.full && (mask & 2u) && e->cls_on && v.ntiles > 0;
  // whitened parameters (ut, a, Vt, KL), the class path's W and the quadrature statistics: one launch (k_front);
  // SVGPFA_FRONT=0 keeps k_prep_mfma and k_times_fwd_mfma as launches of their own
  // E-step evaluations (variational gradients only, or the value alone) keep the whitened kernel tiles of the quadrature
  // nodes across evaluations while the kernels and the nodes stand: the first one behind a kernel build stores them (k_front),
  // every one loads them in k_item_bwd, later ones in k_front too -- the reference's E-step closure runs on cached Ktz as well
  // (stats/svEM.py:218-223).  2 x 266 MB at the north-star size (2 x 4.3 GB at cfg5), allocated at the first such evaluation.
  int kw_mode = 0;
  if (e->front && e->kw_cache && !pl.full && !(mask & SV_MASK_STATS)) {
    if (e->kw_valid) kw_mode = 2;
    else if (mask & 1u) kw_mode = 1;
    v.kw_half = (long long)v.R * v.K * ((v.Q + 15) / 16) * e->NT * 256;      // two images of the tiles: k_front's and k_item_bwd's
    if (kw_mode && !v.Kwc) DALLOC(e, v.Kwc, (size_t)2 * v.kw_half);
    if (kw_mode == 1) e->kw_valid = true;
  }
  v.kw_mode = kw_mode;
  v.kw_use = (kw_mode != 0 && (mask & 1u)) ? 1 : 0;
  v.kw_nt = e->NT;
  if (e->front) launch_front(e, pt, have_w);
  else launch_prep(e, pt, have_w);
  // ---- spike pass (fp64 VALU bound): needs only a = Kzz^-1 m (and W) from k_prep
  // Small shards (one launch for all classes, fp64, quadrature statistics already made by k_front): the ELL blocks and the
  // class items share ONE launch (k_ell_cls) -- both follow k_front, both precede k_item_bwd, neither fills the chip.
  // (k_ell_cls runs its ELL blocks with four waves: its dynamic LDS is sized for e->ell_nw, so any other count keeps the launches apart)
  const bool merged = e->ell_cls && e->front && pl.run_spike && e->cls_on && v.ntiles > 0 && cls_one_launch(e) &&
                      e->precision == SVGPFA_PRECISION_FP64 && e->ell_nw == 4;
  // Blocks per trial of the merged launch's ELL part: its blocks and the class workers' blocks hold the same 64 KB of LDS, two
  // per CU, so the ELL part may take at most half of the chip's 2 x CUs slots or the class items queue up behind it (measured
  // with the engine's usual four blocks per trial: 504 ELL blocks, 48.6 us = the two kernels one after the other)
  int parts_m = e->ell_parts;
  const int ell_roles = (pl.want_lat && (mask & 2u)) ? 2 : 1;       // (latent adjoints alone or dC / dd alone: one role's blocks)
  while (merged && parts_m > 1 && (long long)v.R * parts_m * ell_roles > e->n_cu) parts_m >>= 1;
  if (merged && getenv("SVGPFA_ELLCLS_PARTS")) parts_m = std::max(1, std::min(e->ell_parts, atoi(getenv("SVGPFA_ELLCLS_PARTS"))));   // (experiments)
  // Full size (two class kernels, the combine not folded into k_item_bwd): the combine rides behind the ELL blocks of
  // k_quad_ell_mfma, which follows the class kernels on this stream anyway (SVGPFA_COMB_RIDE=0: a launch of its own)
  const bool comb_rides = e->comb_ride && !merged && pl.run_spike && e->cls_on && v.ntiles > 0 && !v.comb_fold;
  EngineView ve = v;               // this evaluation's view: the number of ELL partials follows the blocks per trial
  if (merged) ve.nell = v.R * parts_m;
  // dC / dd and latent adjoints both wanted, at a size whose grid fills the chip: the fused ELL kernel, whose neuron groups
  // each write a plane of the latent adjoints (k_item_bwd adds them) and an ELL partial
  const bool ell_fused = !merged && e->ellf_groups > 0 && (mask & 2u) && pl.want_lat;
  if (ell_fused) {
    ve.nell = v.R * e->ellf_groups;
    ve.glat2 = (e->ellf_groups == 2) ? (long long)v.R * v.Q * v.K : 0;
  }
  if (pl.run_spike) {
    const int evp = timing_begin(e, 0, sm);
    if (merged) {
      const bool full = pl.full, want_c = (mask & 2u) != 0;
      const int fl = (full ? 1 : 0) | ((full && (mask & 4u)) ? 2 : 0);      // bit 1: time-weighted moments (kernel-parameter gradient)
      const int dc = (full && want_c) ? 1 : 0;          // (have_w holds whenever dc does: k_front made W)
      const unsigned workers = (unsigned)std::min<long long>((long long)v.ntiles * v.K, e->grid_cls);
      const size_t lds_ell = pl.want_lat ? std::max(lds_qell_mfma(e, 0), lds_qell_mfma(e, 1)) : lds_qell_mfma(e, 0);
      const bool ride = pair_rides(e);
      HIPCHK(e, sv_launch_ell_cls(ve, C, d, e->KKT, dc, want_c ? 1 : 0, pl.want_lat ? 1 : 0, parts_m, e->ell_lc[0], e->ell_lc[1],
                                  fl, ride ? 1 : 0, workers, lds_ell, sm));
      if (!v.comb_fold) HIPCHK(e, sv_launch_spike_comb(v, fl | sv_comb_pairs(e), sm));
      if (!ride) launch_pair_pass(e, sm, full, want_c, C);
    } else {
      if ((rc = launch_spike_pass(e, sm, pl.full, (mask & 2u) != 0, C, have_w, comb_rides, (mask & 4u) != 0))) return rc;
    }
    timing_end(e, evp, sm);
  }
  // ---- quadrature chain (f64 MFMA): posterior statistics at the nodes -> ELL term and its adjoints
  if (!e->front && (rc = launch_times_quad(e, sm))) return rc;
  const int comb_fl = (pl.full ? 1 : 0) | ((pl.full && (mask & 4u)) ? 2 : 0) | sv_comb_pairs(e);
  if (ell_fused) {
    if ((rc = launch_quad_ell_fused(e, sm, ve, C, d, comb_rides ? v.R * v.K : 0, comb_fl))) return rc;
  } else if (!merged && (rc = launch_quad_ell(e, sm, C, d, (mask & 2u) ? 1 : 0, pl.want_lat ? 1 : 0, comb_rides ? v.R * v.K : 0,
                                              comb_fl))) return rc;
  // ---- per-item adjoints + backward + the fixed-order cross-trial reductions in one kernel
  const int evb = timing_begin(e, 1, sm);
  if (mask & (4u | 8u)) {
    DISPATCH_NT(e->NT, hipLaunchKernelGGL((k_item_bwd<NT_, true>), dim3(v.R * v.K), dim3(64 * e->tile_nw),
                                          lds_item_bwd(NT_, e->tile_nw, true), sm, ve, pt, mask, C, d, shared));
  } else {
    DISPATCH_NT(e->NT, hipLaunchKernelGGL((k_item_bwd<NT_, false>), dim3(v.R * v.K), dim3(64 * e->tile_nw),
                                          lds_item_bwd(NT_, e->tile_nw, false), sm, ve, pt, mask, C, d, shared));
  }
  timing_end(e, evb, sm);
  hipLaunchKernelGGL(k_fin_items, dim3(v.nth_total + SV_FIN_NS), dim3(256), 0, sm, ve, mask, shared);
  HIPCHK(e, hipGetLastError());
  return SVGPFA_OK;
}

int svgpfa_elbo(svgpfa_engine* e, const double* const* mean, const double* const* cholvec, const double* C,
                const double* d, unsigned mask, double* shared, double* const* g_mean,
                double* const* g_cholvec, double* const* g_Z) {
  if (!e || !C || !d || !shared) return SVGPFA_ERR_INVALID;
  HIPCHK(e, hipSetDevice(e->device));
  int rc;
  if ((rc = check_ready(e))) return rc;
  EngineView& v = e->v;
  PtrTable pt;
  if ((rc = fill_ptrs(e, pt, mean, cholvec, g_mean, g_cholvec, g_Z, mask))) return rc;
  ElboPlan pl;
  pl.mask = mask;
  pl.want_lat = (mask & (1u | 4u | 8u)) != 0;
  pl.full = (mask & (2u | 4u | 8u)) != 0;
  // A value-only evaluation reuses the spike pass's sums when the caller vouches (token) that C is unchanged since the pass
  // that produced them.  Only a value-only pass leaves reusable sums: the full pass accumulates them with different
  // roundings, and cached and un-cached evaluations are to stay bit-identical
  const bool reuse_p0 = !pl.full && e->emb_token != 0 && e->p0_token == e->emb_token;
  pl.run_spike = v.nchunks > 0 && !reuse_p0;
  if (pl.run_spike) e->p0_token = pl.full ? 0 : e->emb_token;
  return enqueue_elbo(e, pt, pl, C, d, shared);
}

int svgpfa_latent_stats(svgpfa_engine* e, const double* const* mean, const double* const* cholvec,
                        double* mu_quad, double* var_quad, double* mu_spk, double* var_spk) {
  if (!e) return SVGPFA_ERR_INVALID;
  HIPCHK(e, hipSetDevice(e->device));
  int rc;
  if ((rc = check_ready(e))) return rc;
  EngineView& v = e->v;
  PtrTable pt;
  if ((rc = fill_ptrs(e, pt, mean, cholvec, nullptr, nullptr, nullptr, 0))) return rc;
  launch_prep(e, pt);
  if ((rc = launch_times_quad(e, e->stream))) return rc;
  if ((rc = launch_times(e, e->stream, v.spk_t, v.spk_off, 0, v.muspk, v.varspk, 0LL, 1LL, (long long)v.S, 1))) return rc;
  {   // lanes per segment: about a quarter of the mean number of spikes per (trial, neuron), a power of two <= 32
    const long long seg = (long long)v.S / std::max(1LL, (long long)v.R * v.N);
    int gs = 1;
    while (gs < 32 && 4 * gs < seg) gs *= 2;
    hipLaunchKernelGGL(k_mseg, dim3((v.N * v.K + SV_FS_OUT - 1) / SV_FS_OUT), dim3(256), 0, e->stream, v, gs);
  }
  {   // the engine keeps [r][k][q]; the caller gets the reference's [r][q][k]
    const long long rqk = (long long)v.R * v.Q * v.K;
    const unsigned g = (unsigned)((rqk + 255) / 256);
    if (mu_quad) hipLaunchKernelGGL(k_transpose_b, dim3(g), dim3(256), 0, e->stream, v.muq, mu_quad, (long long)v.R, (long long)v.K, (long long)v.Q);
    if (var_quad) hipLaunchKernelGGL(k_transpose_b, dim3(g), dim3(256), 0, e->stream, v.varq, var_quad, (long long)v.R, (long long)v.K, (long long)v.Q);
  }
  long long tot = (long long)v.S * v.K;
  if (tot > 0) {   // [K][S] -> [S][K]
    unsigned g = (unsigned)((tot + 255) / 256);
    if (mu_spk) hipLaunchKernelGGL(k_transpose_b, dim3(g), dim3(256), 0, e->stream, v.muspk, mu_spk, 1LL, (long long)v.K, (long long)v.S);
    if (var_spk) hipLaunchKernelGGL(k_transpose_b, dim3(g), dim3(256), 0, e->stream, v.varspk, var_spk, 1LL, (long long)v.K, (long long)v.S);
  }
  HIPCHK(e, hipGetLastError());
  e->have_stats = true;
  return SVGPFA_OK;
}

int svgpfa_freeze_stats(svgpfa_engine* e, const double* const* mean, const double* const* cholvec, const double* C) {
  if (!e || !C) return SVGPFA_ERR_INVALID;
  HIPCHK(e, hipSetDevice(e->device));
  int rc;
  if ((rc = check_ready(e))) return rc;
  if (e->precision == SVGPFA_PRECISION_MIXED)   // the fp32 pair loops would carry their rounding into the frozen sums
    return svgpfa_latent_stats(e, mean, cholvec, nullptr, nullptr, nullptr, nullptr);
  EngineView& v = e->v;
  PtrTable pt;
  if ((rc = fill_ptrs(e, pt, mean, cholvec, nullptr, nullptr, nullptr, 0))) return rc;
  if (!e->cls_valid && (rc = launch_classes(e))) return rc;
  launch_prep(e, pt);
  if ((rc = launch_times_quad(e, e->stream))) return rc;
  // the per-neuron sums of the spike means are the spike part of dC, which the spike-time pass makes (cosine series:
  // ~0.3 ms at the north-star size against 2 ms for the posterior mean and variance at every spike)
  if (v.nchunks > 0 && (rc = launch_spike_pass(e, e->stream, true, true, C, false, false, false))) return rc;     // (dC rows only: no time-weighted moments)
  e->p0_token = 0;                     // that pass overwrote the reusable spike sums
  hipLaunchKernelGGL(k_mseg_from_dc, dim3((v.N * v.K + SV_FS_OUT - 1) / SV_FS_OUT), dim3(256), 0, e->stream, v);
  HIPCHK(e, hipGetLastError());
  e->have_stats = true;
  return SVGPFA_OK;
}

int svgpfa_kl(svgpfa_engine* e, double* out) {
  if (!e || !out) return SVGPFA_ERR_INVALID;
  HIPCHK(e, hipSetDevice(e->device));
  if (!e->have_kernels) FAIL(e, SVGPFA_ERR_STATE, "svgpfa_build_kernels has not been called");
  hipLaunchKernelGGL(k_sum_kl, dim3(1), dim3(256), 0, e->stream, e->v, out);
  HIPCHK(e, hipGetLastError());
  return SVGPFA_OK;
}

int svgpfa_latent_stats_backward(svgpfa_engine* e, const double* const* mean, const double* const* cholvec,
                                 const double* g_mu_quad, const double* g_var_quad, const double* g_mu_spk,
                                 const double* g_var_spk, unsigned mask, double* shared, double* const* g_mean,
                                 double* const* g_cholvec, double* const* g_Z) {
  if (!e || !g_mu_quad || !g_var_quad || !shared) return SVGPFA_ERR_INVALID;
  HIPCHK(e, hipSetDevice(e->device));
  int rc;
  if ((rc = check_ready(e))) return rc;
  EngineView& v = e->v;
  if (v.S > 0 && (!g_mu_spk || !g_var_spk)) FAIL(e, SVGPFA_ERR_INVALID, "spike adjoints missing");
  mask &= (1u | 4u | 8u);
  PtrTable pt;
  if ((rc = fill_ptrs(e, pt, mean, cholvec, g_mean, g_cholvec, g_Z, mask))) return rc;
  if (!v.gmuspk) {   // first use: [K][S] adjoint buffers
    DALLOC(e, v.gmuspk, (size_t)v.K * std::max(v.S, 1));
    DALLOC(e, v.gvarspk, (size_t)v.K * std::max(v.S, 1));
  }
  launch_prep(e, pt);
  {   // the caller's adjoints in the reference's layouts -> the engine's ([R][Q][K] -> [R][K][Q], [S][K] -> [K][S])
    const long long rqk = (long long)v.R * v.Q * v.K, sk = (long long)v.S * v.K;
    const unsigned g = (unsigned)((rqk + 255) / 256);
    hipLaunchKernelGGL(k_transpose_b, dim3(g), dim3(256), 0, e->stream, g_mu_quad, v.gmuq, (long long)v.R, (long long)v.Q, (long long)v.K);
    hipLaunchKernelGGL(k_transpose_b, dim3(g), dim3(256), 0, e->stream, g_var_quad, v.gvarq, (long long)v.R, (long long)v.Q, (long long)v.K);
    if (sk > 0) {
      const unsigned gs = (unsigned)((sk + 255) / 256);
      hipLaunchKernelGGL(k_transpose_b, dim3(gs), dim3(256), 0, e->stream, g_mu_spk, v.gmuspk, 1LL, (long long)v.S, (long long)v.K);
      hipLaunchKernelGGL(k_transpose_b, dim3(gs), dim3(256), 0, e->stream, g_var_spk, v.gvarspk, 1LL, (long long)v.S, (long long)v.K);
    }
  }
  v.kw_mode = v.kw_use = 0;            // (the whitened-tile cache serves svgpfa_elbo's E-step evaluations only)
  const unsigned m2 = mask | SV_MASK_STATS;
  if (mask & (4u | 8u)) {
    DISPATCH_NT(e->NT, hipLaunchKernelGGL((k_item_bwd<NT_, true>), dim3(v.R * v.K), dim3(64 * e->tile_nw),
                                          lds_item_bwd(NT_, e->tile_nw, true), e->stream, v, pt, m2, (const double*)nullptr,
                                          (const double*)nullptr, shared));
  } else {
    DISPATCH_NT(e->NT, hipLaunchKernelGGL((k_item_bwd<NT_, false>), dim3(v.R * v.K), dim3(64 * e->tile_nw),
                                          lds_item_bwd(NT_, e->tile_nw, false), e->stream, v, pt, m2, (const double*)nullptr,
                                          (const double*)nullptr, shared));
  }
  hipLaunchKernelGGL(k_fin_items, dim3(v.nth_total + SV_FIN_NS), dim3(256), 0, e->stream, v, m2, shared);
  HIPCHK(e, hipGetLastError());
  e->p0_token = 0;
  return SVGPFA_OK;
}

int svgpfa_ell_given_stats(svgpfa_engine* e, const double* C, const double* d, int want_grad, double* out) {
  if (!e || !C || !d || !out) return SVGPFA_ERR_INVALID;
  HIPCHK(e, hipSetDevice(e->device));
  if (!e->have_stats) FAIL(e, SVGPFA_ERR_STATE, "svgpfa_latent_stats has not been called since the last data / kernel change");
  EngineView& v = e->v;
  int rc;
  if ((rc = launch_quad_ell(e, e->stream, C, d, want_grad ? 1 : 0, 0))) return rc;
  int nout = v.N * v.K + v.N;
  hipLaunchKernelGGL(k_finalize_stats, dim3((nout + SV_FS_OUT - 1) / SV_FS_OUT + 1), dim3(256), 0, e->stream, v, C, d, want_grad,
                     out);
  HIPCHK(e, hipGetLastError());
  return SVGPFA_OK;
}

int svgpfa_predict(svgpfa_engine* e, const double* const* mean, const double* const* cholvec,
                   const double* times, int32_t T, double* mu, double* var) {
  if (!e || !times || !mu || T <= 0) return SVGPFA_ERR_INVALID;
  HIPCHK(e, hipSetDevice(e->device));
  if (!e->have_kernels) FAIL(e, SVGPFA_ERR_STATE, "svgpfa_build_kernels has not been called");
  EngineView& v = e->v;
  PtrTable pt;
  int rc;
  if ((rc = fill_ptrs(e, pt, mean, cholvec, nullptr, nullptr, nullptr, 0))) return rc;
  launch_prep(e, pt);
  if ((rc = launch_times(e, e->stream, times, nullptr, (int)T, mu, var ? var : mu, (long long)T * v.K, (long long)v.K, 1LL,
                         var ? 1 : 0)))
    return rc;
  HIPCHK(e, hipGetLastError());
  e->have_stats = false;  // k_prep overwrote a / W with these parameters; stats must be recomputed
  return SVGPFA_OK;
}

static int get_mm(svgpfa_engine* e, int k, const double* src, double* out) {
  if (!e || !out || k < 0 || k >= e->v.K) return SVGPFA_ERR_INVALID;
  if (!e->have_kernels) FAIL(e, SVGPFA_ERR_STATE, "svgpfa_build_kernels has not been called");
  if (!e->aux_valid && e->Mmax <= 64) {   // Kzz / L are materialised on demand from the current snapshot
    launch_build(e, 1, nullptr);
    HIPCHK(e, hipGetLastError());
    e->aux_valid = true;
  }
  const LatentMeta& m = e->meta[k];
  HIPCHK(e, hipMemcpyAsync(out, src + m.offMM, sizeof(double) * e->v.R * m.M * m.M, hipMemcpyDeviceToDevice, e->stream));
  return SVGPFA_OK;
}
int svgpfa_get_kzz(svgpfa_engine* e, int32_t k, double* out) { return get_mm(e, k, e ? e->v.Kzz : nullptr, out); }
int svgpfa_get_kzz_chol(svgpfa_engine* e, int32_t k, double* out) { return get_mm(e, k, e ? e->v.L : nullptr, out); }

int svgpfa_get_ktz_quad(svgpfa_engine* e, int32_t k, double* out) {
  if (!e || !out || k < 0 || k >= e->v.K) return SVGPFA_ERR_INVALID;
  if (!e->have_kernels || !e->have_quad) FAIL(e, SVGPFA_ERR_STATE, "need set_quad and build_kernels first");
  EngineView& v = e->v;
  long long ntot = (long long)v.R * v.Q, tot = ntot * e->meta[k].M;
  hipLaunchKernelGGL(k_ktz, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, e->stream, v, (int)k, v.tq,
                     (const long long*)nullptr, v.Q, ntot, out);
  HIPCHK(e, hipGetLastError());
  return SVGPFA_OK;
}

int svgpfa_get_ktz_spikes(svgpfa_engine* e, int32_t k, double* out) {
  if (!e || !out || k < 0 || k >= e->v.K) return SVGPFA_ERR_INVALID;
  if (!e->have_kernels || !e->have_spikes) FAIL(e, SVGPFA_ERR_STATE, "need set_spikes and build_kernels first");
  EngineView& v = e->v;
  long long ntot = v.S, tot = ntot * e->meta[k].M;
  if (tot == 0) return SVGPFA_OK;
  hipLaunchKernelGGL(k_ktz, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, e->stream, v, (int)k, v.spk_t,
                     v.spk_off, 0, ntot, out);
  HIPCHK(e, hipGetLastError());
  return SVGPFA_OK;
}

int svgpfa_get_series_harmonics(svgpfa_engine* e, int32_t* out_host) {
  if (!e || !out_host) return SVGPFA_ERR_INVALID;
  HIPCHK(e, hipSetDevice(e->device));
  if (!e->have_kernels) FAIL(e, SVGPFA_ERR_STATE, "svgpfa_build_kernels has not been called");
  HIPCHK(e, hipMemcpyAsync(out_host, e->v.fourH, sizeof(int32_t) * e->v.K, hipMemcpyDeviceToHost, e->stream));
  HIPCHK(e, hipStreamSynchronize(e->stream));
  for (int k = 0; k < e->v.K; ++k) {
    if (e->meta[k].ktype == SV_RBF && e->have_spikes && e->v.rbf_series) out_host[k] = -2;
    if (e->have_spikes && !e->cls_on) out_host[k] = -1;      // small problem: every latent pair by pair
  }
  return SVGPFA_OK;
}

int svgpfa_set_embedding_token(svgpfa_engine* e, uint64_t token) {
  if (!e) return SVGPFA_ERR_INVALID;
  e->emb_token = token;
  return SVGPFA_OK;
}

int svgpfa_enable_timing(svgpfa_engine* e, int on) {
  if (!e) return SVGPFA_ERR_INVALID;
  e->timing = on != 0;
  e->timing_period = on > 1 ? on : 1;
  e->timing_eval = 0;
  e->ev_used = 0;
  e->ev_tot[0] = e->ev_tot[1] = 0.0;
  e->ev_cnt[0] = e->ev_cnt[1] = 0;
  return SVGPFA_OK;
}

int svgpfa_kernel_ms(svgpfa_engine* e, int32_t which, double* total_ms, int32_t* launches) {
  if (!e || !total_ms || !launches || which < 0 || which > 1) return SVGPFA_ERR_INVALID;
  int rc = fold_events(e);
  if (rc) return rc;
  *total_ms = e->ev_tot[which];
  *launches = (int32_t)e->ev_cnt[which];
  return SVGPFA_OK;
}

int svgpfa_spike_kernel_ms(svgpfa_engine* e, double* total_ms, int32_t* launches) {
  return svgpfa_kernel_ms(e, 0, total_ms, launches);
}

}  // extern "C"
