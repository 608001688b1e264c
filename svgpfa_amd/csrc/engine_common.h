// Shared host/device definitions of the svGPFA ELBO engine (gfx950 only).
//
// HBM layout (all fp64, engine-owned unless noted).  R trials, K latents, M_k inducing
// points, Q quadrature nodes, S spikes on this GPU.
//   per-latent blocks   X[k]  : base offM[k]  = R * sum_{k'<k} M_k'      element (r, j)    at offM  + r*M   + j
//                       XX[k] : base offMM[k] = R * sum_{k'<k} M_k'^2    element (r, i, j) at offMM + r*M*M + i*M + j
//   Zs (snapshot of Z), a = Kzz^-1 m, ut = L^-1 m, adjoint of ut          : X layout
//                       TT[k] : base offTT[k] = R * sum_{k'<k} PT_k': PACKED lower triangle, rows padded to an even length
//                               (PT = M (M + 1) / 2 + ceil(M / 2)), element (r, i, j <= i) at offTT + r*PT + sv_tri_off(i) + j
//   Kzz, L = chol(Kzz) (written only when a getter asks) : XX layout
//   Linv = L^-1, Vt = L^-1 L_S (both lower triangular)   : TT layout -- half the bytes of a full matrix both ways, and a
//                               staging loop's consecutive elements are consecutive addresses across row ends
// Everything downstream of the Cholesky works in WHITENED coordinates (kt = L^-1 kappa(t, Z),
// ut = L^-1 m, Vt = L^-1 L_S): mu = kt.ut, var = s2 - |kt|^2 + |Vt^T kt|^2.  All whitened
// quantities are O(1), which keeps fp64 error at ~sqrt(cond(Kzz)) u instead of cond(Kzz)^2 u
// for the explicit-inverse algebra (measured against a 40-digit referee, tests/tools/).
//   quad stats mu/var/gmu/gvar : [r][k][q]  (a tile of 16 nodes of one latent = 128 contiguous bytes)
//   spikes: times[S], neuron[S] (CSR over trials, neuron-major inside a trial), mu_spk[k][S]
//   spike chunks: <= SPIKE_CHUNK spikes of one trial; partial sums P0/P1 [chunk][sumM], T [chunk][K][2]
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sv_math.h"

typedef double d4 __attribute__((ext_vector_type(4)));      // accumulator tile of v_mfma_f64_16x16x4_f64

#define SV_RBF 0
#define SV_PERIODIC 1
#define SV_MAX_M 64
#define SV_MAX_K 32

#define SV_FH 64                      // most harmonics of the periodic kernel's Fourier series used by the spike pass
#define SV_FROWS 3                    // rows of the per-latent series table: c_n | g_n | e_n (kernels_spike.h)
#define SV_SPL 8                      // spikes per lane in the pair-by-pair spike pass
#define SV_SPIKE_CHUNK (64 * SV_SPL)  // largest number of spikes per wavefront work item of that pass
#define SV_CLS_MAXMEM 16              // latents per class of the class path of the spike pass (one MFMA operand tile)
#define SV_ACOLS (4 * (SV_FH + 1))    // columns of one A-moment row: [harmonic][cos | sin | t cos | t sin]
#define SV_WCOLS (2 * (SV_FH + 1))    // columns of one W row: [harmonic][cos | sin]

// Rows of a tile's transposed times that cls_pass (kernels_spike_cls.h) reads (it fetches a group of SV_CLS_PF spikes' times ahead of the loop
// and a further group in every iteration, unconditionally): rows 0 .. sv_cls_rows_read(maxlen) - 1.  The host pads the
// array behind the last tile by SV_CLS_TT_PAD doubles so that the read-ahead of the LAST tile stays inside the allocation
// (inside the array it lands in the next tile).
#define SV_CLS_PF 4
#define SV_CLS_TT_PAD (2 * SV_CLS_PF * 64)
__host__ __device__ constexpr int sv_cls_rows_read(int maxlen) {
  return (maxlen <= 0) ? SV_CLS_PF : ((maxlen + SV_CLS_PF - 1) / SV_CLS_PF) * SV_CLS_PF + SV_CLS_PF;
}
// every row length a tile can have (rows are cut at SV_SPIKE_CHUNK spikes): the read-ahead of a tile ends at most SV_CLS_TT_PAD
// doubles behind the tile itself, hence -- tiles are contiguous -- inside the next tile or, for the last one, inside the pad
__host__ __device__ constexpr bool sv_cls_tt_pad_covers_every_length() {
  for (int m = 0; m <= 64 * 8; ++m)
    if (64 * sv_cls_rows_read(m) > 64 * m + SV_CLS_TT_PAD) return false;
  return true;
}
static_assert(sv_cls_tt_pad_covers_every_length(), "tt padding shorter than the class kernel's read-ahead");

struct LatentMeta {
  int M, P, ktype, nth;
  long long offM, offMM;
  long long offTT;   // packed lower-triangular per-item matrices (Linv, Vt): R * sum_{k'<k} PT_k', element (r, i, j <= i) at offTT + r PT + sv_tri_off(i) + j
  int PT;            // doubles per packed item: rows padded to an even length (sv_tri_off(M))
  int colM;    // sum_{k'<k} M_k'
  int thoff;   // sum_{k'<k} nth_k'
  double scale2, inv_ls_scale, inv_p_scale;
};

// Row offset in a packed lower triangle whose rows are padded to an EVEN number of doubles (the pad element is a zero the
// producer writes): every (i, 2p), (i, 2p + 1) pair is 16 aligned bytes, so a row can be copied in 16-byte pieces --
// what the LDS-DMA staging of the large factors needs (kernels_mfma.h stage_matrix_dma).
__host__ __device__ constexpr int sv_tri_off(int i) { return i * (i + 1) / 2 + ((i + 1) >> 1); }

struct LatentPtrs {
  const double* mean;
  const double* cholvec;
  double* g_mean;
  double* g_cholvec;
  double* g_Z;
};

// A chunk holds <= 64 * spl consecutive spikes of one trial; chunks are packed densely (every chunk of a trial but the
// last is full), so a (trial, neuron) segment may be cut by a chunk boundary -- at most once.  The chunk that holds the
// segment's start writes its partial sum to gCspk, the chunk that holds the continuation (split bit 1: "head
// continuation") writes its first per-neuron sum to gCspkB: every dC partial keeps exactly one writer and the two
// halves are added in the fixed-order reduction.  A neuron with more spikes in a trial than a chunk holds gets dedicated
// chunks (split bit 0, all spikes of that one neuron); their sums are added in chunk order later.
struct LatentPtrTable {
  LatentPtrs p[SV_MAX_K];
};

struct SpikeChunk {
  int trial;
  int count;
  long long start;  // global spike index of the first spike of the chunk
  int split;        // bit 0: piece of an oversize (trial, neuron) segment; bit 1: the first segment continues the previous
                    // chunk's last; bit 2: the last segment continues in the next chunk
  int neuron;       // that neuron (split chunks only)
};

// ---- class path of the spike pass (kernels_spike_cls.h).  Static per engine: a ROW is one (trial, neuron) segment, one
// half of a segment (cut to fill the trial's tiles), or a piece of an oversize segment (> SV_SPIKE_CHUNK spikes, the same
// threshold as the pair-by-pair pass); a TILE is <= 64 rows of one trial, sorted by length; the spike times of a tile are
// stored transposed (tt[tt_off + i * 64 + row] = i-th spike of the row) so that the per-lane loops load coalesced.
struct SpikeRow {
  int len;
  int neuron;
  int kind;         // 0 whole segment (dC -> plane A, plane B cleared), 1 first half (-> A), 2 second half (-> B), 3 oversize piece (-> slot)
  int slot;         // kind 3: index into the split list
};
struct SpikeTile {
  int trial, nrows, maxlen, row0;
  long long tt_off;
  int id, pad;      // index of the tile in trial order (the order of the partial-moment rows)
};
// What the class path needs to know about one (trial, latent) item: harmonics (-1: the pair-by-pair pass serves it), class,
// phase origin and quarter turns per unit time of the (trial, class).
struct SpikeItem {
  int H, cls;
  double t0, sc;
};
// Classes of latents that share the angular frequency of their cosine series; rebuilt with every kernel build.
struct SpikeClassTab {
  int ncls;
  int type[SV_MAX_K];                    // SV_RBF / SV_PERIODIC
  int nmem[SV_MAX_K];
  int mem[SV_MAX_K][SV_CLS_MAXMEM];      // member latents
  int cls_of[SV_MAX_K];                  // class of a latent, or -1: pair by pair in every trial
  int slot_of[SV_MAX_K];                 // its index among the members
};

// svgpfa_build_kernels' inputs: the caller's Z / theta tensors, snapshotted into engine-owned HBM by k_build_reg
struct SnapTable {
  const double* Z[SV_MAX_K];
  const double* theta[SV_MAX_K];
};

// Everything a kernel needs, passed by value (fits well inside the kernarg segment).
struct EngineView {
  int R, N, K, Q, S;
  int sumM, nchunks, nth_total;
  int ncd, nell;              // rows of dC / dd partials ([ncd][N][2K], [ncd][N]) and number of ELL partials
  int ntype[2];               // number of RBF / periodic latents
  int klist[2][SV_MAX_K];     // their indices: kernels are launched once per covariance type
  double reg;
  const LatentMeta* meta;
  const double* tq;
  const double* wq;
  const double* spk_t;
  const int* spk_n;
  const double* spk_lo;       // R : earliest / latest spike of each trial (window of the RBF kernel's cosine series)
  const double* spk_hi;
  int rbf_series;             // 1: the spike pass may evaluate RBF kernels through their periodised series (default)
  int per_series;             // 1: ... periodic kernels through their Jacobi-Anger series (default)
  const long long* spk_off;   // R+1
  const long long* seg_off;   // R*N+1 : first spike of (trial, neuron)
  const double* spk_cnt;      // N     : spikes per neuron over all local trials
  const SpikeChunk* chunks;
  const int* chunk_first;     // R+1   : first chunk of each trial
  const int* split_chunks;    // indices of the split chunks (usually none)
  int nsplit;
  double* Zs;
  double* theta;              // K x 2 snapshot
  double* four;               // [K][SV_FROWS][SV_FH + 2] series of the periodic kernels (periodic_series), made by k_build_reg
  int* fourH;                 // [K] harmonics used (n = 0 .. H), or -1: evaluate the kernel pair by pair
  double* Kzz; double* L; double* Linv;
  double* a; double* ut; double* Vt; double* klpart;   // klpart [R*K]
  double* muq; double* varq; double* gmuq; double* gvarq;
  long long glat2;            // != 0: gmuq / gvarq of this evaluation are the sum of two planes this many doubles apart (k_quad_ell_fused)
  double* P0; double* P1; double* Tpart;           // spike partials
  double* gCspk;                                   // [R][N][K] spike part of dC per trial (segment, or its first part)
  double* gCspkB;                                  // [R][N][K] second part of segments cut by a chunk boundary / cut to fill a tile: VALID
                                                   // only where seg_cut says so (sv_dc_spike), whatever it holds elsewhere
  const unsigned char* seg_cut;                    // [R][N] static: bit 0 the pair pass's chunks cut the (trial, neuron) segment,
                                                   // bit 1 the class path's rows do (kinds 1 / 2)
  double* splitPart;                               // [nchunks][K] spike part of dC of split chunks
  // class path (kernels_spike_cls.h)
  SpikeClassTab* cls;                              // device copy of the class table
  double* cls_t0; double* cls_sc; int* cls_H;      // [class][R]: phase origin, quarter turns per unit time, harmonics (-1: pair loop)
  int* ser;                                        // [R][K]: 1 = the class path serves this (trial, latent), 0 = the pair loop
  SpikeItem* sitem;                                // [R][K]: the same with the (trial, class) parameters
  int* pair_any;                                   // [R]: 1 when some latent of the trial (with spikes) is left to the pair loop
  int nrows, ntiles, nsplitS;
  const SpikeRow* rows;
  const SpikeTile* tiles;                          // contiguous per trial
  const int* tile_first;                           // R+1
  const SpikeTile* tiles_ord;                      // the same tiles, longest first (work order of k_spike_cls)
  const double* tt;                                // transposed spike times of the tiles
  const int* splitS_neuron; const int* splitS_trial;   // oversize pieces of the class path
  double* splitS;                                  // [nsplitS][K]
  double* Apart;                                   // [ntiles][K][SV_ACOLS] A-moment partials
  double* Wb;                                      // [R][K][SV_WCOLS] posterior-mean weights on the harmonics
  double* P0t; double* P1t; double* Tt;            // [R][sumM] x 2, [R][K][2]: spike sums of class-path items (k_spike_comb)
  int comb_fold;                                   // 1: k_item_bwd's gather combines the moments itself (small shards)
  double* muspk;                                   // [K][S] (stats API only)
  double* varspk;                                  // [K][S] (stats API only)
  double* gmuspk; double* gvarspk;                 // [K][S] adjoints at the spikes (svgpfa_latent_stats_backward only)
  double* gCpart; double* gdpart; double* ellq;    // [ncd][N][2K] raw sums (E^T mu | E^T var), [ncd][N], [nell]
  double* ells; double* gthitem;                   // [R*K], [R*K*2]
  double* Mseg;                                    // [N][K] sum over spikes of mu_spk per neuron (stats path)
  // Whitened kernel tiles at the quadrature nodes, Kw = Linv kappa(t_q, Z), kept across evaluations while the kernels
  // stand (the E-step closure: reference stats/svEM.py:218-223 runs on the Ktz its KernelsMatricesStore cached).  One
  // register image of k_front's accumulator per (item, node tile, panel): [R K][ceil(Q / 16)][NT][4][64] doubles
  // (Kw[a][q] of lane (li, lg) = Ktil[inducing 16 a + 4 q + lg][node li]), and behind it (+ kw_half doubles) the same tiles as
  // the register images k_item_bwd wants (KwT[a][q] of lane (li, lg) = Ktil[inducing 16 a + li][node lg + 4 q]): both kernels
  // then load 512 contiguous bytes per wave instruction.  kw_mode: 0 off, 1 k_front stores both, 2 k_front loads its image
  // instead of evaluating the kernel and whitening; kw_use: k_item_bwd<*, false> loads its image.
  double* Kwc;
  long long kw_half;
  int kw_mode, kw_use, kw_nt;
  int* status;
  unsigned* fin_ticket;       // k_fin_items: scalar blocks that have finished (zero at rest)
  double* fin_part;           // [0] ELL terms known before k_item_bwd, [1] KL, [2 + b] spike-ELL partial of scalar block b
};

// ------------------------------------------------------------------ covariance functions
// kappa(x1, x2) = s2 * ktil(delta), delta = x1 - x2.
//   RBF      (reference stats/kernels.py:33-46): ktil = exp(-delta^2 * h),        h = 1/(2 l^2)
//   periodic (reference stats/kernels.py:73-85): ktil = exp(-sin^2(pi delta/p) h), h = 2/l^2
// Derivatives are expressed through basis values (bz, b0, b1) and constants (cz, c0, c1):
//   d kappa / d x2      = ktil * bz * cz
//   d kappa / d theta_0 = ktil * b0 * c0        (lengthscale, unscaled parameter)
//   d kappa / d theta_1 = ktil * b1 * c1        (period, unscaled parameter; periodic only)
// One MFMA operand fragment from LDS as a ds_read_b64 of its own.  The compiler's load/store optimiser otherwise pairs
// fragment reads that share a base register into ds_read2_b64 -- half the rate of two ds_read_b64 and banked modulo 32
// dwords over 16-lane groups (MI355X_MICROARCH.md, LDS), where the row padding LD = MP + 2 that makes ds_read_b64 (modulo
// 64, 32-lane groups) conflict-free is a two-way conflict: every A-operand read of the MFMA chains cost four times the LDS
// cycles it needs (round 4, read off the device assembly with tests/tools/isa_lds_mix.py).  A volatile access is never paired.
#ifndef SV_LDS_VOLATILE
#define SV_LDS_VOLATILE 1
#endif
__device__ __forceinline__ double lds_frag(const double* p) {
#if SV_LDS_VOLATILE
  // (an explicit LDS pointer: address-space inference leaves volatile accesses alone, and a generic pointer made this a flat load)
  return *(const volatile __attribute__((address_space(3))) double*)p;
#else
  return *p;
#endif
}

__device__ __forceinline__ float lds_frag(const float* p) {
#if SV_LDS_VOLATILE
  return *(const volatile __attribute__((address_space(3))) float*)p;
#else
  return *p;
#endif
}

// LDS hand-off between lanes of ONE wavefront (no s_barrier needed)
__device__ __forceinline__ void lds_wave_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

struct KPar {
  double s2, h, ip, cz, c0, c1;
  double h2;   // base-2 exponent scale: RBF h log2(e); periodic h log2(e) / 2
  double q4;   // periodic: 4 / p  (angle 2 pi delta / p = (pi/2) (q4 delta))
};

__device__ __forceinline__ KPar make_kpar(const LatentMeta& m, const double* th) {
  KPar p;
  const double PI = 3.14159265358979323846;
  const double L2E = 1.4426950408889634074;
  double l = th[0] * m.inv_ls_scale;
  p.s2 = m.scale2;
  if (m.ktype == SV_RBF) {
    p.h = 0.5 / (l * l);
    p.h2 = p.h * L2E;
    p.q4 = 0.0;
    p.ip = 0.0;
    p.cz = p.s2 * 2.0 * p.h;
    p.c0 = p.s2 * 2.0 * p.h / l * m.inv_ls_scale;
    p.c1 = 0.0;
  } else {
    double per = th[1] * m.inv_p_scale;
    p.ip = 1.0 / per;
    p.h = 2.0 / (l * l);
    p.h2 = 0.5 * p.h * L2E;
    p.q4 = 4.0 * p.ip;
    p.cz = p.s2 * PI * p.h * p.ip;
    p.c0 = p.s2 * 2.0 * p.h / l * m.inv_ls_scale;
    p.c1 = p.s2 * PI * p.h * p.ip * p.ip * m.inv_p_scale;
  }
  return p;
}

// ktil via the lean base-2 kernels of sv_math.h (no special cases, <= 1 ulp polynomials):
//   RBF       ktil = 2^(-delta^2 h2)
//   periodic  sin^2(pi delta / p) = (1 - cos(2 pi delta / p)) / 2  ->  ktil = 2^(h2 (cs - 1))
template <int KT>
__device__ __forceinline__ double kern_val(double delta, const KPar& p) {
  if (KT == SV_RBF) {
    return sv_exp2_neg(-(delta * delta) * p.h2);
  } else {
    double sn, cs;
    sv_sincos_quarter(delta * p.q4, sn, cs);
    return sv_exp2_neg(fma(cs, p.h2, -p.h2));
  }
}

template <int KT>
__device__ __forceinline__ void kern_eval(double delta, const KPar& p, double& kt, double& bz,
                                          double& b0, double& b1) {
  if (KT == SV_RBF) {
    double q = delta * delta;
    kt = sv_exp2_neg(-q * p.h2);
    bz = delta;
    b0 = q;
    b1 = 0.0;
  } else {
    double sn, cs;
    sv_sincos_quarter(delta * p.q4, sn, cs);
    kt = sv_exp2_neg(fma(cs, p.h2, -p.h2));
    bz = sn;
    b0 = 0.5 - 0.5 * cs;
    b1 = delta * sn;
  }
}

// Periodic kernel as a cosine series (Jacobi-Anger).  With beta = 1 / l^2 and x = 2 pi delta / p
//     ktil(delta) = exp(-2 sin^2(pi delta / p) / l^2) = exp(beta (cos x - 1)) = sum_{n >= 0} c_n cos(n x),
//     c_0 = e^-beta I_0(beta),   c_n = 2 e^-beta I_n(beta)
// the coefficients fall super-exponentially (beta = 0.44: c_13 ~ 1e-19), so sums over spikes of ktil and of its
// derivatives follow from 4 (H + 1) trigonometric moments of the spikes instead of one exponential per (spike,
// inducing point) pair (kernels_spike.h).  Scaled Bessel functions e^-beta I_n(beta) by Miller's downward recurrence,
// normalised with e^-beta (I_0 + 2 sum I_n) = 1; rows: c_n, g_n = -1/2 dc_n / dbeta with dc_n / dbeta = e^-beta ((I_{n-1} +
// I_{n+1}) / 2 - I_n) (x 2 for n > 0), e_n = n c_n / beta.  Returns the last harmonic H with n c_n >= 1e-18 (every kernel value is <= 1), or -1 when more
// than SV_FH harmonics would be needed (short lengthscales) or the pair-by-pair evaluation is cheaper (few inducing
// points).  One lane per latent, once per kernel build.
// f: workspace of SV_FH + 3 doubles (LDS of the caller: a private array would put the whole kernel on scratch memory).
__device__ inline int periodic_series(double beta, int M, double* tab, double* f) {
  if (!(beta > 0.0) || beta > 45.0) return -1;
  // Start of the downward recurrence, far inside the decaying tail: I_n(beta) <= (beta / 2)^n / n!, so from n = 32 + 2 beta
  // down the truncation error is below 1e-30 of I_0 for every beta <= 45 (SV_FH + 48 = 96 at the upper end; a fixed start
  // at 96 made the class riders of k_build_reg, which repeat this per build for their harmonic counts, 8 us longer than
  // the kernel's item blocks).  Entries above the start are zeros.
  const int NS = min(SV_FH + 48, 32 + (int)(2.0 * beta));
  for (int n = NS + 1; n <= SV_FH + 2; ++n) f[n] = 0.0;
  double fp = 0.0, fc = 1e-280, sum = 0.0;       // f_{n+1}, f_n
  const double tb = 2.0 / beta;
  for (int n = NS; n >= 1; --n) {
    if (n <= SV_FH + 2) f[n] = fc;
    sum += 2.0 * fc;
    const double fm = fma(tb * n, fc, fp);       // f_{n-1} = f_{n+1} + (2 n / beta) f_n
    fp = fc;
    fc = fm;
    if (fc > 1e200) {                            // small beta: the recurrence grows by 2 n / beta per step
      const double sc = 1e-200;
      fc *= sc; fp *= sc; sum *= sc;
      for (int q = n; q <= SV_FH + 2 && q <= NS; ++q) f[q] *= sc;
    }
  }
  f[0] = fc;
  sum += fc;
  const double inv = 1.0 / sum;
  int H = 0;
  if (!tab) {                                    // H only (class riders / k_spike_classes): no table, no divisions
    for (int n = 1; n <= SV_FH + 1 && n <= NS; ++n)
      if (n * (2.0 * (f[n] * inv)) >= 1e-18) H = n;
    if (H > SV_FH) return -1;
    if (12 * (H + 2) > 23 * M) return -1;
    return H;
  }
  for (int n = 0; n <= SV_FH + 1; ++n) {
    const double ive = f[n] * inv, ivep = f[n + 1] * inv, ivem = (n ? f[n - 1] : f[1]) * inv;
    const double cn = (n ? 2.0 : 1.0) * ive;
    tab[n] = cn;
    tab[(SV_FH + 2) + n] = -0.5 * (n ? 2.0 : 1.0) * (0.5 * (ivem + ivep) - ive);   // g_n = -1/2 dc_n / dbeta
    tab[2 * (SV_FH + 2) + n] = n * cn / beta;
    if (n * cn >= 1e-18 || n == 0) H = n;
  }
  if (H > SV_FH) return -1;
  if (12 * (H + 2) > 23 * M) return -1;          // per-spike cost of the series against M kernel evaluations
  return H;
}

// Lane exchanges inside a row of 16 lanes by DPP (v_mov_b32_dpp on the two halves of a double: register-to-register, a few
// cycles) instead of ds_bpermute_b32 (through the LDS pipeline, ~100 cycles per step and a waitcnt): the cross-lane sums of
// this engine sit in the serial sections of its per-item kernels, where nothing overlaps with them.
// (mov_dpp, not update_dpp(x, x, ...): with an `old` operand the compiler first copies x into the destination -- two more
// v_mov_b32 per exchanged double, five instructions per reduction step instead of three, in kernels that are issue-bound;
// a rotation inside a row has no out-of-range source, and every caller runs with all lanes of the wave active)
template <int CTRL>
__device__ __forceinline__ double dpp_mov_f64(double x) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xf, 0xf, true);
  hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
// sum over the 16 lanes of a row (lanes sharing lane >> 4), every lane gets it: rotations by 8, 4, 2, 1 (row_ror)
__device__ __forceinline__ double row16_sum(double x) {
  x += dpp_mov_f64<0x128>(x);
  x += dpp_mov_f64<0x124>(x);
  x += dpp_mov_f64<0x122>(x);
  x += dpp_mov_f64<0x121>(x);
  return x;
}
__device__ __forceinline__ double wave_sum(double x) {     // every lane gets the sum over the wavefront (fixed order)
  x = row16_sum(x);
  x += __shfl_xor(x, 16, 64);
  x += __shfl_xor(x, 32, 64);
  return x;
}

// Deterministic block sum (fixed tree); `red` needs blockDim.x/64 doubles.  All threads get the result.
__device__ __forceinline__ double block_sum(double x, double* red) {
  x = wave_sum(x);
  int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = x;
  __syncthreads();
  double s = 0.0;
  for (int i = 0; i < nw; ++i) s += red[i];
  return s;
}

// Spike part of dC of (trial r, neuron n, latent k): plane A, plus plane B where the pass that served the item (ev.ser) cut the
// segment in two.  Round 5: the passes used to clear plane B for every uncut segment so that it stayed valid when an item
// moved between them (8 MB of zeros stored per evaluation at the north-star size, and 8 MB read back here); which
// segments each pass cuts is static, so a one-byte flag per (trial, neuron) says instead where plane B counts.
__device__ __forceinline__ double sv_dc_spike(const EngineView& ev, int r, int n, int k) {
  const long long o = ((long long)r * ev.N + n) * ev.K + k;
  const double a = ev.gCspk[o];
  const int bit = ev.ser[r * ev.K + k] ? 2 : 1;
  return (ev.seg_cut[(long long)r * ev.N + n] & bit) ? a + ev.gCspkB[o] : a;
}

// dC partials of oversize (trial, neuron) segments (usually none): the pair-by-pair pass leaves them per split chunk, the
// class path per split row; which of the two lists is current for a (trial, latent) follows ev.ser.  Fixed order.
__device__ __forceinline__ double split_sums(const EngineView& ev, int n, int k) {
  double s = 0.0;
  for (int l = 0; l < ev.nsplit; ++l) {
    const int ch = ev.split_chunks[l];
    if (ev.chunks[ch].neuron == n && !ev.ser[ev.chunks[ch].trial * ev.K + k]) s += ev.splitPart[(long long)ch * ev.K + k];
  }
  for (int l = 0; l < ev.nsplitS; ++l)
    if (ev.splitS_neuron[l] == n && ev.ser[ev.splitS_trial[l] * ev.K + k]) s += ev.splitS[(long long)l * ev.K + k];
  return s;
}
