/*
 * svgpfa.h -- C ABI of the MI355X-native svGPFA ELBO engine (libsvgpfa_hip.so).
 *
 * The reference (joacorapela/svGPFA) is pure Python/PyTorch and has no FFI; the seam this
 * library replaces is the object protocol between its EM loop and its SVLowerBound object
 * graph.  Every entry point below names the reference method whose work it
 * does (paths relative to /root/reference/src/svGPFA/).  Host code stays Python and binds
 * these symbols with ctypes (svgpfa_amd/_lib.py); see INTEGRATION.md for the stub a
 * reference maintainer would add.
 *
 * Conventions
 *   - All data pointers are DEVICE pointers (hipMalloc'd / torch ROCm tensor data_ptr()),
 *     fp64, C-contiguous, unless the parameter name ends in _host.
 *   - Per-latent arguments are HOST arrays of n_latents device pointers.
 *   - Every call enqueues on the engine's stream (svgpfa_set_stream) and returns without
 *     waiting; svgpfa_sync() waits and reports device-side failures (non-PD Kzz).
 *   - Return value: SVGPFA_OK or an SVGPFA_ERR_* code; text via svgpfa_last_error().
 *   - One engine = one GPU = one contiguous block of trials.  Nothing here is collective:
 *     the multi-GPU sum is ONE all-reduce of the `shared` vector done by the host
 *     (torch.distributed / RCCL), see svgpfa_shared_len().
 *   - Not re-entrant per engine; distinct engines are independent.
 */
#ifndef SVGPFA_H
#define SVGPFA_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct svgpfa_engine svgpfa_engine;

enum {
  SVGPFA_OK = 0,
  SVGPFA_ERR_INVALID = 1,     /* bad argument / shape / ordering                       */
  SVGPFA_ERR_NOT_PD = 2,      /* Kzz + eps*I (or S) not positive definite: the reference
                                 raises torch.linalg.LinAlgError at utils/miscUtils.py:215 */
  SVGPFA_ERR_HIP = 3,         /* a HIP runtime call failed                             */
  SVGPFA_ERR_STATE = 4,       /* call order violated (e.g. elbo before build_kernels)  */
  SVGPFA_ERR_UNSUPPORTED = 5  /* size beyond what this build handles                   */
};

/* stats/kernels.py: ExponentialQuadraticKernel (:26-64), PeriodicKernel (:66-108) */
enum { SVGPFA_KERNEL_RBF = 0, SVGPFA_KERNEL_PERIODIC = 1 };

/* which parameter groups receive gradients -- the four closures of stats/svEM.py:218-264 */
enum {
  SVGPFA_GRAD_VARIATIONAL = 1, /* q(u) means + cholVecs   (_eStep)              */
  SVGPFA_GRAD_EMBEDDING = 2,   /* C, d                    (_mStepEmbedding)     */
  SVGPFA_GRAD_KERNELS = 4,     /* kernel hyper-parameters (_mStepKernels)       */
  SVGPFA_GRAD_INDPOINTS = 8    /* inducing-point locations(_mStepIndPointsLocs) */
};

/* arithmetic of the ELBO terms (svgpfa_dims.precision).  The reference is fp64 throughout (stats/kernels.py:29,
   utils/miscUtils.py:136,151,227); MIXED is BASELINE.json configs[4] ("fp32 ELBO with fp64 Cholesky"): the spike-time
   pass and the quadrature expected-log-likelihood run in fp32, Kzz, its Cholesky factor, the whitened algebra, the KL
   term and every cross-chunk / cross-trial sum stay fp64.  Tolerances: DESIGN.md section 5. */
enum { SVGPFA_PRECISION_FP64 = 0, SVGPFA_PRECISION_MIXED = 1 };

#define SVGPFA_MAX_IND 64      /* inducing points per latent handled by one wavefront   */
#define SVGPFA_MAX_LATENTS 32

typedef struct {
  int32_t n_trials;            /* R (this engine's shard)                                */
  int32_t n_neurons;           /* N                                                      */
  int32_t n_latents;           /* K                                                      */
  int32_t n_quad;              /* Q Gauss-Legendre nodes per trial                       */
  const int32_t* n_ind;        /* [K] M_k, may differ per latent (utils/initUtils.py:402-415) */
  const int32_t* kernel_type;  /* [K] SVGPFA_KERNEL_*                                    */
  const double* scale;             /* [K] Kernel._scale            (NULL = 1.0)          */
  const double* lengthscale_scale; /* [K] Kernel._lengthscaleScale (NULL = 1.0)          */
  const double* period_scale;      /* [K] Kernel._periodScale      (NULL = 1.0)          */
  int32_t precision;           /* SVGPFA_PRECISION_* (0 = fp64, the reference's arithmetic)      */
} svgpfa_dims;

/* SVGPFAModelFactory.buildModelPyTorch (stats/svGPFAModelFactory.py:40-148), default options. */
int svgpfa_create(const svgpfa_dims* dims, int device, svgpfa_engine** out);
void svgpfa_destroy(svgpfa_engine* e);
const char* svgpfa_last_error(const svgpfa_engine* e);

/* Work is enqueued on this hipStream_t (pass torch.cuda.current_stream().cuda_stream). */
int svgpfa_set_stream(svgpfa_engine* e, void* hip_stream);
/* Wait for the stream; returns SVGPFA_ERR_NOT_PD if any enqueued factorisation failed
 * since the last sync (then svgpfa_last_error names trial/latent). */
int svgpfa_sync(svgpfa_engine* e);

/* IndPointsLocsKMS.setRegParam (stats/kernelsMatricesStore.py:92-93). */
int svgpfa_set_reg(svgpfa_engine* e, double eps);

/* PointProcessELL.setELLCalculationParams (stats/expectedLogLikelihood.py:187-189):
 * leg_quad_points / leg_quad_weights, R x Q each.  Copied into engine-owned HBM. */
int svgpfa_set_quad(svgpfa_engine* e, const double* t_quad, const double* w_quad);

/* PointProcessELL.setMeasurements + __stackSpikeTimes (stats/expectedLogLikelihood.py:149-173).
 * CSR over trials; inside a trial spikes must be grouped by non-decreasing neuron index (the
 * reference's stacking order).  offsets_host: R+1 int64 on the HOST; spike_times: S doubles,
 * spike_neuron: S int32 on the device.  Copied into engine-owned HBM. */
int svgpfa_set_spikes(svgpfa_engine* e, const int64_t* offsets_host,
                      const double* spike_times, const int32_t* spike_neuron);

/* SVLowerBound.buildKernelsMatrices (stats/svLowerBound.py:77-78 -> kernelsMatricesStore.py:107-117,
 * utils/miscUtils.py:209-216): snapshot Z_k (R x M_k) and theta_k (1 or 2 doubles), build
 * Kzz + eps*I, its Cholesky factor, and the inverse factor.  Ktz blocks are never materialised:
 * later evaluations recompute kappa(t, Z) on the fly from this snapshot, exactly as the
 * reference keeps using the matrices of its last buildKernelsMatrices() call. */
int svgpfa_build_kernels(svgpfa_engine* e, const double* const* Z, const double* const* theta);

/* Length (in doubles) of the `shared` output vector:
 *   [0] ELBO  [1] ELL  [2] KL
 *   [3 .. 3+N*K)            dELBO/dC   (row-major N x K)
 *   [.. +N)                 dELBO/dd
 *   [.. +sum_k ntheta_k)    dELBO/dtheta_k, latents in order
 * These are exactly the quantities that must be summed over GPUs when trials are sharded. */
int64_t svgpfa_shared_len(const svgpfa_engine* e);

/* SVLowerBound.eval (stats/svLowerBound.py:47-54) and its autograd backward
 * (stats/svEM.py:274-288) in one call.  Gradients are of +ELBO.  Groups not in grad_mask are
 * neither computed nor written; their pointers may be NULL.  g_mean[k]: R x M_k,
 * g_cholvec[k]: R x P_k (row-major tril order), g_Z[k]: R x M_k. */
int svgpfa_elbo(svgpfa_engine* e,
                const double* const* mean, const double* const* cholvec,
                const double* C, const double* d,
                unsigned grad_mask, double* shared,
                double* const* g_mean, double* const* g_cholvec, double* const* g_Z);

/* SVLowerBound.computeSVPosteriorOnLatentsStats (stats/svLowerBound.py:80-81):
 * posterior mean/variance of every latent at the quadrature nodes and at the spikes.
 * The results stay inside the engine (for svgpfa_ell_given_stats) and are also copied to
 * the optional outputs: mu_quad/var_quad R x Q x K, mu_spk/var_spk S x K. */
int svgpfa_latent_stats(svgpfa_engine* e,
                        const double* const* mean, const double* const* cholvec,
                        double* mu_quad, double* var_quad, double* mu_spk, double* var_spk);

/* Backward of svgpfa_latent_stats plus the KL term: given the adjoints of a caller-defined expected log-likelihood wrt
 * the statistics (g_mu_quad / g_var_quad R x Q x K, g_mu_spk / g_var_spk S x K: d ELL / d mean and variance of every latent
 * at every quadrature node and spike), the gradients of  ELL - KL  wrt the groups in grad_mask (VARIATIONAL, KERNELS,
 * INDPOINTS; the embedding is the caller's).  This is what autograd does for the reference when the likelihood is not
 * the exponential-link point process, e.g. the Gauss-Hermite link of stats/expectedLogLikelihood.py:215-255, whose
 * spike term has no a-priori adjoint: the GP algebra (whitened chain, Kzz path, KL) stays in HIP, the link in the
 * caller.  shared (svgpfa_shared_len doubles): [2] = KL and the trailing dtheta entries are written; the per-trial
 * gradients go to g_mean / g_cholvec / g_Z as in svgpfa_elbo. */
int svgpfa_latent_stats_backward(svgpfa_engine* e,
                                 const double* const* mean, const double* const* cholvec,
                                 const double* g_mu_quad, const double* g_var_quad,
                                 const double* g_mu_spk, const double* g_var_spk,
                                 unsigned grad_mask, double* shared,
                                 double* const* g_mean, double* const* g_cholvec, double* const* g_Z);

/* Diagnostics: how the spike-time pass evaluates each latent's kernel since the last svgpfa_build_kernels.
 * out_host[k] (HOST array of n_latents) = H >= 0: periodic kernel through the first H + 1 terms of its cosine series
 * exp(beta (cos x - 1)) = sum_n c_n cos(n x) (terms below 1e-18 dropped); -2: RBF kernel through the cosine series of
 * its periodisation over the item's time window, period and number of terms chosen per (trial, latent) at run time
 * (pair by pair where the lengthscale is short against the window); -1: one kernel evaluation per (spike, inducing
 * point) pair (periodic latents with a short lengthscale or few inducing points; in fp32 in precision mode "mixed"). */
int svgpfa_get_series_harmonics(svgpfa_engine* e, int32_t* out_host);

/* What svgpfa_ell_given_stats needs of svgpfa_latent_stats -- the statistics at the quadrature nodes and, per (neuron,
 * latent), the sum of the posterior means at that neuron's spikes -- kept resident, without materialising the mean and
 * variance at every spike (the embedding M-step of the EM loop, stats/svEM.py:225-232, needs nothing else of them).
 * C: an N x K embedding matrix; its values do not enter the frozen statistics (the pass that makes the spike means also
 * accumulates C-weighted sums, which are discarded). */
int svgpfa_freeze_stats(svgpfa_engine* e, const double* const* mean, const double* const* cholvec, const double* C);

/* KL divergence term alone (stats/klDivergence.py:18-44) of the parameters of the latest svgpfa_elbo /
 * svgpfa_latent_stats / svgpfa_predict call: out[0]. */
int svgpfa_kl(svgpfa_engine* e, double* out);

/* SVLowerBound.evalELLSumAcrossTrialsAndNeurons(svPosteriorOnLatentsStats=...)
 * (stats/svLowerBound.py:72-75): ELL with the stats of the last svgpfa_latent_stats call frozen.
 * out: [0] ELL, then dELL/dC (N x K), dELL/dd (N)  -- length 1 + N*K + N. */
int svgpfa_ell_given_stats(svgpfa_engine* e, const double* C, const double* d,
                           int want_grad, double* out);

/* SVPosteriorOnLatentsAllTimes.predict (stats/svPosteriorOnLatents.py:57-77):
 * times R x T (same T for every trial); mu/var R x T x K. */
int svgpfa_predict(svgpfa_engine* e,
                   const double* const* mean, const double* const* cholvec,
                   const double* times, int32_t n_times, double* mu, double* var);

/* KernelsMatricesStore getters used by the reference's tests
 * (stats/kernelsMatricesStore.py:119-120,176-181): copy the matrices of latent k out.
 * Kzz / Kzz_chol: R x M_k x M_k.  Ktz blocks are recomputed: quad R x Q x M_k, spikes S x M_k. */
int svgpfa_get_kzz(svgpfa_engine* e, int32_t k, double* out);
int svgpfa_get_kzz_chol(svgpfa_engine* e, int32_t k, double* out);
int svgpfa_get_ktz_quad(svgpfa_engine* e, int32_t k, double* out);
int svgpfa_get_ktz_spikes(svgpfa_engine* e, int32_t k, double* out);

/* Reuse of the spike sums across value-only evaluations.  The reference's E-step closure (stats/svEM.py:218-223)
 * does not rebuild the spike-time kernel matrices: they depend on the spikes, Z, theta only
 * (kernelsMatricesStore.py:208-221).  Here the corresponding quantity is P0_j = sum_s C[n_s, k] kappa(t_s, z_j),
 * which depends on (spikes, Z, theta, C) and NOT on the variational parameters.  `token` identifies the CONTENT of C:
 * when an evaluation that needs no embedding / kernel / inducing-point gradients (grad_mask without bits 2, 4, 8)
 * sees the same non-zero token as the last value-only evaluation that ran the spike pass, and no full-gradient
 * evaluation, svgpfa_build_kernels or svgpfa_set_spikes has happened since, the spike pass is skipped and P0 is
 * reused (results are bit-identical to the un-cached evaluation).  token = 0 (default)
 * disables the reuse.  The caller must change the token whenever it changes C. */
int svgpfa_set_embedding_token(svgpfa_engine* e, uint64_t token);

/* Timing hook for bench.py.  While enabled (on = 1), every svgpfa_elbo call brackets its two largest kernels -- the spike-time
 * pass (which = 0) and k_item_bwd (which = 1) -- with hipEvent pairs on the stream each is launched on; nothing is
 * waited for at record time.  on = P > 1 SAMPLES instead: k_item_bwd is bracketed in evaluations 0, P, 2P, ... since the call,
 * the spike pass in evaluations P/2, P + P/2, ... -- at most one pair per evaluation (an event record is a packet of its own
 * on the stream: two pairs in every evaluation lengthen a 0.52-ms evaluation by 12 us).  svgpfa_kernel_ms synchronises, adds
 * up the elapsed device time of all pairs of that kernel recorded since svgpfa_enable_timing and returns it with the number
 * of launches that were bracketed.
 * svgpfa_spike_kernel_ms = svgpfa_kernel_ms(e, 0, ...). */
int svgpfa_enable_timing(svgpfa_engine* e, int on);
int svgpfa_kernel_ms(svgpfa_engine* e, int32_t which, double* total_ms, int32_t* launches);
int svgpfa_spike_kernel_ms(svgpfa_engine* e, double* total_ms, int32_t* launches);

/* ---- Vector kernels of the EM loop's optimiser -------------------------------------------------------------------
 * The reference optimises every conditional-maximisation step with torch.optim.LBFGS (stats/svEM.py:266-294); its
 * iteration is ~60 small tensor operations.  These stateless entry points (no engine: device pointers, a HIP stream,
 * a caller workspace of svgpfa_lbfgs_workspace_doubles(n, max_history) doubles) do the vector work of one
 * L-BFGS iteration over ONE flat parameter vector of n doubles in the minimum number of passes, with fixed-order
 * (bit-reproducible) sums.  Entries [0, n_weighted) carry `weight` in every sum (replicated parameters count once across
 * ranks: weight 0 on all ranks but one); maxima are unweighted and propagate NaN.  Used by
 * svgpfa_amd/stats/shardedLBFGS.py; arithmetic and decisions stay those of torch.optim.LBFGS.
 *
 * svgpfa_lbfgs_grad_stats: g = sign * g_raw;  out6 = [g.d (0 when d is NULL), g.g, |g|_1, max|g|, sign * *value,
 *   *carry] (the last two only when the pointers are non-NULL: the closure's value and a scalar carried along, so that the
 *   host reads everything an evaluation produced in one transfer).
 * svgpfa_lbfgs_pair_dots: y = g - g_prev -> Y[m], s = t d -> S[m] (history rows of row_stride doubles; rows 0..m-1 hold
 *   the accepted pairs, oldest first); out[0..5) = [y.s, y.y, g.g, s.g, y.g], out[5 + j m + i] for i < m =
 *   S_i.y (j = 0), Y_i.s (1), Y_i.y (2), S_i.g (3), Y_i.g (4).
 * svgpfa_lbfgs_direction: d = -H g + sum_i cY_i Y_i + sum_i cS_i S_i with coef = [cY (m), cS (m), H] on the device;
 *   x0 = x when both are given (the line search's restart point); out_dmax[0] = max|d|.
 * svgpfa_lbfgs_coef: the coefficient step between the two, on the device (one wavefront): dots = pair_dots' 5 m + 5 sums;
 *   gram = [history_rows^2] s_i.y_j | [history_rows^2] y_i.y_j | H, state kept on the device (history_rows <= 128 rows each way,
 *   > m); accepts the candidate pair when y.s > 1e-10 (its row and column join the matrices, H = y.s / y.y), runs the
 *   two-loop recursion in coefficient space and leaves coef = [cY (m + 1), cS (m + 1), H] for svgpfa_lbfgs_direction over
 *   m + 1 rows (zeros for a rejected candidate) and out4 = [g.d, new number of pairs, accepted (0 / 1), H] -- so that the host
 *   can enqueue the direction and the next evaluation without waiting for the sums. */
int64_t svgpfa_lbfgs_workspace_doubles(int64_t n, int32_t max_history);
int svgpfa_lbfgs_coef(void* stream, int32_t m, int32_t history_rows, const double* dots, double* gram, double* coef,
                      double* out4);
int svgpfa_lbfgs_grad_stats(void* stream, int64_t n, const double* g_raw, double sign, const double* d, double* g,
                            int64_t n_weighted, double weight, const double* value, const double* carry, double* work,
                            double* out6);
int svgpfa_lbfgs_pair_dots(void* stream, int64_t n, const double* g, const double* g_prev, const double* d, double t,
                           double* S, double* Y, int64_t row_stride, int32_t m, int64_t n_weighted, double weight,
                           double* work, double* out);
int svgpfa_lbfgs_direction(void* stream, int64_t n, const double* g, const double* S, const double* Y, int64_t row_stride,
                           int32_t m, const double* coef, double* d, const double* x, double* x0, double* work,
                           double* out_dmax);

#ifdef __cplusplus
}
#endif
#endif /* SVGPFA_H */
