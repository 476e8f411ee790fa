/*
 * ps_api.h — C-ABI of libprecondition_amd.so (MI355X / gfx950).
 *
 * The reference (google-research/precondition, precondition/distributed_shampoo.py,
 * "DS:<line>" below) has no FFI: its preconditioner-compute path is jnp/lax calls
 * lowered by XLA.  These entry points are what a binding at the reference's three
 * internal seams would call:
 *
 *   (i)   gram_weighted_update(old, g, axis, w1, w2)            DS:1440-1470,
 *         called from Preconditioner.updated_statistics_from_grad DS:1588
 *         -> ps_stats_update_f32 / ps_stats_update_grouped_f32
 *   (ii)  _matrix_inverse_pth_root_vmap(xs, ps, padding_starts)  DS:2742-2744
 *         = vmap(matrix_inverse_pth_root DS:702-940 | _eigh DS:943-1030)
 *         -> ps_newton_root_batched_f32 / ps_eigh_root_batched_f32
 *         (with power_iteration DS:595-652 -> ps_power_iteration_batched_f32 and
 *          mat_power DS:655-678 -> ps_mat_power_f32 also exposed, since the
 *          reference's tests call them directly)
 *   (iii) jax.lax.all_gather DS:2876-2877 -> ps_comm_unique_id / ps_comm_init /
 *         ps_comm_allgather / ps_comm_destroy (RCCL resolved at run time).  These are the
 *         binding for a host WITHOUT torch; the Python host of this repository issues the same
 *         collective through torch.distributed (backend "nccl" = RCCL), which owns its
 *         communicator, and brackets asynchronous gathers with ps_collective_in_flight.
 *
 * Conventions
 *   - Plain C, no torch/HIP types in signatures: streams are passed as void*
 *     (hipStream_t), device memory as raw pointers.  All matrices are float32,
 *     row-major, leading dimension in elements.
 *   - Every function returns 0 on success, a negative PS_E* code for an invalid
 *     argument, or a positive hipError_t passed through.  Nothing throws or aborts.
 *   - The caller owns every device buffer, including workspace; sizes come from
 *     the *_workspace_bytes functions.  No hidden device allocation happens inside
 *     a compute call (a few KB of pinned host memory are allocated once per
 *     process for convergence flags).
 *   - Calls are stream-ordered.  Data-dependent iteration counts (DS:836-848, 862-864) are
 *     resolved on the device, but the host has to know when to stop queueing steps, so the
 *     iterative entry points WAIT on the GPU while they run -- always one step behind it, so
 *     that the stream never drains:
 *       ps_newton_root_batched*_f32 (staged execution, the default): one event wait per Newton
 *         step, on the step before the one just queued; returns when the last step is queued.
 *       ps_eigh_root_batched_f32 (blocks of more than 128 rows): one stream synchronisation
 *         after the Cholesky factorisation (fallback blocks must be known), then one event wait
 *         per Jacobi sweep, one sweep behind the GPU.  ps_eigh_batched_f32 on matrices of more
 *         than 128 rows: one stream synchronisation per sweep.
 *       ps_diag_mfma_clock and any call made while ps_profile_enable(1) is in force
 *         synchronise the stream.
 *     Everything else (statistics, products, power iteration, quantisation, transform,
 *     eigendecompositions of <= 128 rows, ps_comm_allgather) only enqueues.
 *   - ps_eigh_root_batched_f32 sweeps its blocks on the caller's stream AND on one internal
 *     side stream per host thread (forked from and joined back into the caller's stream with
 *     events, so the call is stream-ordered as a whole).
 *   - Numerical failure is data, not an error (DS:2936-2950): it is reported in
 *     the metrics table and the function still returns 0.
 */
#ifndef PS_API_H_
#define PS_API_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PS_VERSION 306 /* 0.3.6: ps_options grew (eigh_keep_max_cond), PS_EIGH_ACCURATE */

enum {
  PS_OK = 0,
  PS_EINVAL = -1,     /* bad pointer / size / exponent */
  PS_EWORKSPACE = -2, /* workspace too small */
  PS_EUNSUPPORTED = -3,
  PS_EINTERNAL = -4,
  PS_ECOMM = -6,      /* RCCL not loadable, or an RCCL call failed (ps_comm_last_error) */
  PS_EDEVICE = -5     /* called with a HIP device current that is not the one this
                         process first used the library on (one process per GPU) */
};

/* Columns of the per-block metrics table written by the root functions.
 * 0..4 are the reference's TrainingMetrics fields (DS:338-351, 902-907). */
enum {
  PS_M_ERROR = 0,        /* inverse_pth_root_errors : max|M - I| of the last try */
  PS_M_ITERS = 1,        /* inverse_pth_root_iters  : inner iterations of the last try */
  PS_M_ERROR_RATIO = 2,  /* final_error_ratio */
  PS_M_MAX_EV = 3,       /* max_eigen_value (power iteration) */
  PS_M_RETRIES = 4,      /* total_retries = number of tries (>=1) */
  PS_M_TOTAL_ITERS = 5,  /* inner iterations summed over all tries (for FLOP accounting) */
  PS_M_POWER_ITERS = 6,  /* power-iteration steps executed */
  PS_M_AVG_STEPS = 7,    /* Newton root: number of steps, summed over tries, whose M update was computed in
                            full and averaged with its transpose (ps_newton_averaged_steps; for FLOP
                            accounting).  eigh root rows: the condition number lambda_max / lambda_min of the
                            regularised block from its final eigenvalues (+inf if not positive definite) --
                            next recompute's ps_options.iters_hint */
  PS_METRICS_STRIDE = 8
};

/* Symmetry contract of the matrices handed to the power iteration / Newton root.
 * The reference's functions take ANY square matrix (power_iteration DS:595-652 is a
 * plain mat-vec loop, matrix_inverse_pth_root DS:845-846 plain products); symmetric
 * inputs — which statistics are, except after int16 dequantization (DS:2746-2772) —
 * allow computing only the upper tile triangle of every product / mat-vec pass.
 *   PS_SYMMETRY_VERIFY  (default) every block is tested on the device, a_ij == a_ji bit
 *                       for bit; blocks that fail take the full products.
 *   PS_SYMMETRY_ASSUME  the caller guarantees exact symmetry (skips the test pass).
 *   PS_SYMMETRY_GENERAL full products for every block.
 * For a block on the symmetric path the Newton iterates are kept bitwise symmetric (mirrored
 * off the diagonal tiles, (X + X^T)/2 inside them) and the right operand of every product is
 * read transposed: the returned root equals its transpose bit for bit.  A block passed under
 * PS_SYMMETRY_ASSUME that is NOT exactly symmetric gets the root of a nearby symmetric matrix
 * without a diagnostic: use VERIFY (0.13 ms for 256 x 512^2) unless symmetry is certain. */
enum { PS_SYMMETRY_VERIFY = 0, PS_SYMMETRY_ASSUME = 1, PS_SYMMETRY_GENERAL = 2 };

int ps_version(void);
const char* ps_error_string(int code);

/* ---- per-call options of the root entry points (ps_*_root_batched_opt_f32) -------------------
 * Everything that selects HOW a root is computed is an argument: two callers in one process (or
 * two calls of one caller) may use different modes; nothing below is process state.  The plain
 * entry points (ps_newton_root_batched_f32, ...) are the *_opt_* ones with options = NULL =
 * ps_options_init() defaults.  Zero-initialising the struct and setting struct_size also gives the
 * defaults, except where a field says otherwise.
 *
 * products — arithmetic of the n^3 products of the Newton iteration (DS:845-846, mat_power
 *   DS:671-674).  The reference passes jax.lax.Precision through `precision` (DS:599, 708):
 *     PS_PRODUCTS_F32     (Precision.HIGHEST, the reference's default) exact float32 products on
 *                         v_mfma_f32_32x32x2_f32; the parity path.
 *     PS_PRODUCTS_BF16X6  (Precision.HIGH) three-way bf16 split of both operands, six partial
 *                         products on v_mfma_f32_32x32x16_bf16, ~2^-22 relative; the last steps
 *                         of a block (max|M - I| < 1e-3) run in exact float32.
 *     PS_PRODUCTS_BF16X3  (Precision.DEFAULT) two-way split, three partial products, ~2^-16
 *                         relative; exact float32 once max|M - I| < 3e-2.
 *   The bf16 modes apply to exactly symmetric blocks; other blocks run float32 in the same call.
 * accumulation — how a float32 product sums over k (PS_PRODUCTS_F32 only; the call's, not a
 *   block's: a block's bits never depend on what else is in the call):
 *     PS_ACCUM_SEGMENTED  (default) every product of the Newton iteration is summed in segments
 *                         of 128 values of k (blocked summation: the fp32 MFMA is one fmaf chain
 *                         per output element, whose rounding grows like sqrt(K)).  On
 *                         ill-conditioned blocks (cond ~5e3, p = 4: where two float32 evaluations of
 *                         the iteration differ by more than 1e-4) the root is then closer to the
 *                         float64 root than NumPy/OpenBLAS's float32 evaluation of the same
 *                         iteration: 0.55-0.65 x its error (one chain: 1.3-1.6 x).  Same MFMA
 *                         work; +0 % time at 512^2, +3 % at 1024^2.
 *     PS_ACCUM_CHAIN      one fmaf chain over the whole k range (rounds 1-3).
 * averaged_steps — leading Newton steps of a try whose M update is computed in full and averaged
 *   with its transpose instead of being mirrored (section "symmetry" below); -1 = default (4).
 * iters_hint / iters_hint_stride / fast_max_iters — HOST pointer to one float per block (block b
 *   at iters_hint[b * iters_hint_stride]): the block's inverse_pth_root_iters at the PREVIOUS
 *   recompute (a host copy of column PS_M_ITERS of that call's metrics, stride PS_METRICS_STRIDE;
 *   the optimizer reads that table on the host anyway for the failure select of DS:2936-2950 and
 *   its state carries it, DS:338-351).  A block whose hint is in [1, fast_max_iters] (default 8:
 *   condition number below ~1e2) is well conditioned: it takes 0 averaged steps -- mirrored M
 *   updates are exact to 1e-6 there -- which saves the 6 extra tile products of an averaged step
 *   (-7 % at 256 x 512^2, -9 % at 64 x 1024^2).  NULL, 0 or NaN = no hint = the careful path.
 *   ps_eigh_root_batched_opt_f32 (eigh_solver AUTO) reads the same array differently: the block's condition
 *   number at the previous recompute (column PS_M_AVG_STEPS of that call's metrics: lambda_max / lambda_min of
 *   the regularised block, +inf if not positive definite).  Above 2e3 (twice the keep rule's bound) the block goes
 *   to the Jacobi solvers directly -- same bits as after a hand-over, minus the time of the fast path's attempt.
 *   Statistics move slowly (beta2 ~ 0.999), so last recompute's count is a sound predictor; a
 *   block that turns out slower than its hint is only less accurate (mirror noise ~ cond * eps).
 * execution — PS_EXEC_STAGED (default: one launch per product stage, one host event wait per
 *   Newton step, one step behind the GPU) or PS_EXEC_PERSISTENT (one dataflow kernel, no host wait).
 * power_iteration — PS_PI_AUTO (resident when the chip is free, else streaming), PS_PI_STREAMING,
 *   PS_PI_RESIDENT (= AUTO: residency is never forced); pi_timeout_ms: deadline of a resident
 *   launch's waits, < 0 = default 5000, 0 = every wait counts as expired (tests).
 * eigh_* — ps_eigh_root_batched_opt_f32: eigh_sweep_tol (<= 0: default 2e-6) is the scaled
 *   off-diagonal bound that ends the one-sided Jacobi sweeps; eigh_streams (0: default 2);
 *   eigh_solver: PS_EIGH_AUTO (default): matrices of 129 ... 4096 rows take the Householder
 *   tridiagonalisation + float64 divide and conquer + compact-WY back-transformation
 *   (csrc/eigh_td.hip.h; the algorithm class of LAPACK's ssyevd, which the reference's
 *   jnp.linalg.eigh runs in float32, DS:35-38, DS:1007).  ROOT calls keep its result for every block:
 *   measured against the float64 root of the same float32 matrix it is at or below the error of a true
 *   float32 ssyevd on every statistics family tried (Wishart, graded 1e2 ... 1e6, log-uniform, rank
 *   deficient + ridge, low-rank EMA; 169 ... 2048 rows: profiles/r06_eigh_keep_rule.json: 0.1 ... 1.25 x
 *   ssyevd's error), i.e. at the reference's own accuracy, 2-3 x faster than the Jacobi hand-over that was
 *   the default until round 6 (which compared against NumPy's float64-internal eigh by mistake).  Only an
 *   iteration cap or non-finite input hands a block to the Jacobi solvers.  PS_EIGH_ACCURATE: the previous
 *   rule -- a block keeps the fast path's result only if it is positive definite with lambda_max /
 *   lambda_min <= eigh_keep_max_cond (default 1e3); the others (graded spectra, rank-deficient statistics,
 *   indefinite input) are solved again inside the same call by the Jacobi solvers, which are accurate
 *   RELATIVE to each eigenvalue (10 ... 300 x closer to the float64 root than ssyevd on such inputs, at
 *   2-3 x the time).  PLAIN eigenpairs (ps_eigh_batched_opt_f32: _low_rank_root's smallest eigenpairs,
 *   DS:1071) keep the ACCURATE rule under AUTO as well.  PS_EIGH_TRIDIAGONAL keeps the fast path's result
 *   for every block in both modes; PS_EIGH_ONE_SIDED (Hestenes block Jacobi on the float64-accumulated
 *   Cholesky factor) and PS_EIGH_TWO_SIDED (blocked two-sided Jacobi on the matrix itself + float64
 *   re-projection) skip the fast path.  eigh_keep_max_cond (> 0) overrides the bound of whichever rule
 *   applies (+inf = keep everything).
 *   ABI note (PS_VERSION 306): the values of this enum changed twice -- round 5 (still reporting 300)
 *   renumbered PS_EIGH_ONE_SIDED 0 -> 2 when 0 became AUTO; 306 adds PS_EIGH_ACCURATE = 4 and the field
 *   eigh_keep_max_cond (struct_size grows: a caller built against the shorter struct gets the default).
 *   Check ps_version() >= 306 before relying on either.
 * The PS_* environment variables of earlier rounds survive only as a developer override, read in
 * ONE function (csrc/options.cpp ps_dev_env_overrides) and only when PS_DEV_ENV=1 is set. */
enum { PS_PRODUCTS_F32 = 0, PS_PRODUCTS_BF16X6 = 1, PS_PRODUCTS_BF16X3 = 2 };
enum { PS_ACCUM_SEGMENTED = 0, PS_ACCUM_CHAIN = 1 };
enum { PS_EXEC_STAGED = 0, PS_EXEC_PERSISTENT = 1 };
enum { PS_PI_AUTO = 0, PS_PI_STREAMING = 1, PS_PI_RESIDENT = 2 };
enum { PS_EIGH_AUTO = 0, PS_EIGH_TWO_SIDED = 1, PS_EIGH_ONE_SIDED = 2, PS_EIGH_TRIDIAGONAL = 3, PS_EIGH_ACCURATE = 4 };
typedef struct {
  uint32_t struct_size;        /* sizeof(ps_options) of the caller's build */
  int32_t products;            /* PS_PRODUCTS_* */
  int32_t accumulation;        /* PS_ACCUM_* */
  int32_t averaged_steps;      /* -1 = default */
  const float* iters_hint;     /* HOST array, may be NULL */
  int32_t iters_hint_stride;   /* in floats; 0 is read as 1 */
  int32_t fast_max_iters;      /* <= 0 = default (8) */
  float averaged_err_threshold;/* > 0: steps after the second stop averaging once max|M - I| <= it */
  int32_t execution;           /* PS_EXEC_* */
  int32_t power_iteration;     /* PS_PI_* */
  int32_t pi_timeout_ms;       /* < 0 = default */
  float eigh_sweep_tol;        /* <= 0 = default */
  int32_t eigh_streams;        /* 0 = default */
  int32_t eigh_solver;         /* PS_EIGH_AUTO (0, default) | PS_EIGH_TWO_SIDED | PS_EIGH_ONE_SIDED | PS_EIGH_TRIDIAGONAL | PS_EIGH_ACCURATE */
  int32_t reserved[5];         /* reserved[0] = PS_OPTIONS_MAGIC (written by ps_options_init), the rest 0 */
  float eigh_keep_max_cond;    /* <= 0 = the solver rule's default; > 0 (may be +inf): keep the fast path's result up to this lambda_max / lambda_min */
  int32_t reserved2[3];        /* 0 */
} ps_options;
/* Fills *opt with the defaults (struct_size = sizeof(ps_options), reserved[0] = PS_OPTIONS_MAGIC).
 * EVERY ps_options must start from this call: a zero-initialised struct is NOT the defaults
 * (pi_timeout_ms = 0 would mean "every resident wait expires", averaged_steps = 0 "no averaged
 * updates"), so a full-size struct without the marker is refused with PS_EINVAL instead of
 * silently changing the numerics. */
#define PS_OPTIONS_MAGIC 0x5053

void ps_options_init(ps_options* opt);

/* v0 of power_iteration: first n values of
 * numpy.random.RandomState(1729).uniform(-1, 1, n).astype(float32) (DS:642-643),
 * reproduced with a built-in MT19937 so the library needs no NumPy. Host buffer. */
int ps_power_iteration_v0(int n, float* out_host);

/* ---- statistics: S <- w1*S + w2*tensordot(g, g, all axes but `axis`) ----------
 * Replaces gram_weighted_update DS:1468-1470.  The Gram matrix of a gradient
 * block along one axis is a sum over `nseg` segments of X_s X_s^T, where X_s is
 * the [d, k] matrix whose element (i, kk) lives at
 *     layout 0 ("k-contiguous"):  g[s*seg_stride + i*ld + kk]
 *     layout 1 ("d-contiguous"):  g[s*seg_stride + kk*ld + i]
 * which covers every axis of a (strided view of a) 2-D or 3-D block without a
 * copy: for a row-major [m, n] block with leading dimension L, axis 0 is
 * {layout 0, d=m, k=n, ld=L, nseg=1} and axis 1 is {layout 1, d=n, k=m, ld=L,
 * nseg=1}; the middle axis of a [b0,b1,b2] view with strides (s0,s1,1) is
 * {layout 0, d=b1, k=b2, ld=s1, nseg=b0, seg_stride=s0}.
 * stat_out may alias stat_in (in-place).  desc is a HOST array; all pointers
 * inside are device pointers.  One launch covers the whole
 * Python-unrolled loop of DS:1582-1590. */
typedef struct {
  const float* g;
  int32_t layout; /* 0 or 1 */
  int32_t d;      /* statistic is [d, d] */
  int32_t k;      /* contraction length per segment */
  int32_t nseg;   /* >= 1 */
  int64_t ld;
  int64_t seg_stride;
  const float* stat_in;
  float* stat_out;
  int64_t lds;    /* leading dimension of stat_in / stat_out */
} ps_stats_desc;

size_t ps_stats_update_grouped_workspace_bytes(const ps_stats_desc* desc, int count);
int ps_stats_update_grouped_f32(void* stream, const ps_stats_desc* desc, int count,
                                float w1, float w2, void* workspace,
                                size_t workspace_bytes);

/* Convenience form for one row-major [rows, cols] matrix (leading dimension ldg):
 * axis 0: S is [rows,rows] (+= g g^T);  axis 1: S is [cols,cols] (+= g^T g).
 * Needs no workspace (the task table is passed by value). */
int ps_stats_update_f32(void* stream, const float* g, int64_t rows, int64_t cols,
                        int64_t ldg, int axis, const float* stat_in,
                        float* stat_out, int64_t lds, float w1, float w2);

/* ---- power iteration (DS:595-652), batched ------------------------------------
 * a[b]: device pointer to an [n[b], n[b]] matrix (ld = lda[b]); padding_start may
 * be NULL (no padding) or per-block values (rows/cols >= it are treated as zero).
 * out_lambda[b] (device) receives s of the last executed step, out_iters[b]
 * (device, may be NULL) the number of steps executed.  out_v (device, may be
 * NULL) receives the normalised vector, n_max floats per block.  Host arrays:
 * a, n, lda, padding_start.  symmetry: PS_SYMMETRY_* (the matrix need not be symmetric). */
size_t ps_power_iteration_workspace_bytes(int batch, const int32_t* n);
int ps_power_iteration_batched_f32(void* stream, const float* const* a,
                                   const int32_t* n, const int32_t* lda,
                                   const int32_t* padding_start, int batch,
                                   int num_iters, float error_tolerance,
                                   float* out_lambda, int32_t* out_iters,
                                   float* out_v, int32_t ldv, int symmetry,
                                   void* workspace, size_t workspace_bytes);

int ps_power_iteration_batched_opt_f32(void* stream, const float* const* a,
                                       const int32_t* n, const int32_t* lda,
                                       const int32_t* padding_start, int batch,
                                       int num_iters, float error_tolerance,
                                       float* out_lambda, int32_t* out_iters,
                                       float* out_v, int32_t ldv, int symmetry,
                                       void* workspace, size_t workspace_bytes,
                                       const ps_options* options);

/* ---- mat_power (DS:655-678): out = m^p, same multiplication order ------------ */
size_t ps_mat_power_workspace_bytes(int n, int p);
int ps_mat_power_f32(void* stream, const float* m, int n, int ldm, int p,
                     float* out, int ldo, void* workspace, size_t workspace_bytes);

/* ---- batched inverse p-th root, coupled Newton (DS:702-940 under vmap) -------
 * Host arrays of length batch: a (device pointers), n, lda, p, padding_start
 * (NULL = none), out (device pointers), ldo.  metrics: device, [batch][8] floats.
 * Semantics per block equal matrix_inverse_pth_root(a, p, num_iters, ridge_epsilon,
 * error_tolerance, relative_matrix_epsilon, padding_start=...): power iteration
 * for the relative epsilon, <=6 tries with ridge*10^i, inner loop while
 * it<num_iters && err>tol && ratio<1.2, previous H returned if the last step
 * diverged, all-padding blocks forced to 0.  A 1x1 block runs the same iteration
 * (as it does in the reference whenever it is padded to max_size, DS:2841-2843;
 * the reference's unpadded matrix_size == 1 branch raises, DS:850-855/907).
 * symmetry: PS_SYMMETRY_* above.
 * Two executions of the same tile code, bit-identical results (tests/test_gpu_round2.py):
 *   staged (default): one launch per product stage for the whole batch + one control launch
 *     per step; the loop condition of DS:836-848 is evaluated on the device, the host only
 *     reads "blocks still running" from pinned memory one iteration behind the GPU (one
 *     event wait per Newton step).  iters_executed_host (may be NULL) receives the number
 *     of step rounds the host issued.
 *   persistent (PS_NEWTON_PERSISTENT=1): the call only enqueues work; init, products, loop
 *     control, retries and copy-out run in ONE persistent kernel whose workgroups pull
 *     (block, product, tile) items from device-side queues; a block's next product is
 *     released by the last tile of the products it depends on (DS:844-846).
 *     iters_executed_host is set to -1 (per-block counts are in the metrics table).
 *   The staged execution is the faster one on MI355X today. */
size_t ps_newton_root_workspace_bytes(int batch, const int32_t* n,
                                      const int32_t* p,
                                      const int32_t* padding_start);
/* Number of leading Newton steps of a try in which the M update (DS:845) of an exactly symmetric
 * block is computed in full and averaged with its transpose instead of being mirrored from its
 * upper tile triangle (default 4; PS_NEWTON_AVG_STEPS overrides, 0 = mirrored everywhere).  With
 * the opt-in PS_NEWTON_AVG_ERR=t, steps after the second stop averaging once max|M - I| <= t
 * (csrc/newton.hip newton_avg_next: at cond ~7e3, p = 4 the error against the float64 root is
 * 9.3e-4 mirrored everywhere, 2.8e-4 with 2 averaged steps, 1.76e-4 with 4, 1.96e-4 with full
 * products in this library's summation order and 1.26e-4 in NumPy's).  The count per block is
 * column PS_M_AVG_STEPS of the metrics table. */
int ps_newton_averaged_steps(void);
int ps_newton_root_batched_f32(void* stream, const float* const* a,
                               const int32_t* n, const int32_t* lda,
                               const int32_t* p, const int32_t* padding_start,
                               int batch, int num_iters, float ridge_epsilon,
                               float error_tolerance, int relative_matrix_epsilon,
                               int symmetry, float* const* out, const int32_t* ldo,
                               float* metrics, void* workspace,
                               size_t workspace_bytes, int32_t* iters_executed_host);

/* The same call with options (NULL = defaults).  max_ev: NULL, or the device array of
 * ps_newton_root_batched_maxev_f32 (then relative_matrix_epsilon is taken as 1). */
int ps_newton_root_batched_opt_f32(void* stream, const float* const* a,
                                   const int32_t* n, const int32_t* lda,
                                   const int32_t* p, const int32_t* padding_start,
                                   int batch, int num_iters, float ridge_epsilon,
                                   float error_tolerance, int relative_matrix_epsilon,
                                   const float* max_ev, int symmetry, float* const* out,
                                   const int32_t* ldo, float* metrics, void* workspace,
                                   size_t workspace_bytes, int32_t* iters_executed_host,
                                   const ps_options* options);

/* Same with the largest eigenvalue GIVEN (device array max_ev[batch]) instead of the
 * power iteration: the lobpcg_topk_precondition branch of matrix_inverse_pth_root already
 * has it from the top-k eigenpairs (DS:813-817) and roots the DEFLATED matrix (DS:804-812).
 * Always the relative-epsilon form: ridge = ridge_epsilon * max(max_ev[b], 1e-25). */
int ps_newton_root_batched_maxev_f32(void* stream, const float* const* a,
                                     const int32_t* n, const int32_t* lda,
                                     const int32_t* p, const int32_t* padding_start,
                                     int batch, int num_iters, float ridge_epsilon,
                                     float error_tolerance, const float* max_ev,
                                     int symmetry, float* const* out, const int32_t* ldo,
                                     float* metrics, void* workspace,
                                     size_t workspace_bytes, int32_t* iters_executed_host);

/* ---- batched inverse p-th root by symmetric eigendecomposition (DS:943-1030) - */
size_t ps_eigh_root_workspace_bytes(int batch, const int32_t* n);
int ps_eigh_root_batched_f32(void* stream, const float* const* a, const int32_t* n,
                             const int32_t* lda, const int32_t* p,
                             const int32_t* padding_start, int batch,
                             float ridge_epsilon, float error_tolerance,
                             int relative_matrix_epsilon, float* const* out,
                             const int32_t* ldo, float* metrics, void* workspace,
                             size_t workspace_bytes);

int ps_eigh_root_batched_opt_f32(void* stream, const float* const* a, const int32_t* n,
                                 const int32_t* lda, const int32_t* p,
                                 const int32_t* padding_start, int batch,
                                 float ridge_epsilon, float error_tolerance,
                                 int relative_matrix_epsilon, float* const* out,
                                 const int32_t* ldo, float* metrics, void* workspace,
                                 size_t workspace_bytes, const ps_options* options);

/* ---- plain batched symmetric eigendecomposition (jnp.linalg.eigh, DS:1007/1071) ----
 * a[b]: symmetric [n[b], n[b]].  evals[b]: n[b] floats; evecs[b]: [n[b], n[b]] with
 * eigenvectors in COLUMNS (ld = ldv[b]).  Matrices with n <= ps_eigh_sorted_max_n() (128:
 * solved by the LDS-resident single-launch kernel, the whole call only enqueues work) come
 * back in LAPACK's ASCENDING order; larger ones in the Jacobi order (NOT sorted): callers
 * that need ascending order sort those pairs.  Used by the low-rank /
 * Frequent-Directions branch (_low_rank_root DS:1033-1120, _fd_update_root DS:1123-1290).
 * Workspace: ps_eigh_root_workspace_bytes. */
int ps_eigh_sorted_max_n(void);
int ps_eigh_batched_f32(void* stream, const float* const* a, const int32_t* n,
                        const int32_t* lda, int batch, float* const* evals,
                        float* const* evecs, const int32_t* ldv, void* workspace,
                        size_t workspace_bytes);
/* The same with per-call options (eigh_solver, eigh_sweep_tol, eigh_streams of ps_options; NULL = defaults). */
int ps_eigh_batched_opt_f32(void* stream, const float* const* a, const int32_t* n,
                            const int32_t* lda, int batch, float* const* evals,
                            float* const* evecs, const int32_t* ldv, void* workspace,
                            size_t workspace_bytes, const ps_options* options);

/* ---- optional per-kernel timing of the Newton driver (bench/roofline only) ----
 * When enabled, ps_newton_root_batched_f32 brackets the product-stage launches of every
 * Newton step (one launch per product stage, back to back) with a pair of HIP events on the
 * caller's stream and, before returning, accumulates their elapsed times (this makes the call
 * synchronous).  (A pair around EVERY launch perturbs what it measures: an event record is a
 * queue packet of its own, and the next kernel no longer starts under the tail of the previous.)
 * ps_profile_get: total milliseconds and launch count of newton_stage_kernel,
 * milliseconds of the power-iteration launches, and of everything else the
 * call enqueued (init/control/copy-out), since the last ps_profile_reset. */
int ps_profile_enable(int on);
int ps_profile_reset(void);
int ps_profile_get(double* stage_ms, int64_t* stage_launches, double* power_iter_ms,
                   double* other_ms);

/* ---- plain batched product C = op(A) * op(B), row-major ------------------------
 * transa = 0: A is [m,k] (lda >= k);  transa = 1: A is stored [k,m] (lda >= m).
 * transb = 0: B is [k,n] (ldb >= n);  transb = 1: B is stored [n,k] (ldb >= k).
 * Used for the reference's preconditioned_grad contraction
 * tensordot(g, P, axes=[[0],[0]]) (DS:1707; transa = 1) and by tests. */
int ps_gemm_f32(void* stream, int transa, int transb, const float* a, const float* b,
                float* c, int m, int n, int k, int lda, int ldb, int ldc, int batch,
                int64_t stride_a, int64_t stride_b, int64_t stride_c);

/* Grouped form: many independent products in one launch per operand-layout pair.
 * Used by the every-step application of the preconditioners to all gradient
 * blocks of a parameter tree (Preconditioner.preconditioned_grad, DS:1645-1708: per
 * block tensordot(g, P, [[0],[0]]) per axis; the reference unrolls it in Python).
 * c may be a strided view (ldc) so that results land directly in the merged
 * gradient (no merge_partitions copy).  desc is a HOST array of device pointers. */
typedef struct {
  const float* a;
  const float* b;
  float* c;
  int32_t m, n, k;
  int32_t transa, transb; /* as in ps_gemm_f32 */
  int64_t lda, ldb, ldc;
} ps_gemm_desc;

size_t ps_gemm_grouped_workspace_bytes(const ps_gemm_desc* desc, int count);
int ps_gemm_grouped_f32(void* stream, const ps_gemm_desc* desc, int count, void* workspace,
                        size_t workspace_bytes);

/* The same grouped product as a reusable plan: the task / tile tables are built and uploaded into
 * the caller's workspace once (create), and every launch is one or two kernel launches with no
 * host-side table building.  The operand POINTERS, shapes and the workspace must stay unchanged
 * and alive while the plan is used; the operands' CONTENTS may change between launches (the
 * subspace iteration of the FD branch repeats the same products on the same buffers every round).
 * Launches of one plan must be stream-ordered with each other (split-K partials live in the
 * workspace).  destroy frees the host handle only. */
typedef struct ps_gemm_plan ps_gemm_plan;
int ps_gemm_grouped_plan_create(void* stream, const ps_gemm_desc* desc, int count, void* workspace,
                                size_t workspace_bytes, ps_gemm_plan** plan);
int ps_gemm_grouped_plan_launch(void* stream, const ps_gemm_plan* plan);
int ps_gemm_grouped_plan_destroy(ps_gemm_plan* plan);

/* ---- bf16-MFMA products of the Frequent-Directions branch (BASELINE configs[4]) --------
 * _fd_update_root (DS:1123-1290) needs the leading rank+1 singular pairs of the d x (rank+d)
 * update; the reference takes a full SVD (DS:1193), this build a block subspace iteration
 * whose cost is d x d @ d x b products (b ~ rank + 32).  Those run on the bf16 MFMA with
 * float32 accumulation: C = A * Bt^T with both operands K-CONTIGUOUS bf16 arrays, A [m][k]
 * and Bt [n][k].  Each operand is either one bf16 array (a_lo / b_lo NULL: 2^-9 relative
 * operand precision) or a hi/lo pair x = hi + lo (2^-17: hi*hi + lo*hi + hi*lo are
 * accumulated).  Requirements: k % 32 == 0, lda % 8 == 0, ldb % 8 == 0, 16-byte aligned
 * bases (PS_EUNSUPPORTED otherwise).
 * ps_convert_f32_to_bf16 produces the operands: dst_hi[r][c] = bf16(src[r][c]) (round to
 * nearest even), dst_lo (may be NULL) = bf16(src - hi); transpose = 1 writes dst[c][r];
 * transpose = 2 writes the tile-blocked layout of a left operand (a_tiled): tile (i, j) of 128
 * rows x 32 columns is contiguous at ((i * (cols / 32) + j) * 4096) elements, so that a product
 * streams whole DRAM pages of it (cols % 32 == 0; the destination holds ceil(rows/128)*128*cols
 * elements, ldd is ignored); transpose = 3 writes the FRAGMENT-MAJOR layout of ps_fd_cy_step_f32
 * (rows % 64 == 0, cols % 64 == 0, rows * cols elements, ldd ignored): element (r, k) at
 * (((r / 64) * (cols / 16) + k / 16) * 2 + (r % 64) / 32) * 512 + (32 * ((k % 16) / 8) + r % 32) * 8
 * + k % 8, i.e. the 64 lanes x 16 bytes one v_mfma_f32_32x32x16_bf16 consumes are one contiguous
 * kilobyte and a 64-row panel is one sequential stream over k. */
typedef struct {
  const void* a_hi; const void* a_lo;   /* bf16 [m][k], leading dimension lda (elements) */
  const void* b_hi; const void* b_lo;   /* bf16 [n][k], leading dimension ldb */
  float* c;                             /* float32 [m][n], leading dimension ldc */
  int32_t m, n, k;
  int64_t lda, ldb, ldc;
  int32_t a_tiled;   /* 1: the A planes are tile-blocked (ps_convert_f32_to_bf16, transpose = 2):
                        [ceil(m/128)][k/32][128][32], rows past m zero; lda is ignored.
                        2: fragment-major planes (transpose = 3) -- accepted by
                        ps_fd_filter_round_f32 only (ps_gemm_bf16_grouped: PS_EUNSUPPORTED) */
  int32_t symmetric; /* != 0: c = a a^T (b_* == a_*, ldb == lda, m == n, a_tiled == 0): only the
                        upper tile triangle is multiplied, every tile is also stored as its mirror
                        image; the result is bitwise symmetric */
} ps_gemm_bf16_desc;

int ps_convert_f32_to_bf16(void* stream, const float* src, void* dst_hi, void* dst_lo,
                           int64_t rows, int64_t cols, int64_t lds, int64_t ldd,
                           int transpose);
size_t ps_gemm_bf16_grouped_workspace_bytes(const ps_gemm_bf16_desc* desc, int count);
int ps_gemm_bf16_grouped(void* stream, const ps_gemm_bf16_desc* desc, int count,
                         void* workspace, size_t workspace_bytes);

/* ---- fused _transform_grad for a whole parameter tree (DS:3496-3625) -------------
 * Grafting, norm matching of the preconditioned gradient, weight decay, momentum /
 * Nesterov for every parameter in three launches.  All arrays of one parameter are
 * contiguous with `numel` float32 elements.  pgrad == NULL marks a parameter whose
 * preconditioning is skipped (precond_grad = grafting_update, DS:3557-3561).
 * diag_in/diag_out are required for Adagrad/RMSProp grafting types only; param only
 * when weight_decay != 0.  Outputs must not alias inputs of OTHER parameters;
 * x_out may alias x_in of the same parameter (each element is read then written by
 * one thread). */
typedef struct {
  const float* grad;
  const float* pgrad;
  const float* param;
  const float* diag_in;
  float* diag_out;
  const float* mom_in;
  float* mom_out;
  const float* dmom_in;
  float* dmom_out;
  float* upd_out;
  int64_t numel;
} ps_transform_desc;

typedef struct {
  int32_t graft_type;  /* GraftingType values of DS:499-506 */
  int32_t nesterov;
  int32_t moving_average_for_momentum;
  int32_t decoupled_learning_rate;
  int32_t decoupled_weight_decay;
  int32_t run_shampoo;  /* step >= start_preconditioning_step */
  float beta1;
  float beta2_w1;       /* beta2 */
  float beta2_w2;       /* 1 - beta2, or 1 if beta2 == 1 (DS:3521-3522) */
  float diagonal_epsilon;
  float weight_decay;
  float lr;             /* learning_rate(step) */
  float clip_by_scaled_gradient_norm; /* <= 0: off */
} ps_transform_config;

size_t ps_transform_grads_workspace_bytes(const ps_transform_desc* desc, int count);
int ps_transform_grads_f32(void* stream, const ps_transform_desc* desc, int count,
                           const ps_transform_config* cfg, void* workspace,
                           size_t workspace_bytes);

/* ---------------------------------------------------------------------------
 * Quantized optimizer state.  Replaces QuantizedValue.quantize
 * (precondition/quantization_utils.py:45-95) and QuantizedValue.to_float
 * (quantization_utils.py:97-113) as used by distributed_shampoo.py for int16
 * statistics / preconditioners with extract_diagonal=True (DS:2087-2106, 2746-2768,
 * 3012-3281) and int8 momentum (DS:2047-2049, 2111-2114, 3617-3619).
 *
 * A tensor of shape [d0, d1, ...] is passed as rows = d0, cols = prod(d1...) (the
 * reference reduces max|x| over axis 0; a 1-D tensor is rows = n, cols = 1).
 *   bucket_size[c] = max_r |x[r,c]| / (127 | 32767)
 *   codes[r,c]     = round_half_even(x[r,c] / (bucket_size[c] > 0 ? bucket_size[c] : 1))
 *   extract_diagonal (rows == cols): diagonal[r] = x[r,r] and x[r,r] counts as 0.
 * Codes, diagonal and bucket sizes are bit-exact with the reference for finite input.
 * Every tensor of the tree goes in ONE call.  ps_dequantize_f32 is one launch.  ps_quantize_f32 reads
 * contiguous float4-addressable matrices of 64 ... 4096 rows and every tensor of fewer than 64 rows ONCE
 * (register-resident column strips; one launch); taller or large strided tensors take two passes over the
 * input (column maxima, then codes: two more launches).  Same codes on every path.
 * `fvalue` is read by ps_quantize_f32 and written by ps_dequantize_f32. */
typedef struct {
  float* fvalue;        /* [rows, cols] float32, leading dimension ld */
  void* codes;          /* int8_t or int16_t [rows, cols], leading dimension ldq */
  float* diagonal;      /* [rows] when extract_diagonal, else may be NULL */
  float* bucket_size;   /* [cols] */
  int64_t rows, cols, ld, ldq;
  int32_t bits;         /* 8 or 16 */
  int32_t extract_diagonal;
} ps_quant_desc;

size_t ps_quantize_workspace_bytes(const ps_quant_desc* desc, int count);
int ps_quantize_f32(void* stream, const ps_quant_desc* desc, int count, void* workspace,
                    size_t workspace_bytes);
size_t ps_dequantize_workspace_bytes(const ps_quant_desc* desc, int count);
int ps_dequantize_f32(void* stream, const ps_quant_desc* desc, int count, void* workspace,
                      size_t workspace_bytes);

/* ---- seam (iii): the exchange of DS:2876-2877 (jax.lax.all_gather of the roots and of the
 * metrics) as plain RCCL calls, one communicator per process (one process per GPU).  The
 * library does not link RCCL: the entry points resolve ncclGetUniqueId / ncclCommInitRank /
 * ncclAllGather / ncclCommDestroy from the librccl.so already loaded in the process (torch's
 * own, when the host is PyTorch) or from the default library path, and return PS_ECOMM when
 * there is none.  The Python host of this repository performs the same exchange through
 * torch.distributed (backend "nccl" = RCCL); these entry points are what a non-torch host
 * binds instead.
 *   ps_comm_unique_id : rank 0 fills PS_COMM_ID_BYTES bytes, the host ships them to all ranks
 *   ps_comm_init      : collective over `world` ranks; *comm receives an opaque handle
 *   ps_comm_allgather : recv[r * bytes_per_rank ...] = rank r's send buffer, enqueued on
 *                       `stream` (device pointers; send may alias its own slot of recv)
 *   ps_comm_destroy   : releases the communicator */
#define PS_COMM_ID_BYTES 128
int ps_comm_unique_id(void* id_out);
int ps_comm_init(void** comm, int rank, int world, const void* unique_id);
int ps_comm_allgather(void* stream, void* comm, const void* send, void* recv,
                      size_t bytes_per_rank);
int ps_comm_destroy(void* comm);
const char* ps_comm_last_error(void);

/* ---- Frequent-Directions branch (BASELINE configs[4]): device glue of the subspace iteration
 * that replaces the SVD of DS:1193 (precondition_amd/subspace.py).
 *   ps_fd_filter_step_f32: one step of the scaled Chebyshev recurrence for `batch` factors whose
 *     iterates are stacked [batch][n][b] (float32, contiguous).  params = device [batch][4]
 *     {ctr, e, sigma1, degree}.  step 1: y' = (z - ctr y) sigma1 / e; step k >= 2:
 *     y' = (z - ctr y) 2 sigma_k / e - sigma_{k-1} sigma_k y_prev with sigma_1 = sigma1,
 *     sigma_m = 1 / (2 / sigma1 - sigma_{m-1}); a factor with degree < step keeps y.  When
 *     yt_hi is given the new iterate is also written transposed as bf16 (hi, and lo = the bf16
 *     of the remainder when yt_lo is given) at yt[c][j * n + r], leading dimension ldt: the
 *     operand layout of ps_gemm_bf16_grouped for the next C @ Y product; ldt = 0 selects the
 *     fragment-major planes of ps_fd_cy_step_f32 instead (n % 64 == 0, b % 32 == 0; factor j at
 *     j * n * b elements: element (k = row r of the iterate, column c) at
 *     ((r / 16) * (b / 32) + c / 32) * 512 + (32 * ((r % 16) / 8) + c % 32) * 8 + r % 8).
 *     y_next must not alias z, y or y_prev.
 *   ps_chol_rinv_batched_f32: out[j] = R^-1 for the Cholesky factor G_j = R^T R of `batch`
 *     symmetric b x b matrices stacked contiguously (float64 arithmetic, b <= ps_chol_rinv_max_n());
 *     a pivot <= drop_rel * max diag(G_j) drops its direction (zero row and column of out[j]). */
int ps_fd_filter_step_f32(void* stream, const float* z, const float* y, const float* y_prev,
                          float* y_next, void* yt_hi, void* yt_lo, const float* params, int step,
                          int batch, int64_t n, int64_t b, int64_t ldt);
/* ps_fd_filter_round_f32: a whole Chebyshev filter (steps 1 .. max_degree of ps_fd_filter_step_f32
 * with the C @ Y product of ps_gemm_bf16_grouped between them) in one call.  desc[j] (HOST array,
 * `batch` entries) describes z_j = C_j * Y_j with b_hi / b_lo pointing INTO yt_hi / yt_lo at
 * factor j's columns (j * n elements, leading dimension ldt) -- the transposed bf16 copies every
 * recurrence step rewrites -- and c = z + j * n * b.  y0 holds the current block (z = C y0 must be
 * current), y1 and y2 are scratch of the same shape; *result_index (host) receives 0, 1 or 2: the
 * buffer that holds the filtered block.  The product's task tables are built and uploaded once per
 * call.  workspace: ps_gemm_bf16_grouped_workspace_bytes(desc, batch). */
/* ps_fd_cy_step_f32: step >= 2 of the same recurrence with its product in ONE launch:
 * z = C_j y (bf16 hi/lo planes, three products, float32 accumulation), y_next = (z - ctr y) 2 sigma_k / e
 * - sigma_{k-1} sigma_k y_prev, and (nt_hi, nt_lo given) y_next as the fragment-major bf16 planes the
 * next step reads.  c_hi / c_lo: HOST arrays of `batch` <= 16 device pointers to the fragment-major
 * covariances (ps_convert_f32_to_bf16, transpose = 3); yt_hi / yt_lo: the planes of y (written by
 * ps_fd_filter_step_f32 with ldt = 0 or by the previous call; nt_* must be other buffers).
 * n % 128 == 0, b in {32, 64, 96}; PS_EUNSUPPORTED otherwise.  A factor with degree < step: y_next = y.
 * ps_fd_filter_round_f32 uses it when every desc[j].a_tiled == 2: yt_hi / yt_lo must then hold TWO
 * copies of the planes (2 * batch * n * b elements each); desc[j].b_*, c and ldt are ignored. */
int ps_fd_cy_step_f32(void* stream, const void* const* c_hi, const void* const* c_lo, int batch,
                      const void* yt_hi, const void* yt_lo, const float* y, const float* y_prev,
                      float* y_next, void* nt_hi, void* nt_lo, const float* params, int step,
                      int64_t n, int64_t b);
/* ps_fd_cx6_f32: z_j = C_j x_j to float32 accuracy on the bf16 MFMA (the Rayleigh-Ritz product of the
 * subspace iteration): both operands as THREE bf16 planes (v = p0 + p1 + p2) and the six products above
 * 2^-24, float32 accumulation.  c0 / c1 / c2: HOST arrays of `batch` <= 16 device pointers to the
 * fragment-major planes of the covariances (ps_convert_f32_to_bf16x3_frag; p0 / p1 are the hi / lo planes
 * of ps_convert_f32_to_bf16 mode 3); x, z: [batch][n][b] float32; xt0..2: scratch of batch * n * b bf16
 * each (the planes of x, written by the call).  n % 128 == 0, b in {32, 64, 96}. */
int ps_convert_f32_to_bf16x3_frag(void* stream, const float* src, void* p0, void* p1, void* p2,
                                  int64_t rows, int64_t cols, int64_t lds);
int ps_fd_cx6_f32(void* stream, const void* const* c0, const void* const* c1, const void* const* c2,
                  int batch, const float* x, float* z, void* xt0, void* xt1, void* xt2, int64_t n,
                  int64_t b);
int ps_fd_filter_round_f32(void* stream, const ps_gemm_bf16_desc* desc, int batch, float* z,
                           float* y0, float* y1, float* y2, void* yt_hi, void* yt_lo,
                           const float* params, int max_degree, int64_t n, int64_t b, int64_t ldt,
                           void* workspace, size_t workspace_bytes, int32_t* result_index);
 /* ps_fd_round_control_f32: per-round control of the subspace iteration for `batch` factors with
 * Ritz values theta [batch][b] (descending) and residual norms res [batch][b]: writes the params
 * rows {ctr, e, sigma1, degree <= `degree`} of the Chebyshev filter, converged[j] (the k wanted
 * pairs have residuals <= tol * theta_1 or sit in the float32 noise floor n * 2.4e-7 * theta_1) and
 * summary = {all converged, max degree, min degree, any wanted relative residual > 2e-2}. */
int ps_fd_round_control_f32(void* stream, const float* theta, const float* res, int batch, int b,
                            int k, int n, float tol, int degree, float* params,
                            int32_t* converged, int32_t* summary);
/* ps_fd_cov_update_f32: c[j] <- 0.5 ((decay c[j] + gram[j]) + (decay c[j] + gram[j])^T) in place for
 * `batch` stacked n x n matrices c (contiguous) and the HOST array gram of device pointers to
 * contiguous n x n matrices: the covariance decay * W W^T + R R^T of DS:1174-1193, symmetrised, in
 * one pass (the result is bitwise symmetric). */
int ps_fd_cov_update_f32(void* stream, float* c, const float* const* gram, int batch, int64_t n,
                         float decay);
/* ps_fd_round_f32: everything of one outer round of the subspace iteration that is not the filter,
 * in ONE call (precondition_amd/subspace.py _Planned; ~35 launches that the Python host issued one by
 * one):  [orthonormalize != 0:  G = X^T X;  M = R^-1 (ps_chol_rinv_batched_f32, drop 1e-10);  T = X M;
 * G = T^T T;  P = 1.5 I - 0.5 G;  X = T P]   Z = C X (c0 given: ps_fd_cx6_f32, else the plan cx);
 * T = X^T Z;  (theta, Y) = eigenpairs of (T + T^T) / 2, descending;  X <- X Y;  Z <- Z Y;
 * res[j][i] = || z_i - theta_i x_i ||;  ps_fd_round_control_f32(theta, res).
 * The eight products are plans of ps_gemm_grouped_plan_create on exactly these buffers:
 *   gram_x: gram = x^T x    xm: tmp = x m     gram_t: gram = tmp^T tmp    pol: x = tmp polish
 *   cx: z = C x             xtz: t = x^T z    xy: tmp = x y               zy: tmp = z y
 * b <= ps_chol_rinv_max_n().  sym, evals, evecs: scratch of batch*b*b, batch*b, batch*b*b floats;
 * eigh_workspace: ps_eigh_root_workspace_bytes(batch, {b, ...}).  Only enqueues. */
typedef struct {
  const ps_gemm_plan* gram_x; const ps_gemm_plan* xm; const ps_gemm_plan* gram_t; const ps_gemm_plan* pol;
  const ps_gemm_plan* cx; const ps_gemm_plan* xtz; const ps_gemm_plan* xy; const ps_gemm_plan* zy;
  const void* const* c0; const void* const* c1; const void* const* c2;   /* HOST arrays, or all NULL */
  void* xt0; void* xt1; void* xt2;                                       /* scratch of ps_fd_cx6_f32 */
  float* x; float* z; float* tmp;                                        /* [batch][n][b] */
  float* gram; float* m; float* polish; float* t; float* y; float* sym;  /* [batch][b][b] */
  float* evals; float* evecs;                                            /* [batch][b], [batch][b][b] */
  float* theta; float* res;                                              /* [batch][b] */
  void* eigh_workspace; size_t eigh_workspace_bytes;
  float* params; int32_t* converged; int32_t* summary;                   /* ps_fd_round_control_f32 */
  int32_t batch, n, b, k, degree, orthonormalize;
  float tol;
  int32_t reserved;
} ps_fd_round_desc;
int ps_fd_round_f32(void* stream, const ps_fd_round_desc* d);

/* ---- ps_fd_update_batched_f32: ONE call per Frequent-Directions sketch update ----------------------
 * Replaces _fd_update_root (DS:1123-1290; called per factor under vmap at DS:2732-2738) for `batch`
 * factors of equal dimension d, sketch rank and exponent p -- SURVEY.md 8(b)'s ps_fd_update_batched:
 *   prepare  W_j = sketch_j sqrt(eigs_j + ridge_j) (DS:1160-1172), C_j = sym(decay W_j W_j^T + Gram_j)
 *            (the SVD of [sqrt(decay) W | R] at DS:1193 <=> the eigenpairs of C), bf16 planes of C_j;
 *   iterate  Chebyshev-filtered block subspace iteration for the leading rank + 1 eigenpairs: per outer
 *            round ps_fd_round_f32 + ONE 16-byte host read + ps_fd_filter_round_f32, <= max_outer rounds;
 *   finish   DS:1196-1290: deflation by the cutoff singular value, tail, sanity masks (unit norm within
 *            1 %), inverted eigenvalues, packing into the [d, rank + 2] layout of DS:555-592.
 * new_grad: HOST array of `batch` device pointers to contiguous d x d matrices: the Gram matrix of the
 *   (averaged) gradient block, or -- input_is_factor != 0 -- a factor R with R R^T = Gram, which is what the
 *   reference keeps in its statistics slot (DS:1497-1505).
 * prev / out: device [batch][d][rank + 2] packed sketches (may not alias); converged: device [batch] int32,
 *   0 marks a factor whose block iteration did not converge in max_outer rounds (its `out` row is then NOT to
 *   be used: the reference-equivalent full decomposition is the caller's fallback).
 * x0: device [batch][d][b] float32 start block, b = ps_fd_block_columns(rank, d) (any full-rank block; the
 *   Python host passes its seeded Gaussian block, which keeps results bit-identical to the per-step path).
 * tol / degree / max_outer: <= 0 selects 1e-5 / 12 / 14.  Only padding_start == d is supported (no padding).
 * PS_EUNSUPPORTED outside the fused kernels' domain (batch <= 16, d % 128 == 0, b in {32, 64, 96},
 * 4 (rank + 33) <= d): the caller takes its general path.  Synchronises the stream once per outer round.
 * info_host (may be NULL): {outer rounds, filter products, b}. */
typedef struct {
  int32_t batch, d, rank, p;
  float decay, ridge_epsilon, error_tolerance;
  int32_t relative_matrix_epsilon;
  int32_t input_is_factor;
  int32_t degree, max_outer;
  float tol;
  const float* const* new_grad;
  const float* prev;
  float* out;
  int32_t* converged;
  const float* x0;
  void* workspace;
  size_t workspace_bytes;
} ps_fd_update_desc;
int ps_fd_block_columns(int rank, int d);
size_t ps_fd_update_workspace_bytes(const ps_fd_update_desc* d);   /* 0 = unsupported shape */
int ps_fd_update_batched_f32(void* stream, const ps_fd_update_desc* d, int32_t* info_host);
int ps_chol_rinv_max_n(void);
int ps_chol_rinv_batched_f32(void* stream, const float* gram, float* out, int b, int batch,
                             float drop_rel);

/* Diagnostic microbenchmark: the fp32 MFMA rate (TFLOP/s) of a loop of 16 MFMAs interleaved with
 * valu_per_16 (0, 4, 8, 16, 32 or 64) independent VALU adds (wide != 0: 64-bit v_lshl_add_u64),
 * with wgs_per_cu (1 or 2) workgroups of 4 wavefronts per CU.  Shows what ordinary VALU work in
 * a K loop costs the fp32 MFMA pipe.  Synchronises the stream.  PS_EINVAL for other counts. */
int ps_diag_mfma_mix(void* stream, int valu_per_16, int wide, int wgs_per_cu,
                     double* mfma_f32_tflops);
/* ---- health of the resident power iteration; diagnostics --------------------------------
 * The resident execution of the power iteration (one launch, matrices in registers) spin-waits
 * on the workgroups of a block's team and is launched at the co-resident capacity of an
 * otherwise IDLE chip.  Kernels that hold CUs on other streams (an RCCL gather) delay team
 * mates; every wait is bounded by one deadline per launch (PS_PI_TIMEOUT_MS, default 5000) and
 * an expired wait is counted in pinned host memory.  The root entry points re-run their call on
 * the streaming kernels when they see the count move at their first host wait, and the process
 * uses the streaming execution from then on (until ps_power_iteration_reset_health).
 *   ps_collective_in_flight(+1 / -1): the host brackets asynchronous collectives with these;
 *       while the count is positive the streaming execution is used.  Returns the new count.
 *   ps_power_iteration_health: expired waits so far, collectives in flight, and whether the
 *       next call would use the resident execution (any pointer may be NULL).
 *   ps_diag_spin: filler kernel (tests): `workgroups` x `threads` with `lds_bytes` of dynamic
 *       LDS each spin for `ms` milliseconds on `stream`.
 *   ps_diag_mfma_clock: shader clock (GHz) held under fp32-MFMA load on non-trivial operands
 *       and the fp32 MFMA TFLOP/s of that loop, after `warm_ms` of back-to-back launches;
 *       synchronises the stream and allocates two small device buffers. */
int ps_collective_in_flight(int delta);
int ps_power_iteration_health(unsigned* expired_waits, int* collectives_in_flight,
                              int* resident_enabled);
int ps_power_iteration_reset_health(void);
int ps_diag_spin(void* stream, int workgroups, int threads, int lds_bytes, double ms);
int ps_diag_mfma_clock(void* stream, double warm_ms, double* clock_ghz, double* mfma_f32_tflops);

#ifdef __cplusplus
}
#endif
#endif /* PS_API_H_ */
