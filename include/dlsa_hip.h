/*
 * dlsa_hip.h -- C ABI of libdlsa_hip.so: the MI355X (gfx950) engine for the DLSA hot path.
 *
 * The reference (feng-li/dlsa) is pure Python on Spark and has no FFI: the entry points
 * below are what a binding for its per-partition map step, one-round reduce and LARS
 * shrinkage would call.  Each entry cites the reference interface it replaces
 * (paths relative to the reference root).
 *
 * Conventions
 *   - every pointer named X, y, w, beta, H, ... is a DEVICE pointer unless marked host;
 *   - matrices are row-major; `ld*` are leading dimensions in ELEMENTS;
 *   - the caller owns all memory; the library never allocates what it returns.  Scratch
 *     comes from a caller-provided device workspace (`ws`, `ws_bytes`), sized by the
 *     matching *_workspace_bytes() query; the workspace must be 256-byte aligned;
 *   - all device work is enqueued on `stream` (a hipStream_t passed as void*; NULL = the
 *     default stream).  Functions that return host scalars synchronise that stream;
 *   - return value: 0 = DLSA_OK, otherwise a dlsa_status; dlsa_last_error() has the text.
 */
#ifndef DLSA_HIP_H
#define DLSA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    DLSA_OK = 0,
    DLSA_ERR_INVALID = 1,       /* bad argument (null pointer, p<=0, ld<p, ...)             */
    DLSA_ERR_HIP = 2,           /* a HIP runtime call failed                                */
    DLSA_ERR_WORKSPACE = 3,     /* workspace too small / misaligned                         */
    DLSA_ERR_NOT_SPD = 4,       /* Cholesky met a non-positive pivot                        */
    DLSA_ERR_NOT_CONVERGED = 5, /* IRLS hit max_iter (results are still written)            */
    DLSA_ERR_NAN = 6,           /* NaN/Inf in a result (reference: warnings.warn, models.py:144) */
    DLSA_ERR_NO_DEVICE = 7
} dlsa_status;

/* per-partition status written by dlsa_irls_fit_f64 (reference soft-fail conventions,
 * models.py:84-91 zero block, :144-145 NaN warning) */
typedef enum {
    DLSA_PART_OK = 0,
    DLSA_PART_NOT_CONVERGED = 1,
    DLSA_PART_NOT_SPD = 2,
    DLSA_PART_NAN = 3,
    DLSA_PART_EMPTY = 4
} dlsa_part_status;

int dlsa_version(void);
/* copies the last error message of the calling thread into buf (NUL-terminated) */
int dlsa_last_error(char* buf, int len);

/* ---- synthetic rows (replaces simulate_logistic, dlsa/models.py:6-40) -----------------
 * Row i is a pure function of (seed, i) (Philox-4x32-10), so any sharding sees the same
 * rows.  kind 0: x ~ U(-0.5,0.5) (models.py:22); kind 1: x ~ N(0,1/12).
 * y (nullable) ~ Bernoulli(sigmoid(x . beta_true)); beta_true nullable = first int(0.4p)
 * coefficients 1, rest 0 (models.py:12,18-19).  If ones_col != 0, column 0 of X is set to
 * 1.0 (the intercept column of models.py:121-122) and features start at column 1. */
int dlsa_synth_f64(uint64_t seed, int64_t row0, int64_t n, int p, int kind, int ones_col,
                   double* X, int64_t ldx, double* y, const double* beta_true, void* stream);
int dlsa_synth_f32(uint64_t seed, int64_t row0, int64_t n, int p, int kind, int ones_col,
                   float* X, int64_t ldx, float* y, const float* beta_true, void* stream);
/* Linear-model response for rows dlsa_synth_* has written (config 5 is "linear model y = X beta* + N(0,1)";
 * the reference has no linear simulator -- README.md:6 only claims the method): y_i = x_i . beta_true + sigma z_i with
 * z_i ~ N(0,1) from the row's own counter stream (counter (i_lo, i_hi, 0, 2), key (seed+1, 0)), so a chunk generated
 * on the device inside a streaming map step is the same for every sharding and for the CPU oracle. */
int dlsa_synth_response_f64(uint64_t seed, int64_t row0, int64_t n, int p, int ones_col, const double* X, int64_t ldx,
                            const double* beta_true, double sigma, double* y, void* stream);
int dlsa_synth_response_f32(uint64_t seed, int64_t row0, int64_t n, int p, int ones_col, const float* X, int64_t ldx,
                            const float* beta_true, double sigma, float* y, void* stream);
/* fp32-NATIVE linear rows, features and response in ONE launch (config 5's stream at its stated size: the chunk generator inside
 * the streaming map step, dlsa_amd.fit_linear_streaming(kind="gaussian32")).  Same model as the two calls above -- x ~ N(0, 1/12),
 * y = x . beta_true + sigma N(0,1), beta_true nullable = first int(0.4p) coefficients 1 -- on its own counter streams
 * (quad q of row i: Philox counter (i_lo, i_hi, q, 3), key (seed, 0), four 24-bit uniforms -> two Box-Muller pairs; noise:
 * counter (i_lo, i_hi, 0, 4), key (seed+1, 0)), every value formed in fp32 with the device's log2 / sqrt / sin / cos
 * instructions: a row is a pure function of (seed, i); the oracle's numpy fp32 restatement agrees to ~1e-6 absolute.
 * y nullable.  Replaces nothing in the reference (it has no linear simulator): bench / test data only. */
int dlsa_synth_linear_f32(uint64_t seed, int64_t row0, int64_t n, int p, int ones_col, float* X, int64_t ldx,
                          const float* beta_true, double sigma, float* y, void* stream);

/* ---- K3: weighted tall-skinny Gram  H = X' diag(w) X  (dlsa/models.py:130) ------------
 * X: n x p, w: n (NULL = all ones, the linear-model X'X), H: p x p (ldh >= p), both
 * triangles written (exactly symmetric).  accumulate != 0 adds into H.
 * fp64 uses v_mfma_f64_16x16x4_f64; fp32 uses v_mfma_f32_16x16x4_f32. */
size_t dlsa_gram_workspace_bytes(int64_t n, int p, int elem_bytes);
int dlsa_gram_f64(const double* X, int64_t ldx, const double* w, int64_t n, int p,
                  double* H, int64_t ldh, int accumulate, void* ws, size_t ws_bytes, void* stream);
int dlsa_gram_f32(const float* X, int64_t ldx, const float* w, int64_t n, int p,
                  float* H, int64_t ldh, int accumulate, void* ws, size_t ws_bytes, void* stream);
/* fp32 rows, fp64 result: the same MFMA passes as dlsa_gram_f32, but the slab partials are summed in fp64 and stored /
 * added (accumulate != 0) into the fp64 matrix H64.  The streaming linear map step (config 5: 6.25e7 rows per GPU do not
 * fit HBM) calls it chunk after chunk; fp32 rounding then stays inside a chunk's slabs. */
int dlsa_gram_f32_acc64(const float* X, int64_t ldx, const float* w, int64_t n, int p,
                        double* H64, int64_t ldh, int accumulate, void* ws, size_t ws_bytes, void* stream);
/* Measurement hook (no reference counterpart; bench.py's `roofline.kernel` and `shader_clock_GHz`, and the tests that
 * must know which kernel variant they exercised): name of the Gram kernel the calling thread's last Gram launch
 * dispatched, e.g. "gram_cyclic_kernel<true,1>", and -- nullable -- the shader cycles wave 0 of workgroup 0 spent in
 * that launch (its s_memtime delta; 0 for kernels without the probe).  Asking for the cycles waits for the launch's
 * stream and reads 8 bytes back.  cycles / kernel time = the clock the chip held (DVFS), which is what turns a
 * box-to-box difference in rows/s into an attributable one. */
int dlsa_gram_last_kernel(char* name, int len, uint64_t* shader_cycles);

/* ---- K1/K2: fused logit pass over the rows (dlsa/models.py:113-114 inner products) ----
 * eta = X beta, mu = sigmoid(eta); writes w_out[i] = mu(1-mu) (nullable), g = X'(y-mu)
 * (p values, nullable) and loglik = sum y log mu + (1-y) log(1-mu) (1 value, nullable).
 * One read of X. */
size_t dlsa_logit_workspace_bytes(int64_t n, int p);
int dlsa_logit_pass_f64(const double* X, int64_t ldx, const double* y, const double* beta,
                        int64_t n, int p, double* w_out, double* g, double* loglik,
                        void* ws, size_t ws_bytes, void* stream);

/* ---- N1: log-likelihood of c estimator columns (dlsa/models.py:217-222) ---------------
 * par: p x c row-major (ldpar >= c), c <= 8; out: c values.  One read of X for all columns.
 * Workspace: dlsa_logit_workspace_bytes(n, p) is sufficient. */
int dlsa_loglik_f64(const double* X, int64_t ldx, const double* y, int64_t n, int p,
                    const double* par, int64_t ldpar, int c, double* out,
                    void* ws, size_t ws_bytes, void* stream);

/* ---- N3: g = X'v and vv = v'v (nullable) in one read of X: the X'y of a linear-model map step
 * (the reference claims linear DLSA, README.md:6, but ships no implementation).  Workspace as
 * dlsa_logit_workspace_bytes. */
int dlsa_xtv_f64(const double* X, int64_t ldx, const double* v, int64_t n, int p,
                 double* g, double* vv, void* ws, size_t ws_bytes, void* stream);
/* fp32 rows (wide-p linear config): fp64 accumulation, fp32 results; workspace 2x the fp64 query */
int dlsa_xtv_f32(const float* X, int64_t ldx, const float* v, int64_t n, int p,
                 float* g, float* vv, void* ws, size_t ws_bytes, void* stream);

/* ---- one FUSED Newton pass (dlsa/models.py:110-114 + :130 in ONE read of the rows) -------------------------------
 * eta = X beta, mu = sigmoid(eta), w = mu(1-mu);  g = X'(y - mu) (p, nullable), loglik (1, nullable), H = X' diag(w) X
 * (p x p, ldh >= p, both triangles), w_out (n, nullable).  The reference evaluates these in three passes over a
 * partition (fit iteration, predict_proba, the Gram of :130); dlsa_logit_pass_f64 + dlsa_gram_f64 in two.  For narrow
 * designs (49 <= p <= 120, even, 16-byte aligned rows, n >= 8192) the rows staged in LDS for the MFMAs also feed the
 * logistic terms: one launch, one read of X (csrc/irls_pass.hip).  Every other shape runs the two launches behind the same
 * entry point (still on the GPU; dlsa_gram_last_kernel tells which form ran).  Results of the two forms agree to 1e-13. */
size_t dlsa_irls_pass_workspace_bytes(int64_t n, int p);
int dlsa_irls_pass_f64(const double* X, int64_t ldx, const double* y, const double* beta, int64_t n, int p,
                       double* H, int64_t ldh, double* g, double* loglik, double* w_out,
                       void* ws, size_t ws_bytes, void* stream);

/* ---- per-call options of the IRLS driver ---------------------------------------------------
 * The driver of dlsa_irls_fit_* / dlsa_onehot_irls_fit_* chooses its measures (partition chains, subsample cold start, frozen /
 * inherited / pooled factors, secant correction, fused passes, predicted exit, ...) from the shapes; every choice gives the same
 * MLE and Hessian to the parity tolerance.  A caller that wants another policy sets it HERE, per host thread: the options stay in
 * force for the calling thread's next fit and workspace-size calls until replaced or cleared (NULL).  Every int field: -1 =
 * automatic (the measured default), 0 = off, 1 = on, unless said otherwise.  The DLSA_IRLS_* environment variables of the A/B
 * scripts under bench/ are consulted only for fields left on automatic.  The reference has no counterpart (sklearn's solver
 * arguments at dlsa/models.py:110-113 are the nearest). */
typedef struct dlsa_irls_options {
    int struct_bytes;    /* sizeof(dlsa_irls_options), set by dlsa_irls_options_init                                   */
    int chains;          /* partition chains (host threads + streams of one call): 1..8; -1 = by the partitions' size  */
    int seeded;          /* fit partition 0 alone and seed every chain with its state                                   */
    int subsample_div;   /* cold start on rows / d of a partition: d >= 2; 0 or 1 = no subsample                        */
    int factor_div;      /* rows / d feed the stand-in Hessian of an inherited start                                    */
    int warm;            /* partition k + 1 starts from partition k's MLE                                               */
    int inherit;         /* full-data iterations start from an inherited factor (default: p >= 192)                     */
    int pool;            /* pooled preconditioner over the finished partitions                                          */
    int secant;          /* quasi-Newton correction of reused factors                                                   */
    int inverse;         /* explicit inverse of a reused factor                                                         */
    int predict;         /* predicted convergence (skips the confirming pass when tol <= 1e-10)                         */
    int fused;           /* fused Newton pass (one read of the rows per fresh Hessian) where the shape allows it        */
    int fuse_last;       /* the iteration expected to end the run takes the fused pass for the result's Hessian         */
    int small;           /* one-launch kernel for many small partitions                                                 */
    int batched;         /* lock-step fit of all partitions of a call together (narrow designs; -1 = by a cost model)  */
    int qn_threads;      /* workgroup size of the quasi-Newton step kernel (64..1024)                                   */
    int trace;           /* print step norms to stderr                                                                  */
    int lean;            /* fits at fused widths write no weight vector (every Hessian from the fused pass)             */
    int small_cluster;   /* workgroups per partition of the one-launch kernel: 1..16; -1 = by the partitions' count    */
    int own_hessian;     /* wide designs: Newton steps preconditioned by the partition's own reduced-precision Hessian */
    int pooled_start;    /* lock step: full-row iterations start from ONE fit on the leading rows of all partitions together */
    int grad_passes;     /* lock step: at most this many gradient-only passes (pooled Hessian) before the Newton passes; 0 = none; -1 = 4 */
    double freeze_at;    /* freeze the factor once steps are below this multiple of max(1, |beta|); 0 = never; < 0 = automatic (1.0) */
} dlsa_irls_options;
/* Kernel switches outside the IRLS driver (round 5): which build of a kernel runs -- never what it returns.  Per thread, like
 * dlsa_irls_options; every field -1 = automatic.  They replace the DLSA_LARS_* / DLSA_OH_ORDERED / DLSA_LOGIT_RING / DLSA_CHOL_SMALL /
 * DLSA_GRAM_NOWIDE / DLSA_GRAM_DBG environment variables of earlier rounds: the shipped library reads no environment variable for
 * them (builds made with -DDLSA_DEBUG_KNOBS, `make knobs`, still do, for fields left on automatic).  No reference counterpart. */
typedef struct dlsa_kernel_options {
    int struct_bytes;    /* sizeof(dlsa_kernel_options), set by dlsa_kernel_options_init                                  */
    int lars_q;          /* LARS form: 0 = lars.hip everywhere; 1 = lars_q.hip up to 1020 variables, lars_c.hip beyond;     */
                         /* 2 = lars_c.hip from 64 variables; automatic: lars_q.hip up to 448, lars_c.hip up to 2044       */
    int lars_q_wgs;      /* workgroups that share lars_q's fused pass: 1..8                                                */
    int lars_q_threads;  /* its workgroup size: 256 | 512 | 1024                                                           */
    int lars_q_lds;      /* its matrices in LDS where they fit; 0 = global memory                                          */
    int lars_wgs;        /* workgroups of lars.hip's grid kernel (1..32) / of lars_c.hip's column split (2..64)            */
    int lars_threads;    /* lars.hip's workgroup size: 512 | 1024                                                          */
    int logit_ring;      /* narrow designs' logit pass through the fused pass's LDS-DMA ring; 0 = register loads           */
    int chol_small;      /* one-launch SPD inverse for p <= 112; 0 = the blocked Cholesky                                  */
    int gram_wide_f32;   /* the fp32 wide Gram kernel (p >= 768); 0 = the panel kernel                                     */
    int onehot_ordered;  /* accumulation of the structured one-hot passes: 0 unordered, 1 ordered floating point (the Gram's default for caller weights is the exact fixed-point mode) */
    int gram_variant;    /* valid-result A/B bits of the fp64 Gram dispatch: 2 | 4 | 8 | 32 | 64 | 256 (gram.hip)          */
    int cooperative;     /* 1 = multi-workgroup kernels launched with hipLaunchCooperativeKernel; default 0: plain launch + bounded barrier (streams created after a cooperative launch serialise on this runtime) */
} dlsa_kernel_options;
void dlsa_kernel_options_init(dlsa_kernel_options* opt);        /* every field on automatic */
int dlsa_kernel_set_options(const dlsa_kernel_options* opt);     /* NULL: back to automatic   */

/* which driver the calling thread's last dlsa_irls_fit_f64 / dlsa_irls_fit_ex_f64 took: 0 = host-driven partition chains, 1 = the
 * one-launch kernel for many small partitions (p <= 64), 2 = lock step (all partitions of the call together, narrow designs) */
int dlsa_irls_last_fit_path(void);
void dlsa_irls_options_init(dlsa_irls_options* opt);            /* every field on automatic */
int dlsa_irls_set_options(const dlsa_irls_options* opt);         /* NULL: back to automatic   */

/* ---- a2-a6: per-partition exact-MLE fit + local quadratic approximation ---------------
 * Replaces logistic_model's numeric core (dlsa/models.py:110-131) for K partitions stored
 * contiguously: partition k = rows [part_offsets[k], part_offsets[k+1]) of X (host array of
 * K+1 int64).  Newton/IRLS from beta=0 until |delta|_inf <= tol*max(1,|beta|_inf).
 * Outputs (device): coef K x p, Sig_inv K x p x p (evaluated at coef, models.py:130),
 * Sig_invMcoef K x p (models.py:131).  Host outputs (nullable): n_iter[K], status[K]
 * (dlsa_part_status), loglik[K].  If X is to carry an intercept, the caller materialises the
 * leading ones column (models.py:121-122) -- see dlsa_synth_f64(ones_col). */
size_t dlsa_irls_workspace_bytes(int64_t max_rows_per_partition, int p);
int dlsa_irls_fit_f64(const double* X, int64_t ldx, const double* y,
                      const int64_t* part_offsets_host, int K, int p,
                      double tol, int max_iter,
                      double* coef, double* Sig_inv, double* Sig_invMcoef,
                      int* n_iter_host, int* status_host, double* loglik_host,
                      void* ws, size_t ws_bytes, void* stream);

/* ---- the same stages WITHOUT copies of the shard (config-3 scale: 2.5e7 x 500 fp64 = 100 GB per GPU) --------------------
 * (1) Implicit intercept.  The reference prepends a ones column for the Hessian (models.py:121-122) and lets sklearn fit the
 *     intercept (models.py:110-113); materialising [1 | X] would be a second 100 GB.  The *_icpt entries take X with its p
 *     columns and treat the intercept as column 0 of a (p + 1)-column design: beta / g / coef have p + 1 entries, H is
 *     (p + 1) x (p + 1), par has p + 1 rows -- intercept first, as in the reference's output frames (models.py:136-142).
 *     The Hessian's border (sum w, X'w) comes from one extra streaming pass.
 * (2) Strided partitions.  partition_id = i % K (models.py:33) makes partition k the rows k, k + K, k + 2K, ...: a strided
 *     view of the shard (row pitch ldx * K), not a gather.  dlsa_irls_fit_ex_f64 takes, per partition, its first row and
 *     row count (host arrays) and one common row_step (1 = contiguous partitions as dlsa_irls_fit_f64); the partition's
 *     labels are gathered into the workspace (8 bytes per row).  Outputs as dlsa_irls_fit_f64 with p + intercept columns. */
int dlsa_logit_pass_icpt_f64(const double* X, int64_t ldx, const double* y, const double* beta, int64_t n, int p,
                             double* w_out, double* g, double* loglik, void* ws, size_t ws_bytes, void* stream);
int dlsa_loglik_icpt_f64(const double* X, int64_t ldx, const double* y, int64_t n, int p, const double* par,
                         int64_t ldpar, int c, double* out, void* ws, size_t ws_bytes, void* stream);
/* ---- Wide Newton pass (round 5): the logit pass of a wide design (121 <= p + intercept <= 512) that, in the same read of the
 * rows, also accumulates a reduced-precision copy of the partition's own Hessian (models.py:110-114 and :130 in one read):
 * w_out (nullable), g, loglik exactly as dlsa_logit_pass[_icpt]_f64; H_approx = [1 | X]' diag(w) [1 | X] from bf16-rounded
 * sqrt(w) x products accumulated in fp32 (relative error ~1e-3 per entry, far less in the spectrum) -- (p + intercept)^2 doubles,
 * both triangles, intercept first.  H_approx is the PRECONDITIONER of the fit's Newton steps (dlsa_irls_fit*_f64 uses it when a
 * partition is eligible); Sig_inv is never taken from it -- the entry point exports H_approx for diagnostics and tests only.
 * dlsa_newton_wide_eligible: 121 <= p + intercept <= 512, >= 32768 rows, ldx >= p; ANY pitch and alignment is served (rows that are
 * not 16-byte aligned with an even pitch take the kernel's scalar-load form: same results, slower). */
size_t dlsa_newton_wide_workspace_bytes(int64_t n, int p, int intercept);
int dlsa_newton_wide_eligible(const double* X, int64_t ldx, int64_t n, int p, int intercept);
int dlsa_newton_wide_pass_f64(const double* X, int64_t ldx, const double* y, const double* beta, int64_t n, int p, int intercept,
                              double* w_out, double* g, double* loglik, double* H_approx, int64_t ldh, void* ws, size_t ws_bytes,
                              void* stream);
size_t dlsa_gram_icpt_workspace_bytes(int64_t n, int p);
int dlsa_gram_icpt_f64(const double* X, int64_t ldx, const double* w, int64_t n, int p, double* H, int64_t ldh,
                       void* ws, size_t ws_bytes, void* stream);
size_t dlsa_irls_ex_workspace_bytes(int64_t max_rows_per_partition, int p, int intercept, int64_t row_step);
int dlsa_irls_fit_ex_f64(const double* X, int64_t ldx, const double* y, const int64_t* part_first_host,
                         const int64_t* part_rows_host, int64_t row_step, int K, int p, int intercept, double tol, int max_iter,
                         double* coef, double* Sig_inv, double* Sig_invMcoef, int* n_iter_host, int* status_host,
                         double* loglik_host, void* ws, size_t ws_bytes, void* stream);

/* ---- a9: local sum of partition blocks before the one-round all-reduce (dlsa.py:30-34) -
 * out = [ sum_k Sig_inv (p*p) | sum_k Sig_invMcoef (p) | sum_k coef (p) ] contiguous,
 * the message a rank contributes to the RCCL all-reduce.  Blocks whose status is not OK may
 * be skipped via mask_host (nullable, K ints, 0 = include). */
int dlsa_sum_blocks_f64(const double* coef, const double* Sig_inv, const double* Sig_invMcoef,
                        int K, int p, const int* mask_host, double* out, void* stream);

/* ---- a9, the exchange itself: ONE all-reduce (sum) of the rank's message over RCCL / xGMI (dlsa/dlsa.py:30-34: Spark
 * groupby('par_id').sum + toPandas).  For hosts that do not use torch.distributed (whose "nccl" backend is the same RCCL):
 * rank 0 calls dlsa_comm_unique_id and hands the DLSA_COMM_ID_BYTES bytes to the other ranks out of band, every rank calls
 * dlsa_comm_init_rank after selecting its device (hipSetDevice), then dlsa_allreduce_f64(comm, msg, p*p + 2*p, stream)
 * in place, enqueued on `stream`.  RCCL is resolved at run time (dlopen); without it these return DLSA_ERR_HIP. */
#define DLSA_COMM_ID_BYTES 128
int dlsa_comm_unique_id(char* id128);
int dlsa_comm_init_rank(void** comm, int nranks, const char* id128, int rank);
int dlsa_comm_destroy(void* comm);
int dlsa_allreduce_f64(void* rccl_comm, double* buf, int64_t count, void* stream);

/* ---- a10: WLS combine  theta = Sig_inv^{-1} v  (dlsa/dlsa.py:48-49, lstsq on an SPD
 * system) by Cholesky on the device.  S p x p (lds >= p) is not modified. */
size_t dlsa_solve_workspace_bytes(int p);
int dlsa_spd_solve_f64(const double* S, int64_t lds, const double* v, int p, double* theta,
                       void* ws, size_t ws_bytes, void* stream);

/* ---- a10, complete semantics: theta = lstsq(S, v, rcond=None)[0]  (dlsa/dlsa.py:48-49) -------------------------------
 * For an SPD sum this is the Cholesky solve above.  When the sum is singular -- a dummy level that occurs in no partition
 * leaves a zero row / column (models.py:84-91), a collinear dummy set survives -- numpy returns the MINIMUM-NORM least-squares
 * solution with singular values <= eps * p * sigma_max treated as zero.  dlsa_wls_solve_f64 tries the Cholesky solve and
 * falls back to dlsa_sym_pinv_solve_f64 when a pivot fails or is roundoff-sized (L_ii^2 <= 8 eps p S_ii);
 * rank_host (nullable) receives the numerical rank (p for the SPD case).
 * dlsa_sym_pinv_solve_f64: parallel two-sided Jacobi eigendecomposition of the symmetric S (p <= 2048), then
 * theta = V diag(1 / lambda_i : |lambda_i| > rcond * max|lambda|) V' v.  rcond < 0 = eps * p (lstsq's rcond=None).
 * eig_host (nullable, p doubles, host) receives the eigenvalues (unordered). */
size_t dlsa_wls_solve_workspace_bytes(int p);
int dlsa_wls_solve_f64(const double* S, int64_t lds, const double* v, int p, double* theta, int* rank_host,
                       void* ws, size_t ws_bytes, void* stream);
size_t dlsa_sym_pinv_workspace_bytes(int p);
int dlsa_sym_pinv_solve_f64(const double* S, int64_t lds, const double* v, int p, double rcond, double* theta,
                            int* rank_host, double* eig_host, void* ws, size_t ws_bytes, void* stream);

/* ---- a13/a14: LARS path for the least-squares approximation (dlsa/lsa.py:90-212) ------
 * Sigma0 p x p, b0 p (device).  type 0 = 'lar', 1 = 'lasso'.  max_steps <= 0 -> 8*m.
 * Outputs (device): beta_path (max_steps+1) x m row-major (m = p - intercept), beta0,
 * aic, bic (max_steps+1 each); n_steps_host = number of steps taken (path has n_steps+1
 * rows).  Runs as one persistent kernel on the device.  Up to 448 variables the carried-rows
 * form (lars_q.hip): one workgroup up to 200 variables, 4 (beyond 420: 8) workgroups that share
 * the fused pass above; 449 .. 2044 variables the same rows with the pass split by COLUMNS over
 * 16 .. 64 workgroups (lars_c.hip, round 6: one grid barrier per append; p = 2000 in 35 ms);
 * beyond, a grid of up to 32 workgroups on the R^-1 form (lars.hip, two grid barriers
 * per step).  The multi-workgroup kernels need their workgroups resident together: plain
 * launches whose barrier waits are bounded (0.25 s) -- a launch that gives up is rerun on a
 * single workgroup: slower, same path; DLSA_ERR_HIP only if that is impossible -- and whose
 * launches are serialised inside a process; dlsa_kernel_options.cooperative = 1 launches them
 * with hipLaunchCooperativeKernel instead (co-residency or a clean refusal; opt-in because
 * HIP streams created after a process's first cooperative launch serialise on this runtime).
 * p is bounded by the LDS per workgroup: about 2400 columns for the grid kernel, 3300 for
 * the single workgroup it falls back to. */
size_t dlsa_lars_workspace_bytes(int p);
int dlsa_lars_lsa_f64(const double* Sigma0, int64_t lds, const double* b0, int p,
                      int intercept, double n, int type, double eps, int max_steps,
                      double* beta_path, double* beta0, double* aic, double* bic,
                      int* n_steps_host, void* ws, size_t ws_bytes, void* stream);

/* Diagnostics / test hook of the grid kernel's bounded barrier (no reference counterpart: lars_lsa, dlsa/lsa.py:90-212,
 * is host numpy): sets the per-barrier timeout in seconds (<= 0 restores the 0.25 s default) and returns how many grid
 * launches of this process have been given up and rerun on the single-workgroup kernel so far. */
int dlsa_lars_grid_barrier_timeout(double seconds);

/* The same for the one-launch kernel of many small partitions (dlsa_irls_fit*_f64 with K >= 2 partitions of <= 64 columns and
 * <= 65 536 rows; models.py:110-131 per partition): with fewer partitions than CUs several workgroups share a partition and meet
 * at a bounded per-partition barrier; a launch whose barrier times out is rerun with one workgroup per partition.  Sets the
 * timeout in seconds (<= 0 restores the 0.25 s default), returns the number of such reruns of this process so far. */
int dlsa_irls_small_cluster_timeout(double seconds);

/* ---- design matrix (N2) ----
 * Replaces pd.get_dummies + drop(baselines) + standardise + reindex (dlsa/models.py:56-104) and the
 * leading ones column (models.py:121-122) for one chunk whose categorical columns arrive as integer
 * level codes.  Device inputs: num [n x q] raw numeric columns (row-major, ldn), codes [n x f] int32
 * (row-major, ldc).  Output column j < p is described by device arrays kind/src/level/shift/scale:
 *   kind 0: the constant 1;  kind 1: (num[:, src[j]] - shift[j]) / scale[j];
 *   kind 2: 1.0 where codes[:, src[j]] == level[j], else 0.0.
 * seen (device int32[p], nullable) is set to 1 for every column with a non-zero entry: a dummy column
 * with seen == 0 is the reference's "level missing in this data chunk" case (models.py:80-91).
 * p <= 2048. */
int dlsa_design_f64(const double* num, int64_t ldn, int q, const int32_t* codes, int64_t ldc, int f,
                    int64_t n, const int32_t* kind, const int32_t* src, const int32_t* level,
                    const double* shift, const double* scale, int p, double* X, int64_t ldx,
                    int32_t* seen, void* stream);
int dlsa_design_f32(const float* num, int64_t ldn, int q, const int32_t* codes, int64_t ldc, int f,
                    int64_t n, const int32_t* kind, const int32_t* src, const int32_t* level,
                    const double* shift, const double* scale, int p, float* X, int64_t ldx,
                    int32_t* seen, void* stream);

/* ---- structured passes for one-hot designs (N2) ----
 * For a design [intercept | standardised numerics | one-hot factor levels] (the dummy path of logistic_model,
 * dlsa/models.py:56-131) the logit pass is a gather, X'r a histogram and X'WX weighted co-occurrence counts: the
 * passes below read the raw row (q doubles + f int32 codes) instead of the p-column dense row and produce the SAME
 * w / g / loglik / p x p Hessian as dlsa_logit_pass_f64 / dlsa_gram_f64 on the matrix dlsa_design_f64 would build.
 * A plan describes the design: `ndense` <= 8 dense columns (dense_kind 0 = the constant 1, 1 = numeric column
 * dense_src standardised as (x - shift) / scale; dense_col = its output column) and `nfactor` <= 8 factors with
 * nlevels[t] level codes each; level_col (concatenated over the factors) gives the output column of every level,
 * -1 for a level without a column (baseline / dropped).  All descriptor arrays are HOST arrays.  Plan creation
 * fails with DLSA_ERR_INVALID when a factor-pair table does not fit the per-workgroup LDS budget (use the dense
 * path then).  Accumulation: the Hessian inside dlsa_onehot_irls_fit_* (weights mu (1 - mu) <= 1/4 of the library's own logit pass)
 * sums exact 64-bit fixed-point addends of absolute resolution 2^-40 in LDS -- order-independent, bit-reproducible; the public
 * dlsa_onehot_gram_f64, whose caller may bring weights of ANY scale, sums in ordered floating point (full fp64 relative accuracy
 * whatever the scale of w, also bit-reproducible, about 1.8x the time).  DLSA_OH_ORDERED=0 / 1 forces unordered / ordered adds. */
typedef struct dlsa_onehot_plan dlsa_onehot_plan;
int dlsa_onehot_plan_create(int p, int ndense, const int32_t* dense_kind, const int32_t* dense_src,
                            const double* dense_shift, const double* dense_scale, const int32_t* dense_col,
                            int nfactor, const int32_t* nlevels, const int32_t* level_col, dlsa_onehot_plan** out);
void dlsa_onehot_plan_destroy(dlsa_onehot_plan* plan);
int dlsa_onehot_plan_roles(const dlsa_onehot_plan* plan);      /* workgroup roles of the Gram (each streams all rows) */
size_t dlsa_onehot_workspace_bytes(const dlsa_onehot_plan* plan, int64_t n);
int dlsa_onehot_logit_pass_f64(const dlsa_onehot_plan* plan, const double* num, int64_t ldn, const int32_t* codes,
                               int64_t ldc, const double* y, const double* beta, int64_t n, double* w_out, double* g,
                               double* loglik, void* ws, size_t ws_bytes, void* stream);
int dlsa_onehot_gram_f64(const dlsa_onehot_plan* plan, const double* num, int64_t ldn, const int32_t* codes,
                         int64_t ldc, const double* w, int64_t n, double* H, int64_t ldh, void* ws, size_t ws_bytes,
                         void* stream);
/* dlsa_irls_fit_f64 on the raw representation (same outputs, statuses and acceleration policy). */
size_t dlsa_onehot_irls_workspace_bytes(const dlsa_onehot_plan* plan, int64_t max_rows_per_partition);
int dlsa_onehot_irls_fit_f64(const dlsa_onehot_plan* plan, const double* num, int64_t ldn, const int32_t* codes,
                             int64_t ldc, const double* y, const int64_t* part_offsets_host, int K, double tol,
                             int max_iter, double* coef, double* Sig_inv, double* Sig_invMcoef, int* n_iter_host,
                             int* status_host, double* loglik_host, void* ws, size_t ws_bytes, void* stream);

/* N3, streaming form: ONE read of X gives, in fp64 whatever the rows' type, g = X'v (p), colsum = X'1 (p, nullable: the
 * intercept's border of [1 | X]'[1 | X] with the ones column of models.py:121-122 left implicit), stats[0] = v'v and
 * stats[1] = sum v; accumulate != 0 ADDS to what the outputs hold (the chunks before this one). */
size_t dlsa_xtv_stats_workspace_bytes(int p, int elem_bytes);
int dlsa_xtv_stats_f64(const double* X, int64_t ldx, const double* v, int64_t n, int p, double* g, double* colsum,
                       double* stats, int accumulate, void* ws, size_t ws_bytes, void* stream);
int dlsa_xtv_stats_f32(const float* X, int64_t ldx, const float* v, int64_t n, int p, double* g, double* colsum,
                       double* stats, int accumulate, void* ws, size_t ws_bytes, void* stream);

/* dlsa_onehot_irls_fit_f64 for partitions given as (first row, rows, common row step) -- partition_id = i % K (models.py:33)
 * as strided views of the raw numerics and level codes: nothing is gathered but each partition's labels (8 B per row). */
size_t dlsa_onehot_irls_ex_workspace_bytes(const dlsa_onehot_plan* plan, int64_t max_rows_per_partition, int64_t row_step);
int dlsa_onehot_irls_fit_ex_f64(const dlsa_onehot_plan* plan, const double* num, int64_t ldn, const int32_t* codes,
                                int64_t ldc, const double* y, const int64_t* part_first_host, const int64_t* part_rows_host,
                                int64_t row_step, int K, double tol, int max_iter, double* coef, double* Sig_inv,
                                double* Sig_invMcoef, int* n_iter_host, int* status_host, double* loglik_host,
                                void* ws, size_t ws_bytes, void* stream);

/* test hook: host-only validation of the Gram tile plan for p (0 = every tile on/above the diagonal
 * is stored exactly once; outputs: workgroup items, tile slots computed, tiles stored). */
int dlsa_gram_plan_check(int p, int* nitems, int* nslots, int* ntiles);
/* same for the 256-column-panel plan of the wide-p fp32 kernel (used by dlsa_gram_f32 when p >= 768,
 * p % 4 == 0, 16-byte aligned rows). */
int dlsa_gram_wide_plan_check(int p, int* nitems, int* nslots, int* ntiles);

#ifdef __cplusplus
}
#endif
#endif /* DLSA_HIP_H */
