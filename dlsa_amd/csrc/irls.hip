// Per-partition exact-MLE logistic fit + local quadratic approximation: the numeric core of
// logistic_model (dlsa/models.py:110-131) for K row-partitions that sit contiguously in HBM.
//
// Per partition the unpenalised MLE is found by Newton/IRLS on the device.  One Newton iteration =
// fused logit pass (w, g, loglik; HBM-bound) + Gram pass H = X'WX (MFMA-bound, ~6x the logit pass
// at p=500) + Cholesky solve H delta = g.  Two measures cut the number of Gram passes without
// changing the answer (the stopping rule is always evaluated on the FULL partition):
//   * warm start: large partitions are first solved on their leading 1/16 of the rows (every pass
//     16x cheaper); the full-data iterations then start O(n^-1/2) away from the MLE;
//   * partition-to-partition warm start: partition k+1 starts from partition k's MLE (same theta);
//   * frozen Hessian: once |delta|_inf <= 1e-1*max(1,|beta|_inf) the factor of the last Hessian is
//     reused and only logit passes + triangular solves run (linear convergence at a rate
//     ~|beta - beta*|; a stalled frozen iteration (step shrinking by < 4x) refreshes the Hessian).
//   * inherited factor + secant correction: for p >= 192 the full-data iterations start from the factor of a stand-in
//     Hessian (a quarter of the rows at the subsample MLE, or the previous partition's) and every reused factor is
//     corrected with the L-BFGS two-loop recursion on the exact curvature pairs the iterations produce.
// Stop when |delta|_inf <= tol*max(1,|beta|_inf).  The Hessian written to Sig_inv[k] is ALWAYS a
// fresh Gram pass at the returned coef, exactly as the reference evaluates it after the fit
// (models.py:114,130); Sig_invMcoef = Sig_inv . coef (models.py:131).  One D2H copy of 4 doubles
// per iteration is the only host synchronisation.
#include "common.h"
#include "options.h"
#include <math.h>
#include <algorithm>
#include <stdlib.h>
#include <functional>
#include <vector>
#include <map>
#include <mutex>
#include <string>
#include <thread>

namespace dlsa {
int gram_impl_f64(const double* X, int64_t ldx, const double* w, int64_t n, int p, double* H, int64_t ldh,
                  int accumulate, void* ws, size_t ws_bytes, hipStream_t stream);
size_t gram_workspace_bytes_impl(int64_t n, int p, int elem_bytes);
// irls_pass.hip: one Newton pass in one launch where the shape allows it (narrow designs: the rows staged for the MFMAs also
// feed the logistic terms -- one read of X per fresh Hessian instead of two)
bool irls_pass_fused_eligible(const double* X, int64_t ldx, const double* y, int64_t n, int p);
size_t irls_pass_workspace_bytes_impl(int64_t n, int p);
bool irls_pass_fused_icpt_eligible(const double* X, int64_t ldx, const double* y, int64_t n, int p);
int irls_pass_icpt_impl(const double* X, int64_t ldx, const double* y, const double* beta, int64_t n, int p, double* H, int64_t ldh,
                        double* g, double* loglik, double* w_out, void* ws, size_t ws_bytes, hipStream_t stream);
int irls_pass_impl(const double* X, int64_t ldx, const double* y, const double* beta, int64_t n, int p, double* H, int64_t ldh,
                   double* g, double* loglik, double* w_out, double* w_scratch, void* ws, size_t ws_bytes, hipStream_t stream,
                   int* fused_out);
size_t logit_workspace_bytes_impl(int64_t n, int p);
int logit_pass_impl(const double* X, int64_t ldx, const double* y, const double* beta, int64_t n, int p,
                    double* w_out, double* g, double* loglik, void* ws, size_t ws_bytes, hipStream_t stream, int intercept);
bool logit_border_ok(const double* X, int64_t ldx, int p);
int logit_pass_border_impl(const double* X, int64_t ldx, const double* y, const double* beta, int64_t n, int p,
                           double* w_out, double* g, double* loglik, double* border, void* ws, size_t ws_bytes, hipStream_t stream);
int xtv_impl(const double* X, int64_t ldx, const double* v, int64_t n, int p, double* g, double* vv, double* sv,
             void* ws, size_t ws_bytes, hipStream_t s);
// irls_small.hip: all partitions in ONE launch, a workgroup each (many small partitions)
bool irls_small_eligible(const int64_t* rows_host, int K, int pe, double* est_ms = nullptr);
size_t irls_small_workspace_bytes(int K);
int irls_small_fit(const double* X, int64_t ldx, const double* y, const int64_t* first_host, const int64_t* rows_host,
                   int64_t step, int K, int p, int intercept, double tol, int max_iter, double* coef, double* Sig_inv,
                   double* Sig_invMcoef, int* n_iter_host, int* status_host, double* loglik_host, void* ws, size_t ws_bytes,
                   hipStream_t s);
int launch_chol_solve(const double* A, int64_t lda, int64_t strideA, const double* rhs, int64_t stride_rhs,
                      const double* ref, int64_t stride_ref, int p, int nsys, double* Lws, double* xout,
                      int64_t stride_x, double* stats, int64_t stride_stats, hipStream_t s, int reuse_factor);
int launch_matvec(const double* A, int64_t lda, const double* x, int p, double* y, hipStream_t s);
int launch_tri_inverse(const double* L, int p, double* Linv, hipStream_t s);
bool chol_small_ok(int p);
int launch_chol_small(const double* A, int64_t lda, int p, const double* rhs, const double* ref, double* Hinv, double* xout,
                      double* stats, hipStream_t s);
int launch_inv_apply(const double* Linv, int p, const double* rhs, const double* ref, double* xout, double* stats, hipStream_t s);
}  // namespace dlsa
struct dlsa_onehot_plan;
namespace dlsa {
size_t onehot_workspace_bytes_impl(const dlsa_onehot_plan* pl, int64_t n);
int onehot_plan_p(const dlsa_onehot_plan* pl);
int onehot_logit_pass_impl(const dlsa_onehot_plan* pl, const double* num, int64_t ldn, const int32_t* codes, int64_t ldc,
                           const double* y, const double* beta, int64_t n, double* w_out, double* g, double* loglik,
                           void* ws, size_t ws_bytes, hipStream_t s);
int onehot_gram_impl(const dlsa_onehot_plan* pl, const double* num, int64_t ldn, const int32_t* codes, int64_t ldc,
                     const double* w, int64_t n, double* H, int64_t ldh, void* ws, size_t ws_bytes, hipStream_t s, bool irls_weights);
int launch_axpby(const double* a, const double* b, double sc, int n, double* out, hipStream_t s);
int launch_advance(double* prev, double* beta, const double* delta, int n, hipStream_t s);
int launch_matvec_axpy(const double* A, int64_t lda, const double* x, int p, double alpha, const double* z, double beta, double* y, hipStream_t s);
int launch_step_stats(const double* delta, const double* ref, int p, double* stats, hipStream_t s);
// irls_wide.hip: the logit pass of a wide design that also yields the partition's own Hessian in reduced precision (bf16 products)
bool irls_wide_eligible(const double* X, int64_t ldx, int64_t n, int p, int icpt);
size_t irls_wide_workspace_bytes(int64_t n, int p, int icpt);
int irls_wide_pass_impl(const double* X, int64_t ldx, const double* y, const double* beta, int64_t n, int p, int icpt, double* w_out,
                        double* g, double* loglik, double* Happrox, int64_t ldh, void* ws, size_t ws_bytes, hipStream_t stream);
constexpr int64_t kWideMaxRows = 4000000;  // (the pass keeps a bf16 image of the partition, 1 KiB per row at p = 500.  A single 2.5e7-row partition would be served too -- measured: 7 -> 5 full passes, 0.2437 -> 0.2398 s -- for 25 GB more workspace: not taken)
// irls_batch.hip: the lock-step fit of all partitions of a call together
bool irls_batched_eligible(const double* X, int64_t ldx, const double* y, const int64_t* rows_host, int K, int p, int intercept, int64_t row_step,
                           double* est_ms = nullptr);
int irls_batched_fit(const double* X, int64_t ldx, const double* y, const int64_t* first_host, const int64_t* rows_host, int64_t row_step, int K,
                     int p, int intercept, double tol, int max_iter, double* coef, double* Sig_inv, double* Sig_invMcoef, int* n_iter_host, int* status_host,
                     double* loglik_host, hipStream_t stream);
// which driver the calling thread's last fit took: 0 = host-driven partition chains, 1 = the one-launch kernel for small partitions,
// 2 = lock step (dlsa_irls_last_fit_path)
static thread_local int g_last_fit_path = 0;

constexpr int QN_PAIRS = 6;       // secant pairs kept for the quasi-Newton correction

struct IrlsLayout {
    size_t w, g, beta, beta_prev, delta, stats, L, Linv, Hinv, Hpool, Ha, rr, pass, pass_bytes, qn_s, qn_y, qn_rho, qn_alpha, qn_q, qn_gprev, border, total;
};

// pass_bytes: scratch of the data source's logit / Gram passes (they never run concurrently)
static IrlsLayout irls_layout(int64_t max_rows, int p, size_t pass_bytes) {
    IrlsLayout l;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = align_up(off, 256); off = o + bytes; return o; };
    l.w = take((size_t)std::max<int64_t>(max_rows, 1) * sizeof(double));
    l.g = take((size_t)p * sizeof(double));
    l.beta = take((size_t)p * sizeof(double));
    l.beta_prev = take((size_t)p * sizeof(double));
    l.delta = take((size_t)p * sizeof(double));
    l.stats = take(8 * sizeof(double));
    l.L = take((size_t)p * p * sizeof(double));
    l.Linv = take((size_t)p * p * sizeof(double));
    l.Hinv = take((size_t)p * p * sizeof(double));
    l.Hpool = take((size_t)p * p * sizeof(double));
    l.Ha = (p >= 121 && p <= 512) ? take((size_t)p * p * sizeof(double)) : 0;      // the partition's own reduced-precision Hessian (wide designs; offset 0 = none: w sits there)
    l.rr = take((size_t)p * sizeof(double));
    // ... and of the p x p Gram pass that forms the explicit inverse of a reused factor
    l.pass_bytes = std::max(pass_bytes, gram_workspace_bytes_impl(p, p, 8));
    l.pass = take(l.pass_bytes);
    l.qn_s = take((size_t)QN_PAIRS * p * sizeof(double));
    l.qn_y = take((size_t)QN_PAIRS * p * sizeof(double));
    l.qn_rho = take(QN_PAIRS * sizeof(double));
    l.qn_alpha = take(QN_PAIRS * sizeof(double));
    l.qn_q = take((size_t)p * sizeof(double));
    l.qn_gprev = take((size_t)p * sizeof(double));
    l.border = take((size_t)p * sizeof(double));          // [sum w | X'w] left by a logit pass with the implicit intercept (p = p_data + 1 there)
    l.total = align_up(off, 256);
    return l;
}

// ---------------------------------------------------------------------------------------------
// Secant (L-BFGS) correction of the frozen / inherited Hessian.  While the Cholesky factor of an approximate
// Hessian H0 is reused, the iteration delta = H0^-1 g contracts only linearly (rate ~ |I - H0^-1 H|: 0.04-0.08
// with a subsample Hessian at p = 500, i.e. nine to ten full logit passes to 1e-13).  Every iteration also yields
// an exact curvature pair (s = beta_k - beta_k-1, y = g_k-1 - g_k = H s), and the two-loop recursion around the
// triangular solve turns the contraction superlinear for a few tiny single-workgroup kernels per iteration.
// The fixed point is unchanged (g = 0): only the path to the MLE gets shorter.
// ---------------------------------------------------------------------------------------------
struct QnOrder { int idx[QN_PAIRS]; int m; };     // ring slots, oldest first

__device__ __forceinline__ double block_sum(double v, double* red) {
    v = wave_allreduce_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double s = 0.0;
    for (int k = 0; k < (int)(blockDim.x >> 6); ++k) s += red[k];       // fixed order
    return s;
}

__global__ __launch_bounds__(1024) void qn_push_kernel(const double* __restrict__ beta, const double* __restrict__ prev,
                                                       const double* __restrict__ gprev, const double* __restrict__ g, int p,
                                                       double* __restrict__ S, double* __restrict__ Y, double* __restrict__ rho) {
    __shared__ double red[16];
    double sy = 0.0;
    for (int i = threadIdx.x; i < p; i += blockDim.x) {
        const double sv = beta[i] - prev[i], yv = gprev[i] - g[i];
        S[i] = sv; Y[i] = yv;
        sy = fma(sv, yv, sy);
    }
    sy = block_sum(sy, red);
    if (threadIdx.x == 0) *rho = (sy > 0.0 && isfinite(sy)) ? 1.0 / sy : 0.0;      // rho = 0: pair ignored
}

// first loop: q = (g - sum alpha_i y_i) * gscale, newest pair first
__global__ __launch_bounds__(1024) void qn_pre_kernel(const double* __restrict__ g, const double* __restrict__ S,
                                                      const double* __restrict__ Y, const double* __restrict__ rho,
                                                      QnOrder ord, int p, double gscale, double* __restrict__ q,
                                                      double* __restrict__ alpha) {
    extern __shared__ double sm[];
    double* qv = sm;                 // p
    double* red = sm + p;            // 16
    for (int i = threadIdx.x; i < p; i += blockDim.x) qv[i] = g[i];
    __syncthreads();
    for (int k = ord.m - 1; k >= 0; --k) {
        const int slot = ord.idx[k];
        const double* s = S + (int64_t)slot * p;
        const double* y = Y + (int64_t)slot * p;
        double d = 0.0;
        for (int i = threadIdx.x; i < p; i += blockDim.x) d = fma(s[i], qv[i], d);
        const double a = rho[slot] * block_sum(d, red);
        if (threadIdx.x == 0) alpha[slot] = a;
        for (int i = threadIdx.x; i < p; i += blockDim.x) qv[i] = fma(-a, y[i], qv[i]);
        __syncthreads();
    }
    for (int i = threadIdx.x; i < p; i += blockDim.x) q[i] = qv[i] * gscale;
}

// second loop on r = H0^-1 q (in place), oldest pair first; then stats[0] = |r|_inf
__global__ __launch_bounds__(1024) void qn_post_kernel(double* __restrict__ r, const double* __restrict__ S,
                                                       const double* __restrict__ Y, const double* __restrict__ rho,
                                                       const double* __restrict__ alpha, QnOrder ord, int p,
                                                       double* __restrict__ stats) {
    extern __shared__ double sm[];
    double* rv = sm;
    double* red = sm + p;
    for (int i = threadIdx.x; i < p; i += blockDim.x) rv[i] = r[i];
    __syncthreads();
    for (int k = 0; k < ord.m; ++k) {
        const int slot = ord.idx[k];
        const double* s = S + (int64_t)slot * p;
        const double* y = Y + (int64_t)slot * p;
        double d = 0.0;
        for (int i = threadIdx.x; i < p; i += blockDim.x) d = fma(y[i], rv[i], d);
        const double bcoef = rho[slot] * block_sum(d, red);
        const double c = alpha[slot] - bcoef;
        for (int i = threadIdx.x; i < p; i += blockDim.x) rv[i] = fma(c, s[i], rv[i]);
        __syncthreads();
    }
    double mx = 0.0;
    for (int i = threadIdx.x; i < p; i += blockDim.x) { r[i] = rv[i]; mx = fmax(mx, fabs(rv[i])); }
    for (int m = 32; m >= 1; m >>= 1) mx = fmax(mx, __shfl_xor(mx, m, 64));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = 0.0;
        for (int k = 0; k < (int)(blockDim.x >> 6); ++k) a = fmax(a, red[k]);
        stats[0] = a;
    }
}

// One launch per quasi-Newton iteration: [push the new curvature pair] -> first loop -> x = H0^-1 q (explicit inverse) -> second loop
// -> delta and stats, all in one 1024-thread workgroup (the separate kernels above cost five launches and their gaps
// per iteration, which is what small partitions are made of).  A thread owns EPT elements of every vector: q and the
// (up to six) curvature pairs sit in registers, fetched with one round of independent loads, so the two loops are
// register arithmetic + one block reduction per pair; only the mat-vec reads q through LDS.
template <int EPT>
__global__ __launch_bounds__(1024) void qn_step_kernel(const double* __restrict__ beta, const double* __restrict__ prev,
                                                       double* __restrict__ gprev, const double* __restrict__ g,
                                                       double* __restrict__ S, double* __restrict__ Y, double* __restrict__ rho,
                                                       QnOrder ord, int push_slot, int p, double gscale,
                                                       const double* __restrict__ Hinv, double* __restrict__ delta,
                                                       double* __restrict__ stats) {
    extern __shared__ double sm[];
    double* qv = sm;                 // p
    double* part = sm + p;           // p
    double* red = part + p;          // 48
    const int tid = threadIdx.x, nth = blockDim.x, lane = tid & 63, wave = tid >> 6, nw = nth >> 6;
    double q[EPT], sv[QN_PAIRS][EPT], yv[QN_PAIRS][EPT], rh[QN_PAIRS], alpha[QN_PAIRS];
    // pairs by age (position k = ring slot ord.idx[k]); the pair pushed now is the newest one
#pragma unroll
    for (int k = 0; k < QN_PAIRS; ++k) {
        const bool have = k < ord.m && !(push_slot >= 0 && k == ord.m - 1);
        const int slot = ord.idx[k < ord.m ? k : 0];
        rh[k] = have ? rho[slot] : 0.0;
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            const int i = tid + e * nth;
            sv[k][e] = (have && i < p) ? S[(int64_t)slot * p + i] : 0.0;
            yv[k][e] = (have && i < p) ? Y[(int64_t)slot * p + i] : 0.0;
        }
    }
    double sy = 0.0;
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const int i = tid + e * nth;
        const double gi = i < p ? g[i] : 0.0;
        q[e] = gi;
        if (push_slot >= 0 && i < p) {
            const double s1 = beta[i] - prev[i], y1 = gprev[i] - gi;
            S[(int64_t)push_slot * p + i] = s1;
            Y[(int64_t)push_slot * p + i] = y1;
            sy = fma(s1, y1, sy);
#pragma unroll
            for (int k = 0; k < QN_PAIRS; ++k)
                if (k == ord.m - 1) { sv[k][e] = s1; yv[k][e] = y1; }
        }
        if (i < p) gprev[i] = gi;
    }
    if (push_slot >= 0) {
        sy = block_sum(sy, red);
        const double r1 = (sy > 0.0 && isfinite(sy)) ? 1.0 / sy : 0.0;
        if (tid == 0) rho[push_slot] = r1;
#pragma unroll
        for (int k = 0; k < QN_PAIRS; ++k)
            if (k == ord.m - 1) rh[k] = r1;
        __syncthreads();
    }
#pragma unroll
    for (int k = QN_PAIRS - 1; k >= 0; --k) {
        alpha[k] = 0.0;
        if (k < ord.m) {
            double d = 0.0;
#pragma unroll
            for (int e = 0; e < EPT; ++e) d = fma(sv[k][e], q[e], d);
            const double a = rh[k] * block_sum(d, red);
            alpha[k] = a;
#pragma unroll
            for (int e = 0; e < EPT; ++e) q[e] = fma(-a, yv[k][e], q[e]);
            __syncthreads();
        }
    }
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const int i = tid + e * nth;
        if (i < p) qv[i] = q[e] * gscale;
    }
    __syncthreads();
    // r = H0^-1 q with the explicit inverse: a wave takes four rows at a time and issues their 32 loads per lane (512
    // columns) before the first multiply -- clamped addresses, zero multipliers, no branches around the loads: this
    // phase is one CU pulling p^2 doubles from L2, so what counts is loads in flight
    for (int i0 = 4 * wave; i0 < p; i0 += 4 * nw) {
        double acc4[4] = {0.0, 0.0, 0.0, 0.0};
        for (int k0 = 0; k0 < p; k0 += 512) {
            double hv[4][8];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double* row = Hinv + (int64_t)min(i0 + r, p - 1) * p;
#pragma unroll
                for (int c = 0; c < 8; ++c) hv[r][c] = row[min(k0 + lane + 64 * c, p - 1)];
            }
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const int k = k0 + lane + 64 * c;
                const double x = k < p ? qv[min(k, p - 1)] : 0.0;
#pragma unroll
                for (int r = 0; r < 4; ++r) acc4[r] = fma(hv[r][c], x, acc4[r]);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const double t = wave_allreduce_sum(acc4[r]);
            if (lane == 0 && i0 + r < p) part[i0 + r] = t;
        }
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const int i = tid + e * nth;
        q[e] = i < p ? part[i] : 0.0;
    }
#pragma unroll
    for (int k = 0; k < QN_PAIRS; ++k) {
        if (k < ord.m) {
            double d = 0.0;
#pragma unroll
            for (int e = 0; e < EPT; ++e) d = fma(yv[k][e], q[e], d);
            const double c = alpha[k] - rh[k] * block_sum(d, red);
#pragma unroll
            for (int e = 0; e < EPT; ++e) q[e] = fma(c, sv[k][e], q[e]);
            __syncthreads();
        }
    }
    double mx = 0.0, mr = 0.0;
    int bad = 0;
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const int i = tid + e * nth;
        if (i < p) {
            const double v = q[e];
            delta[i] = v;
            mx = fmax(mx, fabs(v));
            if (!isfinite(v)) bad = 1;
            mr = fmax(mr, fabs(beta[i]));
        }
    }
    for (int m = 32; m >= 1; m >>= 1) {
        mx = fmax(mx, __shfl_xor(mx, m, 64));
        mr = fmax(mr, __shfl_xor(mr, m, 64));
        bad |= __shfl_xor(bad, m, 64);
    }
    __syncthreads();
    if (lane == 0) { red[wave] = mx; red[16 + wave] = mr; red[32 + wave] = (double)bad; }
    __syncthreads();
    if (tid == 0) {
        double a = 0.0, b = 0.0, c = 0.0;
        for (int k = 0; k < nw; ++k) { a = fmax(a, red[k]); b = fmax(b, red[16 + k]); c = fmax(c, red[32 + k]); }
        stats[0] = a;
        stats[1] = b;
        if (c != 0.0 && stats[2] == 0.0) stats[2] = 2.0;
    }
}

struct IrlsBuffers {
    double *w, *g, *beta, *prev, *delta, *stats, *L, *Linv, *Hinv, *Hpool;
    double *Ha = nullptr, *rr = nullptr;       // own reduced-precision Hessian (null: not served) and the residual of its inner solve
    int* inv_valid;      // host flag: Linv is the inverse of the factor currently in L (bit 0), Hinv = Linv' Linv (bit 1)
    double* border;      // the Hessian's intercept border [sum w | X'w] of the LAST logit pass (fits with the implicit intercept)
    int64_t* border_rows;   // host: the row count that pass ran over (-1: none): a Gram over the same rows and weights takes it
    double *qn_s, *qn_y, *qn_rho, *qn_alpha, *qn_q, *qn_gprev;
    void* ws_pass; size_t ws_pass_bytes;
};

// The rows of one partition, whatever their representation (dense matrix, or raw numerics + level codes): the two
// passes of a Newton iteration over the leading `nrows` rows.
struct IrlsData {
    std::function<int(const double* beta, int64_t nrows, double* w, double* g, double* ll, const IrlsBuffers& b, hipStream_t s)> logit;
    std::function<int(const double* w, int64_t nrows, double* H, const IrlsBuffers& b, hipStream_t s)> gram;
    // optional: both passes of a fresh-Hessian iteration in ONE launch (w, g, ll as `logit`, H as `gram`); `fusable(nrows)`
    // says whether that launch really is the fused kernel for this many leading rows (else logit + gram are called)
    std::function<int(const double* beta, int64_t nrows, double* w, double* g, double* ll, double* H, const IrlsBuffers& b, hipStream_t s)> pass;
    std::function<bool(int64_t nrows)> fusable;
    // the fused launch needs no weight vector written (w == nullptr is allowed in `pass` and `logit`): at a fusable size every
    // Hessian then comes from `pass`, and the 8 bytes per row of every pass stay unwritten
    bool lean_w = false;
    // optional (wide designs, irls_wide.hip): the logit pass that ALSO yields the partition's own Hessian in reduced precision -- a
    // preconditioner for the steps, never a result
    std::function<int(const double* beta, int64_t nrows, double* w, double* g, double* ll, double* Happrox, const IrlsBuffers& b, hipStream_t s)> approx;
    std::function<bool(int64_t nrows, size_t ws_pass_bytes)> approxable;      // (the wide pass's scratch must fit the pass workspace this call was sized for)
};

// Newton iterations on rows [0, n) of (X, y) starting from the beta already in b.beta.
// `H` receives every fresh Hessian.  On return with DLSA_PART_OK, *fresh says whether H was evaluated
// at the final beta.  Returns a HIP/argument error code (0 = fine) and sets *status.
// Did the Cholesky factorization that launch_chol_solve just enqueued succeed?  (stats[2]: 1 = non-positive pivot, 2 = NaN.)
// One 8-byte read-back; used where a failed factor would otherwise only surface as NaN iterates.
// The per-iteration read-back (four doubles) lands in PINNED host memory, one slot per host thread (a partition chain is a thread):
// a copy into pageable memory goes through the runtime's staging path, ~30 us more per iteration -- and a config-3 fit makes ~130 of them.
// Slots are pooled for the PROCESS: a chain's thread lives for one fit call, so a thread_local pointer would allocate (and leak) a
// pinned page per chain and call.  A thread borrows a slot for its lifetime and hands it back when it exits.
struct ReadbackSlot {
    double* p = nullptr;
    static std::mutex& mu() { static std::mutex* m = new std::mutex; return *m; }            // (never destroyed: threads may exit during process teardown)
    static std::vector<double*>& idle() { static std::vector<double*>* v = new std::vector<double*>; return *v; }
    ReadbackSlot() {
        {
            std::lock_guard<std::mutex> g(mu());
            if (!idle().empty()) { p = idle().back(); idle().pop_back(); }
        }
        if (!p && hipHostMalloc((void**)&p, 64, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); p = nullptr; }
    }
    ~ReadbackSlot() {
        if (!p) return;
        std::lock_guard<std::mutex> g(mu());
        idle().push_back(p);
    }
};
static double* readback_slot() {
    static thread_local ReadbackSlot slot;
    return slot.p;          // (nullptr: the caller falls back to its stack buffer)
}

static int factor_ok(const IrlsBuffers& b, hipStream_t s, bool* ok) {
    double stack_flag = 0.0;
    double* flag = readback_slot();
    if (!flag) flag = &stack_flag;
    DLSA_HIP_CHECK(hipMemcpyAsync(flag, b.stats + 2, sizeof(double), hipMemcpyDeviceToHost, s));
    DLSA_HIP_CHECK(hipStreamSynchronize(s));
    *ok = (*flag == 0.0);
    return DLSA_OK;
}

static bool inv_enabled(int p) {
    const char* e = knob("DLSA_IRLS_INVERSE");
    return (e ? atoi(e) != 0 : true) && ((size_t)4 * p + 48) * sizeof(double) <= 64 * 1024;
}

static bool qn_enabled() {
    const char* e = knob("DLSA_IRLS_SECANT");
    return e ? atoi(e) != 0 : true;
}

// inherit_scale > 0: b.L already holds the Cholesky factor of a Hessian H0 with  H(beta) ~= inherit_scale * H0
// (the same model on a row subsample, or on the previous partition): the iterations start as quasi-Newton steps
// delta = H0^-1 (g / inherit_scale) -- logit passes and triangular solves only -- and a fresh Gram pass is taken
// only if those steps stop contracting.  The fixed point is the same MLE; only the path changes.
static int newton_run(const IrlsData& d, int64_t n, int p, double tol, int max_iter,
                      double freeze_at, double* H, const IrlsBuffers& b, hipStream_t s, int* status, int* iters,
                      int* gram_passes, double* loglik, bool* fresh, double inherit_scale = 0.0, bool lean = false) {
    double ll_prev = -INFINITY, ll = 0.0, dprev = INFINITY, dprev2 = INFINITY;
    const char* env_pred = knob("DLSA_IRLS_PREDICT");
    // prediction leaves Sig_inv / loglik evaluated up to 10 tol away from the returned coef (see below): only when that
    // is far below the 1e-10 parity tolerance, i.e. never for a loose caller-supplied tol
    const bool predict_on = (env_pred ? atoi(env_pred) != 0 : true) && tol <= 1e-10;
    bool have_prev = false, need_H = !(inherit_scale > 0.0), have_factor = inherit_scale > 0.0;
    double gscale = inherit_scale > 0.0 ? 1.0 / inherit_scale : 1.0;
    int halvings = 0;
    // secant pairs (quasi-Newton correction of the reused factor): ring of QN_PAIRS slots
    const bool qn_on = qn_enabled();
    QnOrder ord; ord.m = 0;
    int qn_next = 0;
    bool qn_have_gprev = false;      // b.qn_gprev holds the gradient at b.prev
    const size_t qn_shm = ((size_t)p + 16) * sizeof(double);
    *status = DLSA_PART_NOT_CONVERGED;
    *fresh = false;
    const char* env_tr = knob("DLSA_IRLS_TRACE");
    const bool trace_on = env_tr && atoi(env_tr) != 0;           // (0 = off, as the header documents)
    const char* env_fl = knob("DLSA_IRLS_FUSE_LAST");
    const bool can_fuse = d.pass && d.fusable && d.fusable(n);
    const bool fuse_last = can_fuse && (env_fl ? atoi(env_fl) != 0 : true);
    // lean: no weights are written -- every Hessian of this run comes from the fused pass, the closing one too (irls_fit_core)
    double* const wout = (lean && d.lean_w && can_fuse) ? nullptr : b.w;
    bool peek_next = false;
    // Own reduced-precision Hessian (wide designs): the steps of a reused factor contract at the rate of that factor's sampling error
    // against THIS partition (~0.05 with the pooled factor at 1e6 x 500: nine to ten passes).  After the first such step the pass is
    // taken in the form that also yields H~ = the partition's own Hessian from bf16 products (irls_wide.hip), and every later step
    // solves H~ delta = g -- by iterative refinement on the explicit inverse of the factor in hand (||I - H0^-1 H~|| ~ 0.05: three
    // sweeps of two mat-vecs, no factorisation).  H~ at an iterate e away from the MLE contracts by ~2e + 1.5e-4 per pass; it is
    // refreshed when that is not enough.  The gradient, the stopping rule and the safeguard stay fp64: the fixed point is the MLE.
    const char* env_own = knob("DLSA_IRLS_OWN_HESSIAN");
    // (only once the iterate is near the MLE -- the last step at most 0.2 max(1, |beta|): far from it, e.g. the first iterations from
    // beta = 0, the weights move so much between iterates that refinement on the previous factor's inverse does not converge, and
    // exact Hessians are the right tool)
    const bool can_approx = (env_own ? atoi(env_own) != 0 : true) && d.approx && d.approxable && b.Ha && d.approxable(n, b.ws_pass_bytes) && inv_enabled(p);
    double near_scale = 1.0;          // max(1, |beta|) of the last completed iteration
    bool have_Ha = false, approx_ok = true, want_refresh = false;
    constexpr int kRefine = 3;
    auto ensure_inverse = [&]() -> int {
        if (!(*b.inv_valid & 1)) {
            int rc = launch_tri_inverse(b.L, p, b.Linv, s);
            if (rc) return rc;
            *b.inv_valid = 1;
        }
        if (!(*b.inv_valid & 2)) {
            int rc = gram_impl_f64(b.Linv, p, nullptr, p, p, b.Hinv, p, 0, b.ws_pass, b.ws_pass_bytes, s);
            if (rc) return rc;
            *b.inv_valid |= 2;
        }
        return DLSA_OK;
    };
    for (int it = 1; it <= max_iter; ++it) {
        ++*iters;
        const bool fresh_now = need_H || !have_factor;
        // peek: the iteration expected to end the run takes the fused pass for the RESULT's Hessian only -- the step still comes
        // from the factor in hand (no refactorisation, the secant pairs stay)
        const bool peek = !fresh_now && peek_next && can_fuse;
        peek_next = false;
        int rc;
        if ((fresh_now || peek) && can_fuse) {
            rc = d.pass(b.beta, n, wout, b.g, b.stats + 3, H, b, s);     // w, g, loglik and H in one read of the rows
            if (rc) return rc;
        } else if (can_approx && approx_ok && !fresh_now && !peek && isfinite(dprev) && dprev <= 0.2 * near_scale &&
                   ((!have_Ha && it >= 2) || (have_Ha && want_refresh))) {
            rc = d.approx(b.beta, n, b.w, b.g, b.stats + 3, b.Ha, b, s);    // w, g, loglik as the logit pass + the partition's own H~
            if (rc) return rc;
            have_Ha = true;
            want_refresh = false;
        } else {
            rc = d.logit(b.beta, n, fresh_now ? b.w : wout, b.g, b.stats + 3, b, s);
            if (rc) return rc;
            if (fresh_now) {
                rc = d.gram(b.w, n, H, b, s);
                if (rc) return rc;
            }
        }
        if (fresh_now) {
            ++*gram_passes;
            gscale = 1.0;
            ord.m = 0;                               // a new H0: the old pairs go
            have_Ha = false;                         // (... and the exact factor serves from here)
        }
        if (have_Ha) {
            // delta = H~^-1 g by refinement on M = gscale * H0^-1:  delta = M g;  delta += M (g - H~ delta), kRefine times
            rc = ensure_inverse();
            if (rc) return rc;
            rc = launch_matvec_axpy(b.Hinv, p, b.g, p, gscale, nullptr, 0.0, b.delta, s);
            if (rc) return rc;
            for (int k = 0; k < kRefine; ++k) {
                rc = launch_matvec_axpy(b.Ha, p, b.delta, p, -1.0, b.g, 1.0, b.rr, s);
                if (rc) return rc;
                rc = launch_matvec_axpy(b.Hinv, p, b.rr, p, gscale, b.delta, 1.0, b.delta, s);
                if (rc) return rc;
            }
            rc = launch_step_stats(b.delta, b.beta, p, b.stats, s);
            if (rc) return rc;
            ord.m = 0; qn_have_gprev = false;        // (the secant pairs belong to the steps of the bare factor)
        }
        const bool qn_now = qn_on && !fresh_now && !have_Ha;
        int push_slot = -1;
        if (qn_now && qn_have_gprev) {               // the accepted step prev -> beta gives a curvature pair
            push_slot = qn_next;
            qn_next = (qn_next + 1) % QN_PAIRS;
            if (ord.m < QN_PAIRS) ord.idx[ord.m++] = push_slot;
            else { for (int k = 0; k + 1 < QN_PAIRS; ++k) ord.idx[k] = ord.idx[k + 1]; ord.idx[QN_PAIRS - 1] = push_slot; }
        }
        const bool fused = qn_now && inv_enabled(p) && ((size_t)4 * p + 48 + 2 * QN_PAIRS) * sizeof(double) <= 64 * 1024;
        if (have_Ha) {
            // (the step is there)
        } else if (fused) {
            // reused factor, secant correction on: invert the factor once, then ONE launch per iteration
            if (!(*b.inv_valid & 1)) {
                rc = launch_tri_inverse(b.L, p, b.Linv, s);
                if (rc) return rc;
                *b.inv_valid = 1;
            }
            if (!(*b.inv_valid & 2)) {
                // H0^-1 = L^-T L^-1 is the Gram matrix of the rows of L^-1 (its upper triangle is zero): one tiny pass of
                // the Gram kernel, after which H0^-1 q is ONE row-wise mat-vec instead of a row-wise and a column-wise
                // triangular one (the column-wise one was 250 dependent loads per thread at p = 500)
                rc = gram_impl_f64(b.Linv, p, nullptr, p, p, b.Hinv, p, 0, b.ws_pass, b.ws_pass_bytes, s);
                if (rc) return rc;
                *b.inv_valid |= 2;
            }
            const size_t shm = ((size_t)2 * p + 48) * sizeof(double);
            // One element per thread up to 1024 columns.  Fewer waves make the kernel's ~20 block reductions cheaper, more waves its
            // mat-vec: 512 threads for p <= 512 (bench/qn_threads_ab.sh, average duration inside chained fits: p = 100 58 -> 45 us,
            // p = 260 82 -> 73, p = 500 335 -> 189; 256 threads: 51 / 94 / 357).  DLSA_QN_THREADS overrides.
            const char* qn_e = knob("DLSA_QN_THREADS");
            const int qn_threads_env = qn_e ? atoi(qn_e) : 0;
            const int qn_default = p <= 512 ? 512 : 1024;
            const int qn_threads = qn_threads_env >= 64 && qn_threads_env <= 1024 && (p <= qn_threads_env || p > 1024) ? qn_threads_env : qn_default;
#define DLSA_QN_STEP(EPT) hipLaunchKernelGGL(qn_step_kernel<EPT>, dim3(1), dim3(qn_threads), shm, s, (const double*)b.beta, \
                                             (const double*)b.prev, b.qn_gprev, (const double*)b.g, b.qn_s, b.qn_y, b.qn_rho, ord, \
                                             push_slot, p, gscale, (const double*)b.Hinv, b.delta, b.stats)
            if (p <= 1024) DLSA_QN_STEP(1);
            else DLSA_QN_STEP(2);
#undef DLSA_QN_STEP
            DLSA_HIP_CHECK(hipGetLastError());
            qn_have_gprev = true;
        } else {
            if (push_slot >= 0)
                hipLaunchKernelGGL(qn_push_kernel, dim3(1), dim3(1024), 0, s, (const double*)b.beta, (const double*)b.prev,
                                   (const double*)b.qn_gprev, (const double*)b.g, p, b.qn_s + (int64_t)push_slot * p,
                                   b.qn_y + (int64_t)push_slot * p, b.qn_rho + push_slot);
            if (qn_on) {                              // remember the gradient at this beta for the next pair
                DLSA_HIP_CHECK(hipMemcpyAsync(b.qn_gprev, b.g, (size_t)p * sizeof(double), hipMemcpyDeviceToDevice, s));
                qn_have_gprev = true;
            }
            const double* rhs = b.g;
            if (qn_now && ord.m > 0) {
                hipLaunchKernelGGL(qn_pre_kernel, dim3(1), dim3(1024), qn_shm, s, (const double*)b.g, (const double*)b.qn_s,
                                   (const double*)b.qn_y, (const double*)b.qn_rho, ord, p, gscale, b.qn_q, b.qn_alpha);
                rhs = b.qn_q;
            } else if (!fresh_now && gscale != 1.0) {
                rc = launch_axpby(b.g, b.g, gscale - 1.0, p, b.g, s);       // g <- g * gscale (inherited factor of H / scale)
                if (rc) return rc;
            }
            if (fresh_now && qn_on && chol_small_ok(p)) {
                // p <= 112: H^-1 and the step in ONE launch (chol.hip: spd_inverse_small_kernel).  No factor, no Linv: with the secant
                // correction on, every later step of this factor goes through the fused quasi-Newton kernel, which reads H^-1 only
                rc = launch_chol_small(H, p, p, rhs, b.beta, b.Hinv, b.delta, b.stats, s);
                *b.inv_valid = 3;
            } else if (fresh_now) {
                *b.inv_valid = 0;
                rc = launch_chol_solve(H, p, 0, rhs, 0, b.beta, 0, p, 1, b.L, b.delta, 0, b.stats, 0, s, 0);
            } else if (inv_enabled(p)) {
                // a reused factor: invert it once, then every solve is two mat-vecs instead of 2p/32 dependent block steps
                if (!(*b.inv_valid & 1)) {
                    rc = launch_tri_inverse(b.L, p, b.Linv, s);
                    if (rc) return rc;
                    *b.inv_valid = 1;
                }
                rc = launch_inv_apply(b.Linv, p, rhs, b.beta, b.delta, b.stats, s);
            } else {
                rc = launch_chol_solve(H, p, 0, rhs, 0, b.beta, 0, p, 1, b.L, b.delta, 0, b.stats, 0, s, 1);
            }
            if (rc) return rc;
            if (qn_now && ord.m > 0) {
                hipLaunchKernelGGL(qn_post_kernel, dim3(1), dim3(1024), qn_shm, s, b.delta, (const double*)b.qn_s,
                                   (const double*)b.qn_y, (const double*)b.qn_rho, (const double*)b.qn_alpha, ord, p, b.stats);
                DLSA_HIP_CHECK(hipGetLastError());
            }
        }
        have_factor = true;
        double h_stack[4];
        double* h = readback_slot();
        if (!h) h = h_stack;
        DLSA_HIP_CHECK(hipMemcpyAsync(h, b.stats, 4 * sizeof(double), hipMemcpyDeviceToHost, s));
        DLSA_HIP_CHECK(hipStreamSynchronize(s));
        ll = h[3];
        *loglik = ll;
        if (trace_on) fprintf(stderr, "[irls] n=%lld it=%d fresh=%d step=%.3e |beta|=%.3e ll=%.10e\n", (long long)n, it, (int)fresh_now + 2 * (int)have_Ha, h[0], h[1], ll);
        if (h[2] == 1.0) { *status = DLSA_PART_NOT_SPD; return DLSA_OK; }
        if (h[2] == 2.0 || !isfinite(ll)) { *status = DLSA_PART_NAN; return DLSA_OK; }
        // safeguard: the previous step overshot (log-likelihood dropped) -> halve it, refresh H
        if (have_prev && ll < ll_prev - 1e-12 * fabs(ll_prev) && halvings < 30) {
            ++halvings;
            ord.m = 0; qn_have_gprev = false;                            // the rejected step gives no valid pair
            approx_ok = false; have_Ha = false;                          // (whatever preconditioner produced it: exact Hessians from here)
            rc = launch_axpby(b.beta, b.prev, -1.0, p, b.delta, s);      // delta = beta - prev
            if (rc) return rc;
            rc = launch_axpby(b.prev, b.delta, 0.5, p, b.beta, s);       // beta = prev + delta/2
            if (rc) return rc;
            need_H = true;
            continue;
        }
        halvings = 0;
        const double scale = std::max(1.0, h[1]);
        near_scale = scale;
        if (h[0] <= tol * scale) {
            *status = DLSA_PART_OK;
            *fresh = fresh_now || peek;
            return DLSA_OK;
        }
        // Predicted convergence: the steps have contracted by r <= 0.2 twice in a row, this one is within 10 tol, and the
        // next one (<= r |delta|) would pass the test: take the step and stop.  The returned coef then meets the tolerance
        // without the confirming logit pass.  b.w and the log-likelihood are those of the PREVIOUS iterate, |delta| <= 10 tol
        // (1e-12 relative at the default tol) away from the returned coef; the closing Gram of irls_fit_core uses them, so
        // Sig_inv / Sig_invMcoef carry a relative error of that order (with a 100 tol window the 2.5e7 x 500 fit showed 3e-11
        // on Sig_invMcoef against the confirmed run: too close to the 1e-10 parity tolerance) and loglik an O(delta^2) one
        // (the gradient vanishes at the MLE).  tests/test_gpu_fullsize.py holds both to 1e-11 at config-3 scale.
        if (predict_on && it >= 3 && isfinite(dprev2) && h[0] <= 10.0 * tol * scale) {
            const double r = std::max(h[0] / dprev, dprev / dprev2);
            if (r <= 0.2 && r * h[0] <= tol * scale) {
                rc = launch_axpby(b.beta, b.delta, 1.0, p, b.beta, s);
                if (rc) return rc;
                *status = DLSA_PART_OK;
                // the closing Gram of irls_fit_core would use b.w of THIS iterate: when this iteration evaluated H (from the
                // same w) the matrix is already there
                *fresh = fresh_now || peek;
                return DLSA_OK;
            }
        }
        // H~ taken e away from the MLE contracts by ~2e per pass: once a step of it shrank by less than 100x and several passes are
        // still ahead, the next pass refreshes it (one refresh at the second iterate is what a warm-started partition needs)
        if (have_Ha && isfinite(dprev) && h[0] > 1e-2 * dprev && h[0] > 1e-9 * scale) want_refresh = true;
        // frozen-Hessian policy: keep the factor while steps are small and still shrinking fast
        if (!fresh_now && h[0] > 0.25 * dprev) need_H = true;            // stalled: refresh
        else need_H = h[0] > freeze_at * scale;
        // The iteration that is expected to CONVERGE runs fused: when the steps contract at rate r and the next one (<= r |delta|)
        // would pass the test, that iteration's logit pass would be followed by the closing Gram of irls_fit_core at the same
        // beta -- two reads of the rows; the fused pass gives the converged check and Sig_inv in one.  A wrong guess costs the
        // difference between a fused pass and a logit pass and buys a fresh Hessian.
        if (fuse_last && !need_H && isfinite(dprev) && h[0] < dprev) {
            // secant-corrected steps contract faster and faster (observed at p = 100: 6.8e-3, 1.1e-3, 9.5e-5): extrapolate the
            // rate by its own trend.  "Ends the run" = passes the step test, or takes the predicted-convergence exit above.
            double r = h[0] / dprev;
            if (isfinite(dprev2) && dprev < dprev2) r *= std::min(1.0, r / (dprev / dprev2));
            const double next = r * h[0];
            if (next <= tol * scale || (predict_on && next <= 10.0 * tol * scale && r * next <= tol * scale)) peek_next = true;
        }
        dprev2 = dprev;
        dprev = h[0];
        rc = launch_advance(b.prev, b.beta, b.delta, p, s);              // prev = beta, beta += delta (one launch: a copy + a launch cost ~10 us of gaps per iteration)
        if (rc) return rc;
        ll_prev = ll;
        have_prev = true;
    }
    return DLSA_OK;
}

// The partition loop shared by every data representation.  make_data(r0) gives the passes over the rows that start
// at row r0; pass_bytes(rows) the scratch those passes need.
// Per-chain state of the partition loop: its buffers and what one partition hands to the next (warm start, inherited / pooled factor).
struct IrlsChain {
    IrlsBuffers b;
    int inv_valid_flag = 0;
    int64_t border_rows = -1;
    char* extra = nullptr;               // the data source's per-chain scratch (the gathered labels of a strided partition)
    bool have_warm = false;
    int64_t factor_rows = 0;             // rows behind the Hessian whose factor sits in b.L (0 = none usable)
    int64_t factor_rows_sub = 0;
    int pooled = 0;                      // partitions summed in b.Hpool
    int64_t pooled_rows = 0;
    int overall = DLSA_OK;
    int rc = DLSA_OK;
    std::string err;
};

// Independent partition CHAINS.  The partitions of one call are fitted one after the other because each starts from its
// predecessor's MLE and factor -- but a fit of a SMALL partition is a string of launches that leave most of the GPU idle
// (config 4's structured shard: 1e6 rows x 76 B per pass = 44 us, then a 31 us single-workgroup quasi-Newton step, ...).
// So small partitions are dealt round-robin to up to four chains, each with its own stream, host thread, workspace slice and
// warm-start state: chain c fits partitions c, c + S, c + 2S, ... in order, the chains overlap on the device.  Results do not
// depend on the overlap (every chain is deterministic; the MLE is unique), only on S, which is a function of the shapes.
constexpr int IRLS_MAX_CHAINS = 8;          // (the default cap is 4: irls_chain_cap)
static int irls_chain_cap(int64_t max_rows, double bytes_per_row) {
    const char* e = knob("DLSA_IRLS_CHAINS");
    if (e) return std::min(IRLS_MAX_CHAINS, std::max(1, atoi(e)));
    // measured (bench/ab_chains.sh, same box, one chain -> four): 76 MB partitions (config 4 structured) 22.1 -> 14.5 ms, 0.8 GB (config 2,
    // K = 10) 14.6 -> 12.2, 2.1 GB (config 4 dense) 67.3 -> 59.9, 4 GB (config 3, K = 25) 287.5 -> 256.4 seeded (unseeded: 300, the
    // extra cold starts cost more than the overlap gives); partitions beyond 8 GB fill the GPU for milliseconds per launch
    return (double)max_rows * bytes_per_row <= 8e9 ? 4 : 1;
}
// Seeding (partition 0 alone, its state copied to every chain) saves S - 1 cold starts but serialises one partition: it pays when
// a cold start is expensive -- not for the 128 MB raw partitions of a structured design (config 4: 14.5 unseeded, 15.9 seeded).
static bool irls_chain_seed(int64_t max_rows, double bytes_per_row) {
    const char* e = knob("DLSA_IRLS_SEED");
    if (e) return atoi(e) != 0;
    return (double)max_rows * bytes_per_row >= 2.56e8;
}

static std::mutex g_chain_mu;
static std::map<std::pair<int, int>, hipStream_t> g_chain_streams;     // (device, chain) -> side stream, created once
static int chain_stream(int dev, int c, hipStream_t* out) {
    std::lock_guard<std::mutex> lk(g_chain_mu);
    auto key = std::make_pair(dev, c);
    auto f = g_chain_streams.find(key);
    if (f == g_chain_streams.end()) {
        hipStream_t st;
        DLSA_HIP_CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        f = g_chain_streams.emplace(key, st).first;
    }
    *out = f->second;
    return DLSA_OK;
}

using IrlsMakeData = std::function<IrlsData(int, int64_t, char*, hipStream_t)>;

static int irls_fit_core(const IrlsMakeData& make_data, const std::function<size_t(int64_t)>& pass_bytes, size_t extra_bytes,
                         int chain_cap, bool chain_seed, const int64_t* part_offsets_host, int K, int p, double tol, int max_iter, double* coef,
                         double* Sig_inv, double* Sig_invMcoef, int* n_iter_host, int* status_host, double* loglik_host,
                         void* ws, size_t ws_bytes, void* stream) {
    int64_t max_rows = 0;
    for (int k = 0; k < K; ++k) {
        DLSA_REQUIRE(part_offsets_host[k + 1] >= part_offsets_host[k], "irls_fit: part_offsets not monotone");
        max_rows = std::max(max_rows, part_offsets_host[k + 1] - part_offsets_host[k]);
    }
    const IrlsLayout l = irls_layout(max_rows, p, pass_bytes(max_rows));
    const size_t extra_al = align_up(extra_bytes, 256), chain_bytes = extra_al + align_up(l.total, 256);
    if (!ws || ws_bytes < chain_bytes || ((uintptr_t)ws & 255)) {
        set_error("irls_fit: workspace %zu bytes needed (256-aligned), got %zu", chain_bytes, ws_bytes);
        return DLSA_ERR_WORKSPACE;
    }
    // chains: at least two partitions each, as many as the workspace holds
    int S = std::min<int64_t>(std::min(chain_cap, (K - 1) / 2), (int64_t)(ws_bytes / chain_bytes));
    S = std::max(S, 1);
    hipStream_t s0 = (hipStream_t)stream;
    // tuning knobs for experiments (defaults are the production policy)
    const char* env_sub = knob("DLSA_IRLS_SUBSAMPLE");
    const char* env_frz = knob("DLSA_IRLS_FREEZE");
    // 0/1 disables the warm start.  Wide designs (p >= 384) start from the smallest of 1/64, 1/32, 1/16 of the rows that still pins the
    // MLE (>= 200 p rows): at p = 500 the 1/16 subsample's own Newton run (Gram passes over 1.5e6 rows) costs more than its slightly
    // better start buys (bench/ab_subsample.sh, 2.5e7 x 500: 254-255 ms at 16, 241-242 at 64; p = 100: 11.3 ms at 16, 11.8-11.9 at 32-64)
    const int sub_div = env_sub ? atoi(env_sub) : 16;
    auto subsample_rows = [&](int64_t nk) -> int64_t {
        if (sub_div <= 1) return 0;
        if (!env_sub && p >= 384)
            for (int d = 64; d > 16; d /= 2) {
                const int64_t ns = nk / d;
                if (ns >= 200 * (int64_t)p && ns >= 50000) return ns;
            }
        if (!env_sub && p >= 192)          // a partition too short for 1/16 (1e6 x 500: 62500 < 200 p rows) still profits from a larger share
            for (int d = 16; d >= 4; d /= 2) {
                const int64_t ns = nk / d;
                if (ns >= 200 * (int64_t)p && ns >= 50000) return ns;
            }
        return nk / sub_div;
    };
    const double freeze_at = env_frz ? atof(env_frz) : 1.0;    // 0 disables the frozen Hessian

    const char* env_warm = knob("DLSA_IRLS_WARM");
    const bool warm_ok = env_warm ? atoi(env_warm) != 0 : true;   // 0 disables partition-to-partition warm starts
    const char* env_inh = knob("DLSA_IRLS_INHERIT");
    // Starting from an inherited Cholesky factor trades one Gram pass (~n p^2 flops) for a few more logit passes
    // (~n p bytes each): worth it once the Gram pass costs several logit passes, i.e. for p of a few hundred
    // (measured: 2.5e7 x 500 fit 0.41 -> 0.33 s; at p = 100 the extra iterations cost more than the Gram they save).
    const bool inherit_ok = env_inh ? atoi(env_inh) != 0 : (p >= 192);
    // Lean passes (round 4): where the fused Newton pass serves a partition (49 <= p <= 120) no pass writes the weight vector --
    // the stand-in Hessians of inherited factors are the one consumer the fused pass does not replace, so only without them.
    // Config 2: the fused pass 2.53 -> 2.22 ms, the logit pass 1.39 -> 1.3 ms.  DLSA_IRLS_LEAN=0 keeps the weights.
    const char* env_lean = knob("DLSA_IRLS_LEAN");
    const bool lean = (env_lean ? atoi(env_lean) != 0 : true) && !inherit_ok;
    const char* env_fd = knob("DLSA_IRLS_FACTOR_DIV");
    const int fac_div = env_fd ? atoi(env_fd) : 4;              // rows / fac_div feed the stand-in Hessian (0/1: the subsample's)
    // Pooled preconditioner: the exact Hessians of the finished partitions (their Sig_inv, evaluated at their MLEs) are
    // summed, and after 1, 2, 4, 8, ... partitions the sum is factored and replaces the inherited factor.  Partitions of
    // one data set share the population Hessian, so the sum over m partitions misses the next partition's Hessian only by
    // its own sampling noise (~sqrt(p/n_k)) plus 1/sqrt(m) of it -- half the error of the stand-in from a quarter of the
    // first partition -- and the quasi-Newton iterations of every later partition get shorter for O(log K) factorizations.
    const char* env_pool = knob("DLSA_IRLS_POOL");
    const bool pool_ok = (env_pool ? atoi(env_pool) != 0 : true) && inherit_ok && K > 1;

    std::vector<IrlsChain> chains((size_t)S);
    for (int c = 0; c < S; ++c) {
        IrlsChain& cs = chains[(size_t)c];
        char* cbase = (char*)ws + (size_t)c * chain_bytes;
        cs.extra = cbase;
        char* base = cbase + extra_al;
        IrlsBuffers& b = cs.b;
        b.w = (double*)(base + l.w);
        b.g = (double*)(base + l.g);
        b.beta = (double*)(base + l.beta);
        b.prev = (double*)(base + l.beta_prev);
        b.delta = (double*)(base + l.delta);
        b.stats = (double*)(base + l.stats);   // [0..2] solver stats, [3] loglik
        b.L = (double*)(base + l.L);
        b.Linv = (double*)(base + l.Linv);
        b.Hinv = (double*)(base + l.Hinv);
        b.Hpool = (double*)(base + l.Hpool);
        b.Ha = l.Ha ? (double*)(base + l.Ha) : nullptr;
        b.rr = (double*)(base + l.rr);
        b.inv_valid = &cs.inv_valid_flag;
        b.border = (double*)(base + l.border); b.border_rows = &cs.border_rows;
        b.qn_s = (double*)(base + l.qn_s); b.qn_y = (double*)(base + l.qn_y); b.qn_rho = (double*)(base + l.qn_rho);
        b.qn_alpha = (double*)(base + l.qn_alpha); b.qn_q = (double*)(base + l.qn_q); b.qn_gprev = (double*)(base + l.qn_gprev);
        b.ws_pass = base + l.pass;
        b.ws_pass_bytes = l.pass_bytes;
    }

    // one partition on one chain (everything below is enqueued on the chain's stream s)
    auto fit_partition = [&](int k, IrlsChain& cs, hipStream_t s) -> int {
        const int64_t r0 = part_offsets_host[k];
        const int64_t nk = part_offsets_host[k + 1] - r0;
        const IrlsData d = make_data(k, r0, cs.extra, s);
        double* Hk = Sig_inv + (int64_t)k * p * p;
        double* ck = coef + (int64_t)k * p;
        double* sk = Sig_invMcoef + (int64_t)k * p;
        int st = DLSA_PART_OK, iters = 0, grams = 0;
        double ll = 0.0;
        if (nk == 0) {
            // empty partition: the reference's zero block (models.py:84-91)
            DLSA_HIP_CHECK(hipMemsetAsync(Hk, 0, (size_t)p * p * sizeof(double), s));
            DLSA_HIP_CHECK(hipMemsetAsync(ck, 0, (size_t)p * sizeof(double), s));
            DLSA_HIP_CHECK(hipMemsetAsync(sk, 0, (size_t)p * sizeof(double), s));
            st = DLSA_PART_EMPTY;
        } else {
            bool fresh = false;
            int rc;
            // Start from the previous partition's MLE when there is one: partitions of one data set
            // estimate the same theta, so the iterate starts O(sqrt(p/n_k)) from the answer and the
            // fit needs ~2 Gram passes instead of ~6.  The MLE is unique, so the result is the same;
            // a warm start that fails (status != OK) is repeated cold.
            bool warm = cs.have_warm;
            if (warm && pool_ok && cs.pooled > 0 && (cs.pooled & (cs.pooled - 1)) == 0) {
                if (qn_enabled() && chol_small_ok(p)) {
                    rc = launch_chol_small(cs.b.Hpool, p, p, cs.b.g, cs.b.beta, cs.b.Hinv, cs.b.delta, cs.b.stats, s);
                    *cs.b.inv_valid = 3;
                } else {
                    rc = launch_chol_solve(cs.b.Hpool, p, 0, cs.b.g, 0, cs.b.beta, 0, p, 1, cs.b.L, cs.b.delta, 0, cs.b.stats, 0, s, 0);
                    *cs.b.inv_valid = 0;
                }
                if (rc) return rc;
                bool ok = false;
                rc = factor_ok(cs.b, s, &ok);
                if (rc) return rc;
                cs.factor_rows = ok ? cs.pooled_rows : 0;          // a failed cs.pooled factor: this partition takes its own Hessian
            }
            for (int attempt = 0; attempt < 2; ++attempt) {
                st = DLSA_PART_OK; iters = 0; grams = 0;
                double inherit = (warm && inherit_ok && cs.factor_rows > 0) ? (double)nk / (double)cs.factor_rows : 0.0;
                if (!warm) {
                    DLSA_HIP_CHECK(hipMemsetAsync(cs.b.beta, 0, (size_t)p * sizeof(double), s));
                    cs.factor_rows = 0;
                    // cold start of a large partition: solve its leading 1/sub_div rows first
                    const int64_t nsub = subsample_rows(nk);
                    if (nsub >= 200 * (int64_t)p && nsub >= 50000) {
                        int st_sub = 0, it_sub = 0, gr_sub = 0;
                        double ll_sub = 0.0;
                        // ... and that subsample starts from ITS leading 1/sub_div (rows / 256): the first Newton
                        // iterations from zero, the expensive ones, run on the smallest sample that still pins the MLE
                        const int64_t nsub2 = nsub / sub_div;
                        if (nsub2 >= 100 * (int64_t)p && nsub2 >= 20000) {
                            rc = newton_run(d, nsub2, p, 1e-3, max_iter, freeze_at, Hk, cs.b, s, &st_sub, &it_sub, &gr_sub,
                                            &ll_sub, &fresh, 0.0, lean);
                            if (rc) return rc;
                            if (st_sub != DLSA_PART_OK) DLSA_HIP_CHECK(hipMemsetAsync(cs.b.beta, 0, (size_t)p * sizeof(double), s));
                            st_sub = 0; it_sub = 0; gr_sub = 0;
                        }
                        rc = newton_run(d, nsub, p, 1e-6, max_iter, freeze_at, Hk, cs.b, s, &st_sub, &it_sub,
                                        &gr_sub, &ll_sub, &fresh, 0.0, lean);
                        if (rc) return rc;
                        if (st_sub != DLSA_PART_OK) {   // degenerate subsample: plain cold start
                            DLSA_HIP_CHECK(hipMemsetAsync(cs.b.beta, 0, (size_t)p * sizeof(double), s));
                        } else if (inherit_ok) {
                            // Factor a Hessian AT the subsample MLE that will stand in for the full one.  The quasi-Newton
                            // contraction rate is the relative sampling error of that Hessian, ~2 sqrt(p / rows): on the
                            // 1/16 subsample that is 0.08 at p = 500 (ten full logit passes to 1e-13); one Gram pass over
                            // a quarter of the rows (a quarter of a full pass) brings it to 0.02 (six passes).
                            const int64_t nfac = fac_div > 1 ? std::max<int64_t>(nsub, nk / fac_div) : nsub;
                            if (nfac > nsub) {
                                rc = d.logit(cs.b.beta, nfac, cs.b.w, nullptr, nullptr, cs.b, s);
                                if (rc) return rc;
                            }
                            if (nfac > nsub || !fresh) {     // (cs.b.w of the subsample run are the weights at exactly this beta)
                                rc = d.gram(cs.b.w, nfac, Hk, cs.b, s);
                                if (rc) return rc;
                            }
                            rc = launch_chol_solve(Hk, p, 0, cs.b.g, 0, cs.b.beta, 0, p, 1, cs.b.L, cs.b.delta, 0, cs.b.stats, 0, s, 0);
                            if (rc) return rc;
                            *cs.b.inv_valid = 0;
                            bool ok = false;
                            rc = factor_ok(cs.b, s, &ok);
                            if (rc) return rc;
                            if (ok) {                        // a stand-in that does not factor is simply not inherited
                                inherit = (double)nk / (double)nfac;
                                cs.factor_rows_sub = nfac;
                            }
                        }
                    }
                }
                rc = newton_run(d, nk, p, tol, max_iter, freeze_at, Hk, cs.b, s, &st, &iters, &grams, &ll, &fresh,
                                inherit, lean);
                if (rc) return rc;
                if (grams > 0) cs.factor_rows = nk;         // cs.b.L now factors a Hessian of this partition
                else if (inherit > 0.0 && cs.factor_rows == 0) cs.factor_rows = cs.factor_rows_sub;
                if (st == DLSA_PART_OK || !warm) break;
                warm = false;                            // warm start failed: repeat from zero
            }
            if (st != DLSA_PART_OK) cs.factor_rows = 0;
            cs.have_warm = (st == DLSA_PART_OK) && warm_ok;
            if (st == DLSA_PART_OK && !fresh) {
                // Sig_inv must be the Hessian AT the returned coef: cs.b.w holds the weights of the last logit pass, which ran
                // at exactly this beta -- or, after a predicted exit (newton_run), one step of <= 10 tol before it
                // (lean runs wrote no weights: the fused pass evaluates H at the returned coef itself)
                if (lean && d.lean_w && d.pass && d.fusable && d.fusable(nk)) rc = d.pass(cs.b.beta, nk, nullptr, cs.b.g, cs.b.stats + 3, Hk, cs.b, s);
                else rc = d.gram(cs.b.w, nk, Hk, cs.b, s);
                if (rc) return rc;
            } else if (st == DLSA_PART_NOT_CONVERGED) {
                // report the Hessian at the last iterate
                rc = d.logit(cs.b.beta, nk, cs.b.w, nullptr, nullptr, cs.b, s);
                if (rc) return rc;
                rc = d.gram(cs.b.w, nk, Hk, cs.b, s);
                if (rc) return rc;
            }
            DLSA_HIP_CHECK(hipMemcpyAsync(ck, cs.b.beta, (size_t)p * sizeof(double), hipMemcpyDeviceToDevice, s));
            rc = launch_matvec(Hk, p, cs.b.beta, p, sk, s);
            if (rc) return rc;
            if (pool_ok && st == DLSA_PART_OK) {
                if (cs.pooled == 0) DLSA_HIP_CHECK(hipMemcpyAsync(cs.b.Hpool, Hk, (size_t)p * p * sizeof(double), hipMemcpyDeviceToDevice, s));
                else {
                    rc = launch_axpby(cs.b.Hpool, Hk, 1.0, p * p, cs.b.Hpool, s);
                    if (rc) return rc;
                }
                ++cs.pooled;
                cs.pooled_rows += nk;
            }
        }
        if (n_iter_host) n_iter_host[k] = iters;
        if (status_host) status_host[k] = st;
        if (loglik_host) loglik_host[k] = ll;
        if (st == DLSA_PART_NOT_CONVERGED && cs.overall == DLSA_OK) cs.overall = DLSA_ERR_NOT_CONVERGED;
        if (st == DLSA_PART_NOT_SPD && cs.overall == DLSA_OK) cs.overall = DLSA_ERR_NOT_SPD;
        if (st == DLSA_PART_NAN && cs.overall == DLSA_OK) cs.overall = DLSA_ERR_NAN;
        return DLSA_OK;
    };
    // Seeded chains (irls_chain_seed): partition 0 is fitted alone, its MLE, factor, inverse and pooled Hessian are
    // copied into every chain's slice, and the chains share out partitions 1 .. K - 1 -- ONE cold start per call instead of S.
    const bool seeded = S > 1 && chain_seed;
    const int k_first = seeded ? 1 : 0;
    auto run_chain = [&](int c, hipStream_t s) -> int {
        IrlsChain& cs = chains[(size_t)c];
        for (int k = k_first + c; k < K; k += S) {
            const int rc = fit_partition(k, cs, s);
            if (rc) return rc;
        }
        DLSA_HIP_CHECK(hipStreamSynchronize(s));
        return DLSA_OK;
    };

    int rc0 = DLSA_OK;
    if (S == 1) {
        rc0 = run_chain(0, s0);
    } else {
        int dev = 0;
        DLSA_HIP_CHECK(hipGetDevice(&dev));
        if (seeded) {
            IrlsChain& c0 = chains[0];
            const int rc = fit_partition(0, c0, s0);
            if (rc) return rc;
            const size_t pb = (size_t)p * sizeof(double), ppb = (size_t)p * p * sizeof(double);
            for (int c = 1; c < S; ++c) {
                IrlsChain& cs = chains[(size_t)c];
                DLSA_HIP_CHECK(hipMemcpyAsync(cs.b.beta, c0.b.beta, pb, hipMemcpyDeviceToDevice, s0));
                DLSA_HIP_CHECK(hipMemcpyAsync(cs.b.L, c0.b.L, ppb, hipMemcpyDeviceToDevice, s0));
                DLSA_HIP_CHECK(hipMemcpyAsync(cs.b.Linv, c0.b.Linv, ppb, hipMemcpyDeviceToDevice, s0));
                DLSA_HIP_CHECK(hipMemcpyAsync(cs.b.Hinv, c0.b.Hinv, ppb, hipMemcpyDeviceToDevice, s0));
                DLSA_HIP_CHECK(hipMemcpyAsync(cs.b.Hpool, c0.b.Hpool, ppb, hipMemcpyDeviceToDevice, s0));
                // (the pooled factor's refresh solves against b.g before the chain's first pass has written it: a defined right-hand side,
                // or the discarded solution's finiteness check would depend on what the workspace held)
                DLSA_HIP_CHECK(hipMemcpyAsync(cs.b.g, c0.b.g, pb, hipMemcpyDeviceToDevice, s0));
                cs.inv_valid_flag = c0.inv_valid_flag;
                cs.have_warm = c0.have_warm;
                cs.factor_rows = c0.factor_rows;
                cs.factor_rows_sub = c0.factor_rows_sub;
                cs.pooled = c0.pooled;
                cs.pooled_rows = c0.pooled_rows;
            }
        }
        hipEvent_t fork;
        DLSA_HIP_CHECK(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
        DLSA_HIP_CHECK(hipEventRecord(fork, s0));                       // the side chains start behind the seed (and whatever the caller enqueued)
        std::vector<hipStream_t> st((size_t)S, s0);
        for (int c = 1; c < S; ++c) {
            int rc = chain_stream(dev, c, &st[(size_t)c]);
            if (rc) { (void)hipEventDestroy(fork); return rc; }
            DLSA_HIP_CHECK(hipStreamWaitEvent(st[(size_t)c], fork, 0));
        }
        std::vector<std::thread> workers;
        const dlsa_irls_options caller_opt = irls_options_snapshot();      // the chains' threads run under the caller's options
        const dlsa_kernel_options caller_kopt = kernel_options_snapshot();
        for (int c = 1; c < S; ++c)
            workers.emplace_back([&, c]() {
                irls_options_adopt(caller_opt);
                kernel_options_adopt(caller_kopt);
                IrlsChain& cs = chains[(size_t)c];
                if (hipSetDevice(dev) != hipSuccess) { cs.rc = DLSA_ERR_HIP; cs.err = "hipSetDevice failed in a chain thread"; return; }
                cs.rc = run_chain(c, st[(size_t)c]);
                if (cs.rc) { char buf[512]; dlsa_last_error(buf, (int)sizeof(buf)); cs.err = buf; }
            });
        rc0 = run_chain(0, s0);
        for (auto& t : workers) t.join();
        // (every chain has synchronised its stream: whatever the caller enqueues next on `stream` sees all results)
        (void)hipEventDestroy(fork);
        for (int c = 1; c < S && rc0 == DLSA_OK; ++c)
            if (chains[(size_t)c].rc) { set_error("%s", chains[(size_t)c].err.c_str()); rc0 = chains[(size_t)c].rc; }
    }
    if (rc0) return rc0;
    int overall = DLSA_OK;
    for (const IrlsChain& cs : chains) {
        if (cs.overall == DLSA_ERR_NOT_CONVERGED && overall == DLSA_OK) overall = DLSA_ERR_NOT_CONVERGED;
        if (cs.overall == DLSA_ERR_NOT_SPD && overall == DLSA_OK) overall = DLSA_ERR_NOT_SPD;
        if (cs.overall == DLSA_ERR_NAN && overall == DLSA_OK) overall = DLSA_ERR_NAN;
    }
    if (overall == DLSA_ERR_NOT_CONVERGED) set_error("irls_fit: at least one partition hit max_iter");
    if (overall == DLSA_ERR_NOT_SPD) set_error("irls_fit: a partition's Hessian is not positive definite");
    if (overall == DLSA_ERR_NAN) set_error("irls_fit: NaN/Inf in a partition's fit");
    return overall;
}

static size_t dense_pass_bytes(int64_t rows, int p) {
    const size_t wide = (p >= 121 && p <= 512 && rows >= 32768 && rows <= kWideMaxRows) ? irls_wide_workspace_bytes(rows, p, 0) : 0;
    return std::max(std::max(irls_pass_workspace_bytes_impl(rows, p), wide), std::max(gram_workspace_bytes_impl(rows, p, 8), logit_workspace_bytes_impl(rows, p)));
}

// ---- implicit intercept (the ones column of models.py:121-122 is never materialised) --------------------------------
// H = [1 | X]' diag(w) [1 | X]  (p + 1) x (p + 1):  the p x p block is the Gram kernel's, the border is sum(w) and X'w from
// one streaming pass (xtv kernel: HBM-bound, ~1/6 of the Gram pass at p = 500); w == nullptr = all ones.
__global__ void icpt_border_kernel(double* __restrict__ H, int64_t ldh, int p) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < p) H[(int64_t)(i + 1) * ldh] = H[i + 1];
}
// (border: [sum w | X'w] already known -- the logit pass that produced w left it, logit_pass_border_impl -- instead of a pass of its own)
int gram_icpt_impl(const double* X, int64_t ldx, const double* w, int64_t n, int p, double* H, int64_t ldh,
                   void* ws, size_t ws_bytes, hipStream_t s, const double* border = nullptr) {
    int rc = gram_impl_f64(X, ldx, w, n, p, H + ldh + 1, ldh, 0, ws, ws_bytes, s);
    if (rc) return rc;
    if (n == 0) { DLSA_HIP_CHECK(hipMemsetAsync(H, 0, (size_t)(p + 1) * sizeof(double), s)); }
    else if (border) { DLSA_HIP_CHECK(hipMemcpyAsync(H, border, (size_t)(p + 1) * sizeof(double), hipMemcpyDeviceToDevice, s)); }
    else {
        rc = xtv_impl(X, ldx, w, n, p, H + 1, nullptr, H, ws, ws_bytes, s);       // row 0 = [sum w | X'w]
        if (rc) return rc;
    }
    hipLaunchKernelGGL(icpt_border_kernel, dim3((p + 255) / 256), dim3(256), 0, s, H, ldh, p);
    DLSA_HIP_CHECK(hipGetLastError());
    return DLSA_OK;
}

// out[j] = y[first + j * step]
__global__ void gather_strided_kernel(const double* __restrict__ y, int64_t first, int64_t step, int64_t n, double* __restrict__ out) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) out[j] = y[first + j * step];
}

}  // namespace dlsa

extern "C" {

int dlsa_irls_last_fit_path(void) { return dlsa::g_last_fit_path; }

size_t dlsa_irls_workspace_bytes(int64_t max_rows_per_partition, int p) {
    if (p <= 0 || p > 2048 || max_rows_per_partition < 0) return 0;
    // one slice per partition chain (irls_fit_core): small partitions are fitted on up to four streams at once
    return dlsa::align_up(dlsa::irls_layout(max_rows_per_partition, p, dlsa::dense_pass_bytes(max_rows_per_partition, p)).total, 256) *
           (size_t)dlsa::irls_chain_cap(max_rows_per_partition, 8.0 * p);
}

int dlsa_irls_fit_f64(const double* X, int64_t ldx, const double* y, const int64_t* part_offsets_host,
                      int K, int p, double tol, int max_iter, double* coef, double* Sig_inv,
                      double* Sig_invMcoef, int* n_iter_host, int* status_host, double* loglik_host,
                      void* ws, size_t ws_bytes, void* stream) {
    using namespace dlsa;
    DLSA_REQUIRE(X && y && part_offsets_host && coef && Sig_inv && Sig_invMcoef, "irls_fit: null argument");
    DLSA_REQUIRE(K > 0 && p > 0 && p <= 2048 && ldx >= p, "irls_fit: bad shape K=%d p=%d ldx=%lld", K, p, (long long)ldx);
    DLSA_REQUIRE(max_iter > 0 && tol > 0, "irls_fit: bad tol/max_iter");
    {
        // the workspace contract holds whichever driver the shapes pick (the lock-step driver allocates from the stream's pool, but a
        // call that is wrong for one driver must not pass because the cost model chose another)
        int64_t mx = 0;
        for (int k = 0; k < K; ++k) mx = std::max(mx, part_offsets_host[k + 1] - part_offsets_host[k]);
        const size_t need = align_up(irls_layout(mx, p, dense_pass_bytes(mx, p)).total, 256);
        if (!ws || ws_bytes < need || ((uintptr_t)ws & 255)) {
            set_error("irls_fit: workspace %zu bytes needed (256-aligned), got %zu", need, ws_bytes);
            return DLSA_ERR_WORKSPACE;
        }
    }
    {
        std::vector<int64_t> rows((size_t)K);
        bool mono = true;
        for (int k = 0; k < K; ++k) { rows[(size_t)k] = part_offsets_host[k + 1] - part_offsets_host[k]; mono &= rows[(size_t)k] >= 0; }
        // (both the one-launch kernel and the lock step beat the chains: the cheaper estimate of the two -- 300 x 3e4 x 64: 17 / 9.7 ms)
        double t_small = 0.0, t_lock = 0.0;
        const bool small_ok = mono && K <= 8192 && irls_small_eligible(rows.data(), K, p, &t_small) && ws && ws_bytes >= irls_small_workspace_bytes(K) &&
                              !((uintptr_t)ws & 255);
        const bool lock_ok = mono && irls_batched_eligible(X, ldx, y, rows.data(), K, p, 0, 1, &t_lock);
        if (small_ok && !(lock_ok && t_lock < t_small)) {
            g_last_fit_path = 1;
            return irls_small_fit(X, ldx, y, part_offsets_host, rows.data(), 1, K, p, 0, tol, max_iter, coef, Sig_inv, Sig_invMcoef,
                                  n_iter_host, status_host, loglik_host, ws, ws_bytes, (hipStream_t)stream);
        }
        if (lock_ok) {
            g_last_fit_path = 2;
            return irls_batched_fit(X, ldx, y, part_offsets_host, rows.data(), 1, K, p, 0, tol, max_iter, coef, Sig_inv, Sig_invMcoef, n_iter_host,
                                    status_host, loglik_host, (hipStream_t)stream);
        }
        g_last_fit_path = 0;
    }
    auto make_data = [=](int, int64_t r0, char*, hipStream_t) {
        const double* Xk = X + r0 * ldx;
        const double* yk = y + r0;
        IrlsData d;
        d.logit = [=](const double* beta, int64_t nrows, double* w, double* g, double* ll, const IrlsBuffers& b, hipStream_t s) {
            return logit_pass_impl(Xk, ldx, yk, beta, nrows, p, w, g, ll, b.ws_pass, b.ws_pass_bytes, s, 0);
        };
        d.gram = [=](const double* w, int64_t nrows, double* H, const IrlsBuffers& b, hipStream_t s) {
            return gram_impl_f64(Xk, ldx, w, nrows, p, H, p, 0, b.ws_pass, b.ws_pass_bytes, s);
        };
        d.pass = [=](const double* beta, int64_t nrows, double* w, double* g, double* ll, double* H, const IrlsBuffers& b, hipStream_t s) {
            return irls_pass_impl(Xk, ldx, yk, beta, nrows, p, H, p, g, ll, w, nullptr, b.ws_pass, b.ws_pass_bytes, s, nullptr);
        };
        d.fusable = [=](int64_t nrows) { return irls_pass_fused_eligible(Xk, ldx, yk, nrows, p); };
        d.lean_w = true;
        d.approx = [=](const double* beta, int64_t nrows, double* w, double* g, double* ll, double* Ha, const IrlsBuffers& b, hipStream_t s) {
            return irls_wide_pass_impl(Xk, ldx, yk, beta, nrows, p, 0, w, g, ll, Ha, p, b.ws_pass, b.ws_pass_bytes, s);
        };
        d.approxable = [=](int64_t nrows, size_t wsb) { return nrows <= kWideMaxRows && irls_wide_eligible(Xk, ldx, nrows, p, 0) && irls_wide_workspace_bytes(nrows, p, 0) <= wsb; };
        return d;
    };
    int64_t max_rows = 0;
    for (int k = 0; k < K; ++k) max_rows = std::max(max_rows, part_offsets_host[k + 1] - part_offsets_host[k]);
    return irls_fit_core(make_data, [=](int64_t rows) { return dense_pass_bytes(rows, p); }, 0, irls_chain_cap(max_rows, 8.0 * p),
                         irls_chain_seed(max_rows, 8.0 * p), part_offsets_host, K, p, tol, max_iter, coef, Sig_inv, Sig_invMcoef, n_iter_host, status_host, loglik_host,
                         ws, ws_bytes, stream);
}

size_t dlsa_irls_ex_workspace_bytes(int64_t max_rows_per_partition, int p, int intercept, int64_t row_step) {
    if (p <= 0 || p + (intercept ? 1 : 0) > 2048 || max_rows_per_partition < 0 || row_step < 1) return 0;
    const int pe = p + (intercept ? 1 : 0);
    const size_t ybuf = row_step > 1 ? dlsa::align_up((size_t)std::max<int64_t>(max_rows_per_partition, 1) * sizeof(double), 256) : 0;
    const size_t chain = ybuf + dlsa::align_up(dlsa::irls_layout(max_rows_per_partition, pe, std::max(dlsa::dense_pass_bytes(max_rows_per_partition, p),
                                                                         dlsa::dense_pass_bytes(max_rows_per_partition, pe))).total, 256);
    return chain * (size_t)dlsa::irls_chain_cap(max_rows_per_partition, 8.0 * p) + (1 << 20);
}

int dlsa_irls_fit_ex_f64(const double* X, int64_t ldx, const double* y, const int64_t* part_first_host,
                         const int64_t* part_rows_host, int64_t row_step, int K, int p, int intercept, double tol, int max_iter,
                         double* coef, double* Sig_inv, double* Sig_invMcoef, int* n_iter_host, int* status_host,
                         double* loglik_host, void* ws, size_t ws_bytes, void* stream) {
    using namespace dlsa;
    DLSA_REQUIRE(X && y && part_first_host && part_rows_host && coef && Sig_inv && Sig_invMcoef, "irls_fit_ex: null argument");
    const int pe = p + (intercept ? 1 : 0);
    DLSA_REQUIRE(K > 0 && p > 0 && pe <= 2048 && ldx >= p && row_step >= 1, "irls_fit_ex: bad shape K=%d p=%d ldx=%lld step=%lld", K, p,
                 (long long)ldx, (long long)row_step);
    DLSA_REQUIRE(max_iter > 0 && tol > 0, "irls_fit_ex: bad tol/max_iter");
    int64_t max_rows = 0;
    std::vector<int64_t> offs((size_t)K + 1, 0);
    for (int k = 0; k < K; ++k) {
        DLSA_REQUIRE(part_rows_host[k] >= 0 && part_first_host[k] >= 0, "irls_fit_ex: negative partition shape");
        max_rows = std::max(max_rows, part_rows_host[k]);
        offs[(size_t)k + 1] = offs[(size_t)k] + part_rows_host[k];
    }
    const size_t need = dlsa_irls_ex_workspace_bytes(max_rows, p, intercept, row_step);
    if (!ws || ws_bytes < need || ((uintptr_t)ws & 255)) {
        set_error("irls_fit_ex: workspace %zu bytes needed (256-aligned), got %zu", need, ws_bytes);
        return DLSA_ERR_WORKSPACE;
    }
    double t_small = 0.0, t_lock = 0.0;
    const bool small_ok = irls_small_eligible(part_rows_host, K, pe, &t_small);
    const bool lock_ok = irls_batched_eligible(X, ldx, y, part_rows_host, K, p, intercept, row_step, &t_lock);
    if (small_ok && !(lock_ok && t_lock < t_small)) {
        g_last_fit_path = 1;
        return irls_small_fit(X, ldx, y, part_first_host, part_rows_host, row_step, K, p, intercept, tol, max_iter, coef, Sig_inv,
                              Sig_invMcoef, n_iter_host, status_host, loglik_host, ws, ws_bytes, (hipStream_t)stream);
    }
    if (lock_ok) {
        g_last_fit_path = 2;
        return irls_batched_fit(X, ldx, y, part_first_host, part_rows_host, row_step, K, p, intercept, tol, max_iter, coef, Sig_inv, Sig_invMcoef, n_iter_host,
                                status_host, loglik_host, (hipStream_t)stream);
    }
    g_last_fit_path = 0;
    const size_t ybytes = row_step > 1 ? align_up((size_t)std::max<int64_t>(max_rows, 1) * sizeof(double), 256) : 0;
    const int64_t pitch = ldx * row_step;                     // rows first, first + step, ...: a strided view, no copy of X
    auto make_data = [=](int k, int64_t, char* extra, hipStream_t st) {
        const double* Xk = X + part_first_host[k] * ldx;
        const double* yk = y + part_first_host[k];
        const int64_t nk = part_rows_host[k];
        if (row_step > 1 && nk > 0) {                         // the labels of the partition, gathered once (8 bytes per row) into the chain's slice
            double* ybuf = (double*)extra;
            hipLaunchKernelGGL(gather_strided_kernel, dim3((unsigned)((nk + 255) / 256)), dim3(256), 0, st, y, part_first_host[k],
                               row_step, nk, ybuf);
            yk = ybuf;
        }
        IrlsData d;
        // with the implicit intercept the logit pass also leaves the Hessian's border [sum w | X'w] of its rows and weights: the Gram over
        // the SAME rows (the driver always takes w from the last logit pass) then skips its own border pass -- one read of the partition
        // less per fresh Hessian (config 3's reference-faithful call: 15.6 ms of every 107)
        const bool with_border = intercept && logit_border_ok(Xk, pitch, p);
        d.logit = [=](const double* beta, int64_t nrows, double* w, double* g, double* ll, const IrlsBuffers& b, hipStream_t s) {
            if (with_border && w) {
                *b.border_rows = nrows;
                return logit_pass_border_impl(Xk, pitch, yk, beta, nrows, p, w, g, ll, b.border, b.ws_pass, b.ws_pass_bytes, s);
            }
            if (intercept) *b.border_rows = -1;
            return logit_pass_impl(Xk, pitch, yk, beta, nrows, p, w, g, ll, b.ws_pass, b.ws_pass_bytes, s, intercept);
        };
        d.gram = [=](const double* w, int64_t nrows, double* H, const IrlsBuffers& b, hipStream_t s) {
            if (intercept) return gram_icpt_impl(Xk, pitch, w, nrows, p, H, pe, b.ws_pass, b.ws_pass_bytes, s,
                                                 (w == b.w && *b.border_rows == nrows) ? b.border : nullptr);
            return gram_impl_f64(Xk, pitch, w, nrows, p, H, p, 0, b.ws_pass, b.ws_pass_bytes, s);
        };
        if (!intercept) {
            d.pass = [=](const double* beta, int64_t nrows, double* w, double* g, double* ll, double* H, const IrlsBuffers& b, hipStream_t s) {
                return irls_pass_impl(Xk, pitch, yk, beta, nrows, p, H, p, g, ll, w, nullptr, b.ws_pass, b.ws_pass_bytes, s, nullptr);
            };
            d.fusable = [=](int64_t nrows) { return irls_pass_fused_eligible(Xk, pitch, yk, nrows, p); };
            d.lean_w = true;
        } else {                 // round 4: the fused kernel carries the implicit intercept as a ones column in its LDS stages
            d.pass = [=](const double* beta, int64_t nrows, double* w, double* g, double* ll, double* H, const IrlsBuffers& b, hipStream_t s) {
                *b.border_rows = -1;                   // (this launch rewrites b.w: the border a logit pass left belongs to other weights)
                if (!irls_pass_fused_icpt_eligible(Xk, pitch, yk, nrows, p)) {
                    int rc = logit_pass_impl(Xk, pitch, yk, beta, nrows, p, w, g, ll, b.ws_pass, b.ws_pass_bytes, s, 1);
                    return rc ? rc : gram_icpt_impl(Xk, pitch, w, nrows, p, H, pe, b.ws_pass, b.ws_pass_bytes, s);
                }
                return irls_pass_icpt_impl(Xk, pitch, yk, beta, nrows, p, H, pe, g, ll, w, b.ws_pass, b.ws_pass_bytes, s);
            };
            d.fusable = [=](int64_t nrows) { return irls_pass_fused_icpt_eligible(Xk, pitch, yk, nrows, p); };
            d.lean_w = true;
        }
        d.approx = [=](const double* beta, int64_t nrows, double* w, double* g, double* ll, double* Ha, const IrlsBuffers& b, hipStream_t s) {
            if (intercept) *b.border_rows = -1;             // (this pass rewrites b.w: the border a logit pass left belongs to other weights)
            return irls_wide_pass_impl(Xk, pitch, yk, beta, nrows, p, intercept, w, g, ll, Ha, pe, b.ws_pass, b.ws_pass_bytes, s);
        };
        d.approxable = [=](int64_t nrows, size_t wsb) { return nrows <= kWideMaxRows && irls_wide_eligible(Xk, pitch, nrows, p, intercept) && irls_wide_workspace_bytes(nrows, p, intercept) <= wsb; };
        return d;
    };
    return irls_fit_core(make_data, [=](int64_t rows) { return std::max(dense_pass_bytes(rows, p), dense_pass_bytes(rows, pe)); },
                         ybytes, irls_chain_cap(max_rows, 8.0 * p), irls_chain_seed(max_rows, 8.0 * p), offs.data(), K, pe, tol, max_iter, coef, Sig_inv, Sig_invMcoef,
                         n_iter_host, status_host, loglik_host, ws, ws_bytes, stream);
}

// The same Hessian as a stand-alone entry: H = [1 | X]' diag(w) [1 | X], (p + 1) x (p + 1), intercept first (models.py:121-130).
size_t dlsa_gram_icpt_workspace_bytes(int64_t n, int p) {
    if (p <= 0 || p > 2047 || n < 0) return 0;
    return dlsa::dense_pass_bytes(n, p);
}
int dlsa_gram_icpt_f64(const double* X, int64_t ldx, const double* w, int64_t n, int p, double* H, int64_t ldh,
                       void* ws, size_t ws_bytes, void* stream) {
    using namespace dlsa;
    DLSA_REQUIRE((X || n == 0) && H && p > 0 && p <= 2047 && n >= 0 && ldx >= p && ldh >= p + 1, "gram_icpt: bad argument");
    if (!ws || ws_bytes < dense_pass_bytes(n, p) || ((uintptr_t)ws & 255)) {
        set_error("gram_icpt: workspace %zu bytes needed (256-aligned), got %zu", dense_pass_bytes(n, p), ws_bytes);
        return DLSA_ERR_WORKSPACE;
    }
    return gram_icpt_impl(X, ldx, w, n, p, H, ldh, ws, ws_bytes, (hipStream_t)stream);
}

// One-hot designs: the same fit on raw numerics + level codes (onehot.hip), never materialising the dense matrix.
size_t dlsa_onehot_irls_workspace_bytes(const dlsa_onehot_plan* plan, int64_t max_rows_per_partition) {
    if (!plan || max_rows_per_partition < 0) return 0;
    return dlsa::align_up(dlsa::irls_layout(max_rows_per_partition, dlsa::onehot_plan_p(plan),
                                            dlsa::onehot_workspace_bytes_impl(plan, max_rows_per_partition)).total, 256) *
           (size_t)dlsa::irls_chain_cap(max_rows_per_partition, 128.0);       // (raw rows: a few numerics + level codes)
}

int dlsa_onehot_irls_fit_f64(const dlsa_onehot_plan* plan, const double* num, int64_t ldn, const int32_t* codes, int64_t ldc,
                             const double* y, const int64_t* part_offsets_host, int K, double tol, int max_iter,
                             double* coef, double* Sig_inv, double* Sig_invMcoef, int* n_iter_host, int* status_host,
                             double* loglik_host, void* ws, size_t ws_bytes, void* stream) {
    using namespace dlsa;
    DLSA_REQUIRE(plan && y && part_offsets_host && coef && Sig_inv && Sig_invMcoef, "onehot irls_fit: null argument");
    DLSA_REQUIRE(K > 0 && max_iter > 0 && tol > 0, "onehot irls_fit: bad K/tol/max_iter");
    const int p = onehot_plan_p(plan);
    auto make_data = [=](int, int64_t r0, char*, hipStream_t) {
        const double* numk = num ? num + r0 * ldn : nullptr;
        const int32_t* codesk = codes ? codes + r0 * ldc : nullptr;
        const double* yk = y + r0;
        IrlsData d;
        d.logit = [=](const double* beta, int64_t nrows, double* w, double* g, double* ll, const IrlsBuffers& b, hipStream_t s) {
            return onehot_logit_pass_impl(plan, numk, ldn, codesk, ldc, yk, beta, nrows, w, g, ll, b.ws_pass, b.ws_pass_bytes, s);
        };
        d.gram = [=](const double* w, int64_t nrows, double* H, const IrlsBuffers& b, hipStream_t s) {
            return onehot_gram_impl(plan, numk, ldn, codesk, ldc, w, nrows, H, p, b.ws_pass, b.ws_pass_bytes, s, true);
        };
        // (no own-Hessian steps here: measured on config 4's structured shard they halve the iterations -- 10 -> 5 per partition, exact
        // quadratic convergence -- but two structured Grams per partition cost what the five saved launch-bound iterations did: 13.7 -> 15.2 ms)
        return d;
    };
    int64_t max_rows = 0;
    for (int k = 0; k < K; ++k) max_rows = std::max(max_rows, part_offsets_host[k + 1] - part_offsets_host[k]);
    return irls_fit_core(make_data, [=](int64_t rows) { return onehot_workspace_bytes_impl(plan, rows); }, 0,
                         irls_chain_cap(max_rows, 128.0), irls_chain_seed(max_rows, 128.0), part_offsets_host, K, p, tol, max_iter, coef, Sig_inv, Sig_invMcoef,
                         n_iter_host, status_host, loglik_host, ws, ws_bytes, stream);
}

// The same for partitions given as (first row, rows, common row step): partition_id = i % K (models.py:33) is first = 0..K-1,
// step = K -- strided VIEWS of the raw numerics and the level codes (row pitches ldn * step, ldc * step), nothing gathered
// but the partition's labels (8 bytes per row).
size_t dlsa_onehot_irls_ex_workspace_bytes(const dlsa_onehot_plan* plan, int64_t max_rows_per_partition, int64_t row_step) {
    if (!plan || max_rows_per_partition < 0 || row_step < 1) return 0;
    const size_t ybuf = row_step > 1 ? dlsa::align_up((size_t)std::max<int64_t>(max_rows_per_partition, 1) * sizeof(double), 256) : 0;
    return ybuf * (size_t)dlsa::irls_chain_cap(max_rows_per_partition, 128.0) + dlsa_onehot_irls_workspace_bytes(plan, max_rows_per_partition);
}

int dlsa_onehot_irls_fit_ex_f64(const dlsa_onehot_plan* plan, const double* num, int64_t ldn, const int32_t* codes, int64_t ldc,
                                const double* y, const int64_t* part_first_host, const int64_t* part_rows_host, int64_t row_step,
                                int K, double tol, int max_iter, double* coef, double* Sig_inv, double* Sig_invMcoef,
                                int* n_iter_host, int* status_host, double* loglik_host, void* ws, size_t ws_bytes, void* stream) {
    using namespace dlsa;
    DLSA_REQUIRE(plan && y && part_first_host && part_rows_host && coef && Sig_inv && Sig_invMcoef, "onehot irls_fit_ex: null argument");
    DLSA_REQUIRE(K > 0 && max_iter > 0 && tol > 0 && row_step >= 1, "onehot irls_fit_ex: bad K/tol/max_iter/row_step");
    const int p = onehot_plan_p(plan);
    int64_t max_rows = 0;
    std::vector<int64_t> offs((size_t)K + 1, 0);
    for (int k = 0; k < K; ++k) {
        DLSA_REQUIRE(part_rows_host[k] >= 0 && part_first_host[k] >= 0, "onehot irls_fit_ex: negative partition shape");
        max_rows = std::max(max_rows, part_rows_host[k]);
        offs[(size_t)k + 1] = offs[(size_t)k] + part_rows_host[k];
    }
    const size_t need = dlsa_onehot_irls_ex_workspace_bytes(plan, max_rows, row_step);
    if (!ws || ws_bytes < need || ((uintptr_t)ws & 255)) {
        set_error("onehot irls_fit_ex: workspace %zu bytes needed (256-aligned), got %zu", need, ws_bytes);
        return DLSA_ERR_WORKSPACE;
    }
    const size_t ybytes = row_step > 1 ? align_up((size_t)std::max<int64_t>(max_rows, 1) * sizeof(double), 256) : 0;
    const int64_t pn = ldn * row_step, pc = ldc * row_step;
    auto make_data = [=](int k, int64_t, char* extra, hipStream_t st) {
        const double* numk = num ? num + part_first_host[k] * ldn : nullptr;
        const int32_t* codesk = codes ? codes + part_first_host[k] * ldc : nullptr;
        const double* yk = y + part_first_host[k];
        const int64_t nk = part_rows_host[k];
        if (row_step > 1 && nk > 0) {
            double* ybuf = (double*)extra;
            hipLaunchKernelGGL(gather_strided_kernel, dim3((unsigned)((nk + 255) / 256)), dim3(256), 0, st, y, part_first_host[k],
                               row_step, nk, ybuf);
            yk = ybuf;
        }
        IrlsData d;
        d.logit = [=](const double* beta, int64_t nrows, double* w, double* g, double* ll, const IrlsBuffers& b, hipStream_t s) {
            return onehot_logit_pass_impl(plan, numk, pn, codesk, pc, yk, beta, nrows, w, g, ll, b.ws_pass, b.ws_pass_bytes, s);
        };
        d.gram = [=](const double* w, int64_t nrows, double* H, const IrlsBuffers& b, hipStream_t s) {
            return onehot_gram_impl(plan, numk, pn, codesk, pc, w, nrows, H, p, b.ws_pass, b.ws_pass_bytes, s, true);
        };
        return d;
    };
    return irls_fit_core(make_data, [=](int64_t rows) { return onehot_workspace_bytes_impl(plan, rows); }, ybytes,
                         irls_chain_cap(max_rows, 128.0), irls_chain_seed(max_rows, 128.0), offs.data(), K, p, tol, max_iter, coef, Sig_inv, Sig_invMcoef, n_iter_host,
                         status_host, loglik_host, ws, ws_bytes, stream);
}

}  // extern "C"
