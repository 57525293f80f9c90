// Per-partition exact-MLE logistic fit + local quadratic approximation: the numeric core of
// logistic_model (dlsa/models.py:110-131) for K row-partitions that sit contiguously in HBM.
//
// Per partition:  beta <- 0;  repeat { logit pass (w, g, loglik);  Gram pass H = X'WX;
// Newton step H delta = g by Cholesky on the device;  stop when |delta|_inf <= tol*max(1,|beta|_inf) }.
// The Hessian of the last iteration is written straight into Sig_inv[k], so it is evaluated at
// the returned coef exactly as the reference evaluates it after the fit (models.py:114,130);
// Sig_invMcoef = Sig_inv . coef (models.py:131).  One small D2H copy of 4 doubles per iteration
// is the only host synchronisation.
#include "common.h"
#include <math.h>
#include <algorithm>

namespace dlsa {
int gram_impl_f64(const double* X, int64_t ldx, const double* w, int64_t n, int p, double* H, int64_t ldh,
                  int accumulate, void* ws, size_t ws_bytes, hipStream_t stream);
size_t gram_workspace_bytes_impl(int64_t n, int p, int elem_bytes);
size_t logit_workspace_bytes_impl(int64_t n, int p);
int logit_pass_impl(const double* X, int64_t ldx, const double* y, const double* beta, int64_t n, int p,
                    double* w_out, double* g, double* loglik, void* ws, size_t ws_bytes, hipStream_t stream);
int launch_chol_solve(const double* A, int64_t lda, int64_t strideA, const double* rhs, int64_t stride_rhs,
                      const double* ref, int64_t stride_ref, int p, int nsys, double* Lws, double* xout,
                      int64_t stride_x, double* stats, int64_t stride_stats, hipStream_t s);
int launch_matvec(const double* A, int64_t lda, const double* x, int p, double* y, hipStream_t s);
int launch_axpby(const double* a, const double* b, double sc, int n, double* out, hipStream_t s);

struct IrlsLayout {
    size_t w, g, beta, beta_prev, delta, stats, L, gram, logit, total;
};

static IrlsLayout irls_layout(int64_t max_rows, int p) {
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
    l.gram = take(gram_workspace_bytes_impl(max_rows, p, 8));
    l.logit = take(logit_workspace_bytes_impl(max_rows, p));
    l.total = align_up(off, 256);
    return l;
}

}  // namespace dlsa

extern "C" {

size_t dlsa_irls_workspace_bytes(int64_t max_rows_per_partition, int p) {
    if (p <= 0 || p > 2048 || max_rows_per_partition < 0) return 0;
    return dlsa::irls_layout(max_rows_per_partition, p).total;
}

int dlsa_irls_fit_f64(const double* X, int64_t ldx, const double* y, const int64_t* part_offsets_host,
                      int K, int p, double tol, int max_iter, double* coef, double* Sig_inv,
                      double* Sig_invMcoef, int* n_iter_host, int* status_host, double* loglik_host,
                      void* ws, size_t ws_bytes, void* stream) {
    using namespace dlsa;
    DLSA_REQUIRE(X && y && part_offsets_host && coef && Sig_inv && Sig_invMcoef, "irls_fit: null argument");
    DLSA_REQUIRE(K > 0 && p > 0 && p <= 2048 && ldx >= p, "irls_fit: bad shape K=%d p=%d ldx=%lld", K, p, (long long)ldx);
    DLSA_REQUIRE(max_iter > 0 && tol > 0, "irls_fit: bad tol/max_iter");
    int64_t max_rows = 0;
    for (int k = 0; k < K; ++k) {
        DLSA_REQUIRE(part_offsets_host[k + 1] >= part_offsets_host[k], "irls_fit: part_offsets not monotone");
        max_rows = std::max(max_rows, part_offsets_host[k + 1] - part_offsets_host[k]);
    }
    const IrlsLayout l = irls_layout(max_rows, p);
    if (!ws || ws_bytes < l.total || ((uintptr_t)ws & 255)) {
        set_error("irls_fit: workspace %zu bytes needed (256-aligned), got %zu", l.total, ws_bytes);
        return DLSA_ERR_WORKSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    char* base = (char*)ws;
    double* d_w = (double*)(base + l.w);
    double* d_g = (double*)(base + l.g);
    double* d_beta = (double*)(base + l.beta);
    double* d_prev = (double*)(base + l.beta_prev);
    double* d_delta = (double*)(base + l.delta);
    double* d_stats = (double*)(base + l.stats);   // [0..2] solver stats, [3] loglik
    double* d_L = (double*)(base + l.L);
    void* ws_gram = base + l.gram;
    const size_t ws_gram_bytes = gram_workspace_bytes_impl(max_rows, p, 8);
    void* ws_logit = base + l.logit;
    const size_t ws_logit_bytes = logit_workspace_bytes_impl(max_rows, p);

    int overall = DLSA_OK;
    for (int k = 0; k < K; ++k) {
        const int64_t r0 = part_offsets_host[k];
        const int64_t nk = part_offsets_host[k + 1] - r0;
        const double* Xk = X + r0 * ldx;
        const double* yk = y + r0;
        double* Hk = Sig_inv + (int64_t)k * p * p;
        double* ck = coef + (int64_t)k * p;
        double* sk = Sig_invMcoef + (int64_t)k * p;
        int st = DLSA_PART_OK, iters = 0;
        double ll = 0.0;
        if (nk == 0) {
            // empty partition: the reference's zero block (models.py:84-91)
            DLSA_HIP_CHECK(hipMemsetAsync(Hk, 0, (size_t)p * p * sizeof(double), s));
            DLSA_HIP_CHECK(hipMemsetAsync(ck, 0, (size_t)p * sizeof(double), s));
            DLSA_HIP_CHECK(hipMemsetAsync(sk, 0, (size_t)p * sizeof(double), s));
            st = DLSA_PART_EMPTY;
        } else {
            DLSA_HIP_CHECK(hipMemsetAsync(d_beta, 0, (size_t)p * sizeof(double), s));
            double ll_prev = -INFINITY;
            bool have_prev = false;
            st = DLSA_PART_NOT_CONVERGED;
            int halvings = 0;
            for (int it = 1; it <= max_iter; ++it) {
                iters = it;
                int rc = logit_pass_impl(Xk, ldx, yk, d_beta, nk, p, d_w, d_g, d_stats + 3, ws_logit, ws_logit_bytes, s);
                if (rc) return rc;
                rc = gram_impl_f64(Xk, ldx, d_w, nk, p, Hk, p, 0, ws_gram, ws_gram_bytes, s);
                if (rc) return rc;
                rc = launch_chol_solve(Hk, p, 0, d_g, 0, d_beta, 0, p, 1, d_L, d_delta, 0, d_stats, 0, s);
                if (rc) return rc;
                double h[4];
                DLSA_HIP_CHECK(hipMemcpyAsync(h, d_stats, sizeof(h), hipMemcpyDeviceToHost, s));
                DLSA_HIP_CHECK(hipStreamSynchronize(s));
                ll = h[3];
                if (h[2] == 1.0) { st = DLSA_PART_NOT_SPD; break; }
                if (h[2] == 2.0 || !isfinite(ll)) { st = DLSA_PART_NAN; break; }
                // step-halving safeguard: the previous full step overshot (log-likelihood dropped)
                if (have_prev && ll < ll_prev - 1e-12 * fabs(ll_prev) && halvings < 30) {
                    ++halvings;
                    // beta <- beta_prev + (beta - beta_prev)/2
                    rc = launch_axpby(d_beta, d_prev, -1.0, p, d_delta, s);
                    if (rc) return rc;
                    rc = launch_axpby(d_prev, d_delta, 0.5, p, d_beta, s);
                    if (rc) return rc;
                    continue;
                }
                halvings = 0;
                if (h[0] <= tol * std::max(1.0, h[1])) { st = DLSA_PART_OK; break; }
                DLSA_HIP_CHECK(hipMemcpyAsync(d_prev, d_beta, (size_t)p * sizeof(double), hipMemcpyDeviceToDevice, s));
                rc = launch_axpby(d_beta, d_delta, 1.0, p, d_beta, s);
                if (rc) return rc;
                ll_prev = ll;
                have_prev = true;
            }
            DLSA_HIP_CHECK(hipMemcpyAsync(ck, d_beta, (size_t)p * sizeof(double), hipMemcpyDeviceToDevice, s));
            int rc = launch_matvec(Hk, p, d_beta, p, sk, s);
            if (rc) return rc;
        }
        if (n_iter_host) n_iter_host[k] = iters;
        if (status_host) status_host[k] = st;
        if (loglik_host) loglik_host[k] = ll;
        if (st == DLSA_PART_NOT_CONVERGED && overall == DLSA_OK) overall = DLSA_ERR_NOT_CONVERGED;
        if (st == DLSA_PART_NOT_SPD && overall == DLSA_OK) overall = DLSA_ERR_NOT_SPD;
        if (st == DLSA_PART_NAN && overall == DLSA_OK) overall = DLSA_ERR_NAN;
    }
    DLSA_HIP_CHECK(hipStreamSynchronize(s));
    if (overall == DLSA_ERR_NOT_CONVERGED) set_error("irls_fit: at least one partition hit max_iter");
    if (overall == DLSA_ERR_NOT_SPD) set_error("irls_fit: a partition's Hessian is not positive definite");
    if (overall == DLSA_ERR_NAN) set_error("irls_fit: NaN/Inf in a partition's fit");
    return overall;
}

}  // extern "C"
