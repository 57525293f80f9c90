// Small dense fp64 pieces that sit between the big passes: the p x p Cholesky solve used by
// every Newton step (replacing the inner solver of sklearn's newton-cg, dlsa/models.py:113)
// and by the WLS combine (dlsa/dlsa.py:48-49), the Sig_inv . coef product (models.py:131) and
// the local block sum that feeds the one-round all-reduce (dlsa.py:30-34).
#include "common.h"

namespace dlsa {

constexpr int CHOL_THREADS = 1024;

// One workgroup factors one p x p SPD system in place in `L` (lower triangle, row-major, pitch p)
// and solves L L' x = rhs.  Right-looking; the current column is cached in LDS so the trailing
// update reads L row-wise (coalesced).  stats: [0] max|x|, [1] max|ref| (ref nullable), [2] info
// (0 ok, 1 not SPD, 2 NaN/Inf).
__global__ __launch_bounds__(CHOL_THREADS) void chol_solve_kernel(
        const double* __restrict__ A, int64_t lda, int64_t strideA,
        const double* __restrict__ rhs, int64_t stride_rhs,
        const double* __restrict__ ref, int64_t stride_ref,
        int p, double* __restrict__ Lws, double* __restrict__ xout, int64_t stride_x,
        double* __restrict__ stats, int64_t stride_stats, int reuse_factor) {
    // all LDS lives in the one dynamic array (keeps its base 16-byte aligned)
    extern __shared__ __attribute__((aligned(16))) double sm[];
    double* col = sm;                    // p
    double* vec = sm + p;                // p
    double* redx = sm + 2 * p;           // CHOL_THREADS/64
    double* redr = redx + CHOL_THREADS / 64;
    double& pivot = redr[CHOL_THREADS / 64];
    int& info = *reinterpret_cast<int*>(&redr[CHOL_THREADS / 64 + 1]);
    int& redbad = *reinterpret_cast<int*>(&redr[CHOL_THREADS / 64 + 2]);
    const int sys = blockIdx.x;
    A += sys * strideA;
    rhs += sys * stride_rhs;
    double* L = Lws + (int64_t)sys * p * p;
    xout += sys * stride_x;
    stats += sys * stride_stats;
    const int tid = threadIdx.x;
    const int nth = blockDim.x;

    if (tid == 0) info = 0;
    if (!reuse_factor) {
        // copy the lower triangle (A is symmetric; read the stored row i, columns <= i)
        for (int64_t e = tid; e < (int64_t)p * p; e += nth) {
            const int i = (int)(e / p), k = (int)(e % p);
            L[e] = (k <= i) ? A[(int64_t)i * lda + k] : 0.0;
        }
    }
    for (int i = tid; i < p; i += nth) vec[i] = rhs[i];
    __syncthreads();

    // reuse_factor != 0: L already holds the factor of an earlier (frozen) Hessian -- solves only
    for (int j = 0; j < (reuse_factor ? 0 : p); ++j) {
        if (tid == 0) {
            const double d = L[(int64_t)j * p + j];
            if (!(d > 0.0) || !isfinite(d)) { info = isfinite(d) ? 1 : 2; pivot = 1.0; }
            else pivot = sqrt(d);
        }
        __syncthreads();
        const double s = pivot;
        for (int i = j + tid; i < p; i += nth) {
            const double v = (i == j) ? s : L[(int64_t)i * p + j] / s;
            L[(int64_t)i * p + j] = v;
            col[i] = v;
        }
        __syncthreads();
        // trailing update of rows i > j, columns j < k <= i
        const int m = p - j - 1;
        if (m > 0) {
            // flatten the (i,k) lower-triangular index space row by row in chunks of 64 columns
            for (int i = j + 1 + (tid >> 6); i < p; i += (nth >> 6)) {
                const double li = col[i];
                double* row = L + (int64_t)i * p;
                for (int k = j + 1 + (tid & 63); k <= i; k += 64) row[k] -= li * col[k];
            }
        }
        __syncthreads();
    }
    // forward solve L z = rhs
    for (int j = 0; j < p; ++j) {
        if (tid == 0) vec[j] = vec[j] / L[(int64_t)j * p + j];
        __syncthreads();
        const double zj = vec[j];
        for (int i = j + 1 + tid; i < p; i += nth) vec[i] -= L[(int64_t)i * p + j] * zj;
        __syncthreads();
    }
    // backward solve L' x = z   (column j of L' is row j of L: contiguous)
    for (int j = p - 1; j >= 0; --j) {
        if (tid == 0) vec[j] = vec[j] / L[(int64_t)j * p + j];
        __syncthreads();
        const double xj = vec[j];
        const double* row = L + (int64_t)j * p;
        for (int i = tid; i < j; i += nth) vec[i] -= row[i] * xj;
        __syncthreads();
    }
    // outputs + stats
    double mx = 0.0, mr = 0.0;
    bool bad = false;
    for (int i = tid; i < p; i += nth) {
        const double v = vec[i];
        xout[i] = v;
        mx = fmax(mx, fabs(v));
        if (!isfinite(v)) bad = true;
        if (ref) mr = fmax(mr, fabs(ref[sys * stride_ref + i]));
    }
    if (tid == 0) redbad = 0;
    __syncthreads();
    for (int m2 = 32; m2 >= 1; m2 >>= 1) {
        mx = fmax(mx, __shfl_xor(mx, m2, 64));
        mr = fmax(mr, __shfl_xor(mr, m2, 64));
    }
    if ((tid & 63) == 0) { redx[tid >> 6] = mx; redr[tid >> 6] = mr; }
    if (bad) redbad = 1;
    __syncthreads();
    if (tid == 0) {
        double a = 0.0, b = 0.0;
        for (int k = 0; k < nth / 64; ++k) { a = fmax(a, redx[k]); b = fmax(b, redr[k]); }
        stats[0] = a;
        stats[1] = b;
        stats[2] = (double)(info ? info : (redbad ? 2 : 0));
    }
}

// y = A x   (A p x p row-major, one wave per row)
__global__ void matvec_kernel(const double* __restrict__ A, int64_t lda, const double* __restrict__ x,
                              int p, double* __restrict__ y) {
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= p) return;
    double s = 0.0;
    for (int k = lane; k < p; k += 64) s = fma(A[(int64_t)row * lda + k], x[k], s);
    for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m, 64);
    if (lane == 0) y[row] = s;
}

// out = a + s * b
__global__ void axpby_kernel(const double* a, const double* b, double s,
                             int n, double* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = a[i] + s * b[i];
}

// out[e] = sum over included k of in[k*stride + e]   (fixed order)
__global__ void sum_blocks_kernel(const double* __restrict__ in, int64_t stride, int K,
                                  const int* __restrict__ mask, int64_t count, double* __restrict__ out) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= count) return;
    double s = 0.0;
    for (int k = 0; k < K; ++k)
        if (!mask || mask[k] == 0) s += in[k * stride + e];
    out[e] = s;
}

int launch_chol_solve(const double* A, int64_t lda, int64_t strideA, const double* rhs, int64_t stride_rhs,
                      const double* ref, int64_t stride_ref, int p, int nsys, double* Lws, double* xout,
                      int64_t stride_x, double* stats, int64_t stride_stats, hipStream_t s, int reuse_factor) {
    const size_t shm = ((size_t)2 * p + 2 * (CHOL_THREADS / 64) + 4) * sizeof(double);
    DLSA_REQUIRE(shm <= 60 * 1024, "spd solve: p=%d too large for the single-workgroup solver", p);
    int threads = CHOL_THREADS;
    if (p <= 64) threads = 256;
    hipLaunchKernelGGL(chol_solve_kernel, dim3(nsys), dim3(threads), shm, s, A, lda, strideA, rhs, stride_rhs,
                       ref, stride_ref, p, Lws, xout, stride_x, stats, stride_stats, reuse_factor);
    DLSA_HIP_CHECK(hipGetLastError());
    return DLSA_OK;
}

int launch_matvec(const double* A, int64_t lda, const double* x, int p, double* y, hipStream_t s) {
    hipLaunchKernelGGL(matvec_kernel, dim3((p + 3) / 4), dim3(256), 0, s, A, lda, x, p, y);
    DLSA_HIP_CHECK(hipGetLastError());
    return DLSA_OK;
}

int launch_axpby(const double* a, const double* b, double sc, int n, double* out, hipStream_t s) {
    hipLaunchKernelGGL(axpby_kernel, dim3((n + 255) / 256), dim3(256), 0, s, a, b, sc, n, out);
    DLSA_HIP_CHECK(hipGetLastError());
    return DLSA_OK;
}

}  // namespace dlsa

extern "C" {

size_t dlsa_solve_workspace_bytes(int p) {
    if (p <= 0) return 0;
    return dlsa::align_up((size_t)p * p * sizeof(double), 256) + 256 + 256;
}

int dlsa_spd_solve_f64(const double* S, int64_t lds, const double* v, int p, double* theta,
                       void* ws, size_t ws_bytes, void* stream) {
    using namespace dlsa;
    DLSA_REQUIRE(S && v && theta, "spd_solve: null argument");
    DLSA_REQUIRE(p > 0 && lds >= p, "spd_solve: bad shape p=%d lds=%lld", p, (long long)lds);
    if (!ws || ws_bytes < dlsa_solve_workspace_bytes(p) || ((uintptr_t)ws & 255)) {
        set_error("spd_solve: workspace %zu bytes needed (256-aligned), got %zu", dlsa_solve_workspace_bytes(p), ws_bytes);
        return DLSA_ERR_WORKSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    Arena ar(ws, ws_bytes);
    double* L = (double*)ar.take((size_t)p * p * sizeof(double));
    double* stats = (double*)ar.take(4 * sizeof(double));
    int rc = launch_chol_solve(S, lds, 0, v, 0, nullptr, 0, p, 1, L, theta, 0, stats, 0, s, 0);
    if (rc) return rc;
    double h[3];
    DLSA_HIP_CHECK(hipMemcpyAsync(h, stats, sizeof(h), hipMemcpyDeviceToHost, s));
    DLSA_HIP_CHECK(hipStreamSynchronize(s));
    if (h[2] == 1.0) { set_error("spd_solve: matrix is not positive definite"); return DLSA_ERR_NOT_SPD; }
    if (h[2] == 2.0) { set_error("spd_solve: NaN/Inf in the system"); return DLSA_ERR_NAN; }
    return DLSA_OK;
}

int dlsa_sum_blocks_f64(const double* coef, const double* Sig_inv, const double* Sig_invMcoef, int K, int p,
                        const int* mask_host, double* out, void* stream) {
    using namespace dlsa;
    DLSA_REQUIRE(coef && Sig_inv && Sig_invMcoef && out, "sum_blocks: null argument");
    DLSA_REQUIRE(K > 0 && p > 0, "sum_blocks: bad K=%d p=%d", K, p);
    hipStream_t s = (hipStream_t)stream;
    int* dmask = nullptr;
    if (mask_host) {
        bool any = false;
        for (int k = 0; k < K; ++k) any |= (mask_host[k] != 0);
        if (any) {
            // the mask rides in the tail of `out`'s own message?  no: keep it simple and exact --
            // a K-int scratch is allocated stream-ordered and released before return.
            DLSA_HIP_CHECK(hipMallocAsync((void**)&dmask, (size_t)K * sizeof(int), s));
            DLSA_HIP_CHECK(hipMemcpyAsync(dmask, mask_host, (size_t)K * sizeof(int), hipMemcpyHostToDevice, s));
        }
    }
    const int64_t pp = (int64_t)p * p;
    hipLaunchKernelGGL(sum_blocks_kernel, dim3((unsigned)((pp + 255) / 256)), dim3(256), 0, s, Sig_inv, pp, K, dmask, pp, out);
    hipLaunchKernelGGL(sum_blocks_kernel, dim3((p + 255) / 256), dim3(256), 0, s, Sig_invMcoef, (int64_t)p, K, dmask, (int64_t)p, out + pp);
    hipLaunchKernelGGL(sum_blocks_kernel, dim3((p + 255) / 256), dim3(256), 0, s, coef, (int64_t)p, K, dmask, (int64_t)p, out + pp + p);
    DLSA_HIP_CHECK(hipGetLastError());
    if (dmask) {
        DLSA_HIP_CHECK(hipStreamSynchronize(s));   // mask_host must stay valid until the copy is done
        DLSA_HIP_CHECK(hipFreeAsync(dmask, s));
    }
    return DLSA_OK;
}

}  // extern "C"
