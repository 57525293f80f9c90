// Dense design matrix of one chunk, built on the device from raw numeric columns and integer level codes
// (reference: dlsa/models.py:56-104 -- pd.get_dummies, drop of the baseline levels, standardisation with the
// global mean/std, reindex to the canonical column order -- and the leading ones column of :121-122).
//
// HBM-bound scatter: n*p*sizeof(T) bytes written once, n*(q*sizeof(T) + 4f) read.  A workgroup owns DROWS
// consecutive rows; thread t owns output columns t, t+256, ... and keeps their descriptors in registers, so
// every store instruction of a wave covers 64 consecutive elements of one row and the code / numeric reads
// are L1 broadcasts.  `seen[j]` records whether column j got a non-zero entry (the reference's
// "dummy level missing in this chunk" test, models.py:80-91).
#include "common.h"
#include <algorithm>

namespace dlsa {

constexpr int DTHREADS = 256;
constexpr int DROWS_MAX = 64;       // rows per workgroup pass (fewer when the raw rows are wide)
constexpr int DMAXP = 2048;
constexpr int DLDS_BYTES = 48 * 1024;

// The raw rows of a pass (numeric values as fp64, then the codes) are staged in LDS with coalesced loads, so the
// per-element work below is LDS reads + one streaming store, and the stores of several rows are in flight at once.
template <typename T>
__global__ __launch_bounds__(DTHREADS) void design_kernel(const T* __restrict__ num, int64_t ldn, int q,
                                                          const int32_t* __restrict__ codes, int64_t ldc, int f, int64_t n,
                                                          const int32_t* __restrict__ kind, const int32_t* __restrict__ src,
                                                          const int32_t* __restrict__ level, const double* __restrict__ shift,
                                                          const double* __restrict__ scale, int p, int drows,
                                                          T* __restrict__ X, int64_t ldx, int32_t* __restrict__ seen) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double* snum = reinterpret_cast<double*>(smem);                       // [drows][q]
    int32_t* scode = reinterpret_cast<int32_t*>(smem + (size_t)drows * q * sizeof(double));   // [drows][f]
    // p <= 256: several rows per pass (column stride = next power of two >= p); wider: one row per pass
    int cstride = DTHREADS;
    if (p <= DTHREADS) { cstride = 1; while (cstride < p) cstride <<= 1; }
    const int rlanes = DTHREADS / cstride;
    const int cbase = threadIdx.x % cstride, rlane = threadIdx.x / cstride;
    for (int64_t r0 = (int64_t)blockIdx.x * drows; r0 < n; r0 += (int64_t)gridDim.x * drows) {
        const int nr = (int)min((int64_t)drows, n - r0);
        __syncthreads();                                  // the previous pass is done with the staging buffers
        for (int e = threadIdx.x; e < nr * q; e += DTHREADS) snum[e] = (double)num[(r0 + e / q) * ldn + e % q];
        for (int e = threadIdx.x; e < nr * f; e += DTHREADS) scode[e] = codes[(r0 + e / f) * ldc + e % f];
        __syncthreads();
        // column-outer / row-inner: one descriptor live at a time (few VGPRs -> many waves to hide the store
        // latency); the kind test is hoisted, so only the waves that own numeric columns run the division loop
        for (int j = cbase; j < p; j += cstride) {
            const int kd = kind[j], sj = src[j];
            T* __restrict__ dst = X + r0 * ldx + j;
            bool any = false;
            if (kd == 2) {
                const int lv = level[j];
#pragma unroll 8
                for (int i = rlane; i < nr; i += rlanes) {
                    const bool hit = scode[i * f + sj] == lv;
                    any |= hit;
                    dst[(int64_t)i * ldx] = hit ? T(1) : T(0);
                }
            } else if (kd == 1) {
                const double sh = shift[j], sc = scale[j];
#pragma unroll 4
                for (int i = rlane; i < nr; i += rlanes) {
                    const double v = (snum[i * q + sj] - sh) / sc;
                    any |= (v != 0.0);
                    dst[(int64_t)i * ldx] = (T)v;
                }
            } else {
                any = nr > 0;
#pragma unroll 8
                for (int i = rlane; i < nr; i += rlanes) dst[(int64_t)i * ldx] = T(1);
            }
            if (seen && any) seen[j] = 1;                 // benign race: every writer stores 1
            if (cstride != DTHREADS) break;               // narrow p: one column per thread
        }
    }
}

template <typename T>
static int design_impl(const T* num, int64_t ldn, int q, const int32_t* codes, int64_t ldc, int f, int64_t n,
                       const int32_t* kind, const int32_t* src, const int32_t* level, const double* shift,
                       const double* scale, int p, T* X, int64_t ldx, int32_t* seen, hipStream_t stream) {
    DLSA_REQUIRE(X && kind && src && level && shift && scale, "design: null output or descriptor");
    DLSA_REQUIRE(p > 0 && p <= DMAXP && n >= 0 && ldx >= p, "design: bad shape n=%lld p=%d ldx=%lld (p <= %d)",
                 (long long)n, p, (long long)ldx, DMAXP);
    DLSA_REQUIRE(q >= 0 && f >= 0 && (q == 0 || (num && ldn >= q)) && (f == 0 || (codes && ldc >= f)),
                 "design: bad inputs q=%d f=%d ldn=%lld ldc=%lld", q, f, (long long)ldn, (long long)ldc);
    const size_t row_bytes = (size_t)q * sizeof(double) + (size_t)f * sizeof(int32_t);
    DLSA_REQUIRE(row_bytes <= (size_t)DLDS_BYTES, "design: %d numeric + %d factor columns exceed the staging buffer", q, f);
    if (seen) DLSA_HIP_CHECK(hipMemsetAsync(seen, 0, (size_t)p * sizeof(int32_t), stream));
    if (n == 0) return DLSA_OK;
    int drows = DROWS_MAX;
    if (row_bytes > 0) drows = (int)std::max<size_t>(1, std::min<size_t>(DROWS_MAX, (size_t)DLDS_BYTES / row_bytes));
    const size_t lds = align_up((size_t)drows * q * sizeof(double), 16) + (size_t)drows * f * sizeof(int32_t);
    const int64_t want = (n + drows - 1) / drows;
    const int blocks = (int)std::min<int64_t>(want, (int64_t)kNumCU * 16);
    hipLaunchKernelGGL((design_kernel<T>), dim3(blocks), dim3(DTHREADS), lds, stream, num, ldn, q, codes, ldc, f, n,
                       kind, src, level, shift, scale, p, drows, X, ldx, seen);
    DLSA_HIP_CHECK(hipGetLastError());
    return DLSA_OK;
}

}  // namespace dlsa

extern "C" {

int dlsa_design_f64(const double* num, int64_t ldn, int q, const int32_t* codes, int64_t ldc, int f, int64_t n,
                    const int32_t* kind, const int32_t* src, const int32_t* level, const double* shift,
                    const double* scale, int p, double* X, int64_t ldx, int32_t* seen, void* stream) {
    return dlsa::design_impl<double>(num, ldn, q, codes, ldc, f, n, kind, src, level, shift, scale, p, X, ldx, seen,
                                     (hipStream_t)stream);
}

int dlsa_design_f32(const float* num, int64_t ldn, int q, const int32_t* codes, int64_t ldc, int f, int64_t n,
                    const int32_t* kind, const int32_t* src, const int32_t* level, const double* shift,
                    const double* scale, int p, float* X, int64_t ldx, int32_t* seen, void* stream) {
    return dlsa::design_impl<float>(num, ldn, q, codes, ldc, f, n, kind, src, level, shift, scale, p, X, ldx, seen,
                                    (hipStream_t)stream);
}

}  // extern "C"
