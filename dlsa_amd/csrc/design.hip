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
constexpr int DROWS = 64;
constexpr int DMAXC = 8;            // columns per thread: p <= 2048

template <typename T>
__global__ __launch_bounds__(DTHREADS) void design_kernel(const T* __restrict__ num, int64_t ldn,
                                                          const int32_t* __restrict__ codes, int64_t ldc, int64_t n,
                                                          const int32_t* __restrict__ kind, const int32_t* __restrict__ src,
                                                          const int32_t* __restrict__ level, const double* __restrict__ shift,
                                                          const double* __restrict__ scale, int p,
                                                          T* __restrict__ X, int64_t ldx, int32_t* __restrict__ seen) {
    // p <= 256: several rows per pass (column stride = next power of two >= p); wider: one row per pass
    int cstride = DTHREADS;
    if (p <= DTHREADS) { cstride = 1; while (cstride < p) cstride <<= 1; }
    const int rlanes = DTHREADS / cstride;
    const int cbase = threadIdx.x % cstride, rlane = threadIdx.x / cstride;
    int ck[DMAXC], cs[DMAXC], cl[DMAXC];
    double sh[DMAXC], sc[DMAXC];
    bool any[DMAXC];
#pragma unroll
    for (int c = 0; c < DMAXC; ++c) {
        const int j = cbase + c * cstride;
        ck[c] = -1; cs[c] = 0; cl[c] = 0; sh[c] = 0.0; sc[c] = 1.0; any[c] = false;
        if (j < p && (c == 0 || cstride == DTHREADS)) { ck[c] = kind[j]; cs[c] = src[j]; cl[c] = level[j]; sh[c] = shift[j]; sc[c] = scale[j]; }
    }
    for (int64_t r0 = (int64_t)blockIdx.x * DROWS; r0 < n; r0 += (int64_t)gridDim.x * DROWS) {
        const int64_t r1 = min(r0 + DROWS, n);
        for (int64_t i = r0 + rlane; i < r1; i += rlanes) {
#pragma unroll
            for (int c = 0; c < DMAXC; ++c) {
                if (ck[c] < 0) continue;
                double v;
                if (ck[c] == 0) v = 1.0;
                else if (ck[c] == 1) v = ((double)num[i * ldn + cs[c]] - sh[c]) / sc[c];
                else v = (codes[i * ldc + cs[c]] == cl[c]) ? 1.0 : 0.0;
                any[c] |= (v != 0.0);
                X[i * ldx + cbase + c * cstride] = (T)v;
            }
        }
    }
    if (seen) {
#pragma unroll
        for (int c = 0; c < DMAXC; ++c)
            if (ck[c] >= 0 && any[c]) seen[cbase + c * cstride] = 1;   // benign race: every writer stores 1
    }
}

template <typename T>
static int design_impl(const T* num, int64_t ldn, int q, const int32_t* codes, int64_t ldc, int f, int64_t n,
                       const int32_t* kind, const int32_t* src, const int32_t* level, const double* shift,
                       const double* scale, int p, T* X, int64_t ldx, int32_t* seen, hipStream_t stream) {
    DLSA_REQUIRE(X && kind && src && level && shift && scale, "design: null output or descriptor");
    DLSA_REQUIRE(p > 0 && p <= DTHREADS * DMAXC && n >= 0 && ldx >= p, "design: bad shape n=%lld p=%d ldx=%lld (p <= %d)",
                 (long long)n, p, (long long)ldx, DTHREADS * DMAXC);
    DLSA_REQUIRE(q >= 0 && f >= 0 && (q == 0 || (num && ldn >= q)) && (f == 0 || (codes && ldc >= f)),
                 "design: bad inputs q=%d f=%d ldn=%lld ldc=%lld", q, f, (long long)ldn, (long long)ldc);
    if (seen) DLSA_HIP_CHECK(hipMemsetAsync(seen, 0, (size_t)p * sizeof(int32_t), stream));
    if (n == 0) return DLSA_OK;
    const int64_t want = (n + DROWS - 1) / DROWS;
    const int blocks = (int)std::min<int64_t>(want, (int64_t)kNumCU * 16);
    hipLaunchKernelGGL((design_kernel<T>), dim3(blocks), dim3(DTHREADS), 0, stream, num, ldn, codes, ldc, n,
                       kind, src, level, shift, scale, p, X, ldx, seen);
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
