// Blocked right-looking Cholesky + triangular solves for one p x p SPD system on the device:
// the inner solver of every Newton step (replaces the linear solves inside sklearn's newton-cg,
// dlsa/models.py:113) and the WLS combine  theta = Sig_inv^{-1} v  (dlsa/dlsa.py:48-49).
//
// The matrix is tiny next to the data passes (p <= 2048), so the design goal is latency: the
// factorisation is a short chain of small launches per 32-column block --
//   diag  (one wave factors the 32x32 diagonal block in LDS)
//   panel (thread per row: forward substitution against the diagonal block held in LDS)
//   trail (SYRK update of the trailing lower triangle, 64x64 tiles over many workgroups)
// -- instead of one workgroup grinding through p dependent column steps (6.5 ms at p=500 before;
// see profiles/).  The two triangular solves run in ONE workgroup with one barrier pair per block.
// stats: [0] max|x|, [1] max|ref| (ref nullable), [2] info (0 ok, 1 not SPD, 2 NaN/Inf).
#include "common.h"
#include <algorithm>

namespace dlsa {

constexpr int NB = 32;

__global__ void chol_copy_lower_kernel(const double* __restrict__ A, int64_t lda, int p, double* __restrict__ L,
                                       double* __restrict__ stats) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e == 0) stats[2] = 0.0;
    if (e >= (int64_t)p * p) return;
    const int i = (int)(e / p), k = (int)(e % p);
    L[e] = (k <= i) ? A[(int64_t)i * lda + k] : 0.0;
}

// factor the nb x nb diagonal block at (j0, j0) in place; one wave, lane = row
__global__ __launch_bounds__(64) void chol_diag_kernel(double* __restrict__ L, int p, int j0, int nb,
                                                       double* __restrict__ stats) {
    __shared__ double S[NB][NB + 1];
    const int lane = threadIdx.x;
    for (int e = lane; e < nb * nb; e += 64) S[e / nb][e % nb] = L[(int64_t)(j0 + e / nb) * p + j0 + e % nb];
    __syncthreads();
    int bad = 0;
    for (int j = 0; j < nb; ++j) {
        const double d = S[j][j];
        double s;
        if (!(d > 0.0) || !isfinite(d)) { bad = isfinite(d) ? 1 : 2; s = 1.0; }
        else s = sqrt(d);
        __syncthreads();
        if (lane == j) S[j][j] = s;
        if (lane > j && lane < nb) S[lane][j] /= s;
        __syncthreads();
        if (lane > j && lane < nb) {
            const double lij = S[lane][j];
            for (int k = j + 1; k <= lane; ++k) S[lane][k] -= lij * S[k][j];
        }
        __syncthreads();
    }
    for (int e = lane; e < nb * nb; e += 64)
        if (e % nb <= e / nb) L[(int64_t)(j0 + e / nb) * p + j0 + e % nb] = S[e / nb][e % nb];
    if (lane == 0 && bad) stats[2] = fmax(stats[2], (double)bad);
}

// rows below the diagonal block: L[i, J] <- A[i, J] * L_JJ^{-T}   (thread per row)
__global__ __launch_bounds__(256) void chol_panel_kernel(double* __restrict__ L, int p, int j0, int nb) {
    __shared__ double D[NB][NB + 1];
    for (int e = threadIdx.x; e < nb * nb; e += 256) D[e / nb][e % nb] = L[(int64_t)(j0 + e / nb) * p + j0 + e % nb];
    __syncthreads();
    const int i = j0 + nb + blockIdx.x * 256 + threadIdx.x;
    if (i >= p) return;
    double* row = L + (int64_t)i * p + j0;
    double x[NB];
#pragma unroll
    for (int c = 0; c < NB; ++c) x[c] = (c < nb) ? row[c] : 0.0;
#pragma unroll
    for (int c = 0; c < NB; ++c) {
        if (c < nb) {
            double s = x[c];
#pragma unroll
            for (int t = 0; t < c; ++t) s -= x[t] * D[c][t];
            x[c] = s / D[c][c];
        }
    }
#pragma unroll
    for (int c = 0; c < NB; ++c) if (c < nb) row[c] = x[c];
}

// trailing update: for i >= k >= j1:  L[i][k] -= sum_t L[i][j0+t] * L[k][j0+t]
__global__ __launch_bounds__(256) void chol_trail_kernel(double* __restrict__ L, int p, int j0, int nb, int j1) {
    // tile (ti, tk) with ti >= tk, enumerated along blockIdx.x
    int t = blockIdx.x, ti = 0;
    while (t > ti) { t -= ti + 1; ++ti; }
    const int tk = t;
    __shared__ double Pi[64][NB + 1], Pk[64][NB + 1];
    const int i0 = j1 + ti * 64, k0 = j1 + tk * 64;
    for (int e = threadIdx.x; e < 64 * nb; e += 256) {
        const int r = e / nb, c = e % nb;
        Pi[r][c] = (i0 + r < p) ? L[(int64_t)(i0 + r) * p + j0 + c] : 0.0;
        Pk[r][c] = (k0 + r < p) ? L[(int64_t)(k0 + r) * p + j0 + c] : 0.0;
    }
    __syncthreads();
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;     // thread -> 4x4 outputs: rows ty*4.., cols tx*4..
    double acc[4][4] = {};
    for (int c = 0; c < nb; ++c) {
        double a[4], b[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { a[u] = Pi[ty * 4 + u][c]; b[u] = Pk[tx * 4 + u][c]; }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int v = 0; v < 4; ++v) acc[u][v] = fma(a[u], b[v], acc[u][v]);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int i = i0 + ty * 4 + u;
        if (i >= p) continue;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int k = k0 + tx * 4 + v;
            if (k <= i && k < p) L[(int64_t)i * p + k] -= acc[u][v];
        }
    }
}

// forward (L z = rhs) and backward (L' x = z) solves, blocked by NB, one workgroup
__global__ __launch_bounds__(1024) void chol_tri_solve_kernel(const double* __restrict__ L, int p,
                                                              const double* __restrict__ rhs,
                                                              const double* __restrict__ ref,
                                                              double* __restrict__ xout, double* __restrict__ stats) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    double* vec = sm;                       // p
    double* D = sm + p;                     // NB x (NB+1) diagonal block
    double* red = D + NB * (NB + 1);        // 2 * 16 + 2
    const int tid = threadIdx.x, nth = blockDim.x;
    for (int i = tid; i < p; i += nth) vec[i] = rhs[i];
    __syncthreads();
    const int nblk = (p + NB - 1) / NB;
    for (int bj = 0; bj < nblk; ++bj) {
        const int j0 = bj * NB, nb = min(NB, p - j0);
        for (int e = tid; e < nb * nb; e += nth) D[(e / nb) * (NB + 1) + e % nb] = L[(int64_t)(j0 + e / nb) * p + j0 + e % nb];
        __syncthreads();
        if (tid < 64) {                     // wave 0: sequential substitution inside the block
            for (int c = 0; c < nb; ++c) {
                const double xc = vec[j0 + c] / D[c * (NB + 1) + c];
                __builtin_amdgcn_wave_barrier();
                if (tid == c) vec[j0 + c] = xc;
                if (tid > c && tid < nb) vec[j0 + tid] -= D[tid * (NB + 1) + c] * xc;
                __builtin_amdgcn_wave_barrier();
            }
        }
        __syncthreads();
        for (int i = j0 + nb + tid; i < p; i += nth) {
            const double* row = L + (int64_t)i * p + j0;
            double s = vec[i];
            for (int c = 0; c < nb; ++c) s -= row[c] * vec[j0 + c];
            vec[i] = s;
        }
        __syncthreads();
    }
    for (int bj = nblk - 1; bj >= 0; --bj) {
        const int j0 = bj * NB, nb = min(NB, p - j0);
        for (int e = tid; e < nb * nb; e += nth) D[(e / nb) * (NB + 1) + e % nb] = L[(int64_t)(j0 + e / nb) * p + j0 + e % nb];
        __syncthreads();
        if (tid < 64) {                     // L' x = z inside the block: column c of L' is row c of L
            for (int c = nb - 1; c >= 0; --c) {
                const double xc = vec[j0 + c] / D[c * (NB + 1) + c];
                __builtin_amdgcn_wave_barrier();
                if (tid == c) vec[j0 + c] = xc;
                if (tid < c) vec[j0 + tid] -= D[c * (NB + 1) + tid] * xc;
                __builtin_amdgcn_wave_barrier();
            }
        }
        __syncthreads();
        for (int i = tid; i < j0; i += nth) {          // rows above: vec[i] -= sum_c L[j0+c][i] x_c  (coalesced in i)
            double s = vec[i];
            for (int c = 0; c < nb; ++c) s -= L[(int64_t)(j0 + c) * p + i] * vec[j0 + c];
            vec[i] = s;
        }
        __syncthreads();
    }
    double mx = 0.0, mr = 0.0;
    int bad = 0;
    for (int i = tid; i < p; i += nth) {
        const double v = vec[i];
        xout[i] = v;
        mx = fmax(mx, fabs(v));
        if (!isfinite(v)) bad = 1;
        if (ref) mr = fmax(mr, fabs(ref[i]));
    }
    for (int m2 = 32; m2 >= 1; m2 >>= 1) {
        mx = fmax(mx, __shfl_xor(mx, m2, 64));
        mr = fmax(mr, __shfl_xor(mr, m2, 64));
        bad |= __shfl_xor(bad, m2, 64);
    }
    if ((tid & 63) == 0) { red[tid >> 6] = mx; red[16 + (tid >> 6)] = mr; red[32 + (tid >> 6)] = (double)bad; }
    __syncthreads();
    if (tid == 0) {
        double a = 0.0, b = 0.0, c = 0.0;
        for (int k = 0; k < nth / 64; ++k) { a = fmax(a, red[k]); b = fmax(b, red[16 + k]); c = fmax(c, red[32 + k]); }
        stats[0] = a;
        stats[1] = b;
        if (c != 0.0 && stats[2] == 0.0) stats[2] = 2.0;
    }
}

int launch_chol_solve(const double* A, int64_t lda, int64_t strideA, const double* rhs, int64_t stride_rhs,
                      const double* ref, int64_t stride_ref, int p, int nsys, double* Lws, double* xout,
                      int64_t stride_x, double* stats, int64_t stride_stats, hipStream_t s, int reuse_factor) {
    const size_t shm = ((size_t)p + NB * (NB + 1) + 48 + 4) * sizeof(double);
    DLSA_REQUIRE(shm <= 64 * 1024, "spd solve: p=%d too large", p);
    for (int sys = 0; sys < nsys; ++sys) {
        const double* As = A + sys * strideA;
        double* L = Lws + (int64_t)sys * p * p;
        double* st = stats + sys * stride_stats;
        if (!reuse_factor) {
            const int64_t tot = (int64_t)p * p;
            hipLaunchKernelGGL(chol_copy_lower_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, As, lda, p, L, st);
            for (int j0 = 0; j0 < p; j0 += NB) {
                const int nb = std::min(NB, p - j0);
                hipLaunchKernelGGL(chol_diag_kernel, dim3(1), dim3(64), 0, s, L, p, j0, nb, st);
                const int j1 = j0 + nb;
                const int m = p - j1;
                if (m > 0) {
                    hipLaunchKernelGGL(chol_panel_kernel, dim3((m + 255) / 256), dim3(256), 0, s, L, p, j0, nb);
                    const int nt = (m + 63) / 64;
                    hipLaunchKernelGGL(chol_trail_kernel, dim3(nt * (nt + 1) / 2), dim3(256), 0, s, L, p, j0, nb, j1);
                }
            }
        }
        hipLaunchKernelGGL(chol_tri_solve_kernel, dim3(1), dim3(p <= 64 ? 256 : 1024), shm, s, (const double*)L, p,
                           rhs + sys * stride_rhs, ref ? ref + sys * stride_ref : nullptr, xout + sys * stride_x, st);
    }
    DLSA_HIP_CHECK(hipGetLastError());
    return DLSA_OK;
}

}  // namespace dlsa
