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

// factor the nb x nb diagonal block at (j0, j0) in place; one wave, lane = row, the row lives in REGISTERS.
// Per column j: the pivot is broadcast with a lane read, lane i > j forms l_ij, column j goes through a 32-entry
// LDS buffer once (broadcast reads, no dependent read-modify-write chain) and every lane updates the rest of its
// row: ~32 x (one divide + 31 FMAs).  (The first version kept the block in LDS and walked it with three barriers
// per column: 36 us per block, more than half of a p = 500 factorisation and most of a p = 50 one.)
__global__ __launch_bounds__(64) void chol_diag_kernel(double* __restrict__ L, int p, int j0, int nb,
                                                       double* __restrict__ stats) {
    __shared__ double col[2][NB];
    const int lane = threadIdx.x;
    double r[NB];
#pragma unroll
    for (int k = 0; k < NB; ++k)
        r[k] = (lane < nb && k <= lane) ? L[(int64_t)(j0 + lane) * p + j0 + k] : 0.0;
    int bad = 0;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        if (j < nb) {                                        // wave-uniform
            const double d = __shfl(r[j], j, 64);            // the pivot sits in lane j
            double sq;
            if (!(d > 0.0) || !isfinite(d)) { bad = isfinite(d) ? 1 : 2; sq = 1.0; }
            else sq = sqrt(d);
            const double lj = (lane == j) ? sq : r[j] / sq;  // l_ij for lane i >= j
            if (lane >= j) r[j] = lj;
            if (lane < NB) col[j & 1][lane] = (lane >= j && lane < nb) ? lj : 0.0;
            __syncthreads();
#pragma unroll
            for (int k = j + 1; k < NB; ++k)
                if (k <= lane) r[k] = fma(-lj, col[j & 1][k], r[k]);
        }
    }
    if (lane < nb) {
#pragma unroll
        for (int k = 0; k < NB; ++k)
            if (k <= lane) L[(int64_t)(j0 + lane) * p + j0 + k] = r[k];
    }
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

// ---------------------------------------------------------------------------------------------------------------
// Explicit inverse of the factor, for factors that are REUSED (frozen / inherited Hessians of the IRLS driver): the
// triangular solves above are a chain of ~2p/32 dependent block steps in one workgroup (0.33 ms at p = 500, more than
// a logit pass over 1e6 rows), whereas x = Linv' (Linv g) is two mat-vecs (~20 us).  Linv is built block row by
// block row: Linv[i][i] = L_ii^-1,  Linv[i][j] = -L_ii^-1 sum_{k=j}^{i-1} L[i][k] Linv[k][j]  (one launch per block
// row, one workgroup per block column).
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void tri_inverse_row_kernel(const double* __restrict__ L, int p, int bi,
                                                              double* __restrict__ Linv) {
    __shared__ double Dinv[NB][NB + 1];     // L_ii^-1
    __shared__ double A[NB][NB + 1];        // L[i][k] block
    __shared__ double B[NB][NB + 1];        // Linv[k][j] block
    __shared__ double T[NB][NB + 1];        // accumulated sum
    const int tid = threadIdx.x;
    const int i0 = bi * NB, nbi = min(NB, p - i0);
    const int bj = blockIdx.x;               // block column 0..bi
    const int j0 = bj * NB;                  // (always a full block: j < i)
    // inverse of the diagonal block: thread c < nbi builds column c by forward substitution (L_ii in A)
    for (int e = tid; e < NB * NB; e += 256) {
        const int r = e / NB, c = e % NB;
        A[r][c] = (r < nbi && c <= r) ? L[(int64_t)(i0 + r) * p + i0 + c] : 0.0;
        Dinv[r][c] = 0.0;
    }
    __syncthreads();
    if (tid < nbi) {
        const int c = tid;
        for (int r = c; r < nbi; ++r) {
            double sacc = (r == c) ? 1.0 : 0.0;
            for (int k = c; k < r; ++k) sacc -= A[r][k] * Dinv[k][c];
            Dinv[r][c] = sacc / A[r][r];
        }
    }
    __syncthreads();
    if (bj == bi) {                          // the diagonal block of Linv (and zeros above it)
        for (int e = tid; e < nbi * nbi; e += 256) Linv[(int64_t)(i0 + e / nbi) * p + i0 + e % nbi] = Dinv[e / nbi][e % nbi];
        return;
    }
    const int r = tid / 8, cq = (tid % 8) * 4;            // thread: row r, columns cq..cq+3 of the 32 x 32 result
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    for (int bk = bj; bk < bi; ++bk) {
        const int k0 = bk * NB;
        __syncthreads();
        for (int e = tid; e < NB * NB; e += 256) {
            const int rr = e / NB, cc = e % NB;
            A[rr][cc] = rr < nbi ? L[(int64_t)(i0 + rr) * p + k0 + cc] : 0.0;
            B[rr][cc] = Linv[(int64_t)(k0 + rr) * p + j0 + cc];
        }
        __syncthreads();
#pragma unroll 8
        for (int k = 0; k < NB; ++k) {
            const double a = A[r][k];
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[c] = fma(a, B[k][cq + c], acc[c]);
        }
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 4; ++c) T[r][cq + c] = acc[c];
    __syncthreads();
    if (r < nbi) {
        double out[4] = {0.0, 0.0, 0.0, 0.0};
        for (int k = 0; k <= r; ++k) {
            const double dv = Dinv[r][k];
#pragma unroll
            for (int c = 0; c < 4; ++c) out[c] = fma(dv, T[k][cq + c], out[c]);
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) Linv[(int64_t)(i0 + r) * p + j0 + cq + c] = -out[c];
    }
}

// x = Linv' (Linv rhs) in one workgroup; stats as chol_tri_solve_kernel
__global__ __launch_bounds__(1024) void inv_apply_kernel(const double* __restrict__ Linv, int p,
                                                         const double* __restrict__ rhs, const double* __restrict__ ref,
                                                         double* __restrict__ xout, double* __restrict__ stats) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    double* g = sm;              // p
    double* y = sm + p;          // p
    double* part = y + p;        // 2 * p (two row groups of phase 2)
    double* red = part + 2 * p;  // 48
    const int tid = threadIdx.x, nth = blockDim.x, lane = tid & 63, wave = tid >> 6, nw = nth >> 6;
    for (int i = tid; i < p; i += nth) g[i] = rhs[i];
    __syncthreads();
    for (int i = wave; i < p; i += nw) {                   // y_i = sum_{k<=i} Linv[i][k] g[k], a wave per row
        const double* row = Linv + (int64_t)i * p;
        double sacc = 0.0;
        for (int k = lane; k <= i; k += 64) sacc = fma(row[k], g[k], sacc);
        sacc = wave_allreduce_sum(sacc);
        if (lane == 0) y[i] = sacc;
    }
    __syncthreads();
    {                                                      // x_k = sum_{i>=k} Linv[i][k] y[i]: two row groups per column
        const int half = nth / 2, grp = tid / half, t = tid % half;
        for (int k = t; k < p; k += half) {
            double sacc = 0.0;
            for (int i = k + grp; i < p; i += 2) sacc = fma(Linv[(int64_t)i * p + k], y[i], sacc);
            part[grp * p + k] = sacc;
        }
    }
    __syncthreads();
    double mx = 0.0, mr = 0.0;
    int bad = 0;
    for (int i = tid; i < p; i += nth) {
        const double v = part[i] + part[p + i];
        xout[i] = v;
        mx = fmax(mx, fabs(v));
        if (!isfinite(v)) bad = 1;
        if (ref) mr = fmax(mr, fabs(ref[i]));
    }
    for (int m = 32; m >= 1; m >>= 1) {
        mx = fmax(mx, __shfl_xor(mx, m, 64));
        mr = fmax(mr, __shfl_xor(mr, m, 64));
        bad |= __shfl_xor(bad, m, 64);
    }
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

// ---------------------------------------------------------------------------------------------------------------
// Small systems (p <= 112: config 2's p = 100, config 1's p = 50): H^-1 and the Newton step in ONE launch.
// The blocked path above is 13 dependent launches for p = 100 + 5 for the factor's inverse + the p-row Gram that forms
// H^-1 for the quasi-Newton kernels: ~0.2 ms per fresh Hessian, four to nine times per config-2 fit.  A first one-launch
// version (left-looking Cholesky + forward-substituted inverse + Linv' Linv in LDS) was no faster: 220 us, all of it
// dependent LDS round trips (factor loop 88 us, inverse 59, H^-1 46: timed by leaving the phases out).  This one never
// factors: the SWEEP operator applied to every pivot in turn turns an SPD matrix A into -A^-1 in place,
//     sweep k:  A[i][j] -= A[i][k] A[k][j] / d  (i, j != k),  A[i][k] /= d,  A[k][j] /= d,  A[k][k] = -1 / d,   d = A[k][k],
// and the pivots d are the squares of the Cholesky pivots (d <= 0: not SPD).  The matrix lives in REGISTERS: thread (j, h) holds
// A[i][j] for the rows i = 2 s + h (56 doubles, static indices), so a sweep is 56 FMAs per thread against a SNAPSHOT of row k
// in LDS (every thread publishes its element of row k + 1 while it finishes sweep k) -- one barrier per sweep, no
// read-modify-write of LDS.  With the snapshot's pivot entry stored as d - 1 the one formula
//     A[i][j] -= s[i] s[j] / d
// also produces row k and column k (check: s[k] = d - 1 gives A[k][j] - (d - 1) A[k][j] / d = A[k][j] / d); only the pivot
// itself, kept in a register of its own, is set to -1 / d.  x = H^-1 rhs comes from the registers as well.
// ---------------------------------------------------------------------------------------------------------------
#ifndef CS_SKIP
#define CS_SKIP 0                        // timing experiments only (wrong results): 1 no sweeps
#endif
constexpr int CS_MAXP = 112;
constexpr int CS_NS = CS_MAXP / 2;       // rows per thread

// sweep K and, recursively, K + 1 ..: the recursion is what makes every register index static (a `#pragma unroll` loop with
// the early exit for k >= p is not unrolled by hipcc, and a[] then lives in scratch)
template <int K>
__device__ __forceinline__ void cs_sweeps(double (&a)[CS_NS], double& diag, double (&rk)[2][128], double (&dval)[2], double* red,
                                          int p, int tid, int j, int h) {
    if constexpr (K < CS_MAXP) {
        if (K >= p) return;
        constexpr int cur = K & 1;
        __syncthreads();
        double d = dval[cur];
        if (!(d > 0.0) || !isfinite(d)) { if (tid == 0) red[48] = fmax(red[48], isfinite(d) ? 1.0 : 2.0); d = 1.0; }
        const double dinv = 1.0 / d;
        const double sj = rk[cur][j];
        const double tj = sj * dinv;
#pragma unroll
        for (int sl = 0; sl < CS_NS; ++sl) a[sl] = fma(-rk[cur][2 * sl + h], tj, a[sl]);
        diag = (j == K) ? -dinv : fma(-sj, tj, diag);
        if constexpr (K + 1 < CS_MAXP) {         // publish this thread's element of row K + 1: slot (K + 1) >> 1 of the h == (K + 1) & 1 threads
            if (h == ((K + 1) & 1)) {
                rk[cur ^ 1][j] = (j == K + 1) ? diag - 1.0 : a[(K + 1) >> 1];
                if (j == K + 1) dval[cur ^ 1] = diag;
            }
        }
        cs_sweeps<K + 1>(a, diag, rk, dval, red, p, tid, j, h);
    }
}

// Batched form (irls_batch.hip): workgroup b works on matrix b -- A + b sA, vectors + b sV, Hinv + b sH, stats + b sS -- and leaves at
// once when active[b] == 0.  The single-matrix launch passes zero strides and no mask.
struct CsBatch { int64_t sA, sV, sH, sS; const int* active; };
__global__ __launch_bounds__(256, 2) void spd_inverse_small_kernel(const double* __restrict__ A_, int64_t lda, int p,
                                                                const double* __restrict__ rhs_, const double* __restrict__ ref_,
                                                                double* __restrict__ Hinv_, double* __restrict__ xout_,
                                                                double* __restrict__ stats_, CsBatch bt) {
    const int bi = blockIdx.x;
    if (bt.active && !bt.active[bi]) return;
    const double* __restrict__ A = A_ + bi * bt.sA;
    const double* __restrict__ rhs = rhs_ + bi * bt.sV;
    const double* __restrict__ ref = ref_ ? ref_ + bi * bt.sV : nullptr;
    double* __restrict__ Hinv = Hinv_ + bi * bt.sH;
    double* __restrict__ xout = xout_ + bi * bt.sV;
    double* __restrict__ stats = stats_ + bi * bt.sS;
    __shared__ __attribute__((aligned(16))) double rk[2][128];      // snapshot of the pivot row (= column, by symmetry), indexed by column
    __shared__ double dval[2];                                      // the pivot itself
    __shared__ double gv[128], xpart[2][128], red[52];
    const int tid = threadIdx.x, j = tid & 127, h = tid >> 7;       // h is wave-uniform (waves 0, 1: even rows; 2, 3: odd rows)
    const bool col = j < p;
    double a[CS_NS];
#pragma unroll
    for (int sl = 0; sl < CS_NS; ++sl) {
        const int i = 2 * sl + h;
        a[sl] = (col && i < p) ? A[(int64_t)i * lda + j] : 0.0;
    }
    double diag = col ? A[(int64_t)j * lda + j] : 1.0;
    if (tid < 128) { gv[tid] = tid < p ? rhs[tid] : 0.0; rk[0][tid] = 0.0; rk[1][tid] = 0.0; }
    if (tid == 0) red[48] = 0.0;
    __syncthreads();
    // Row k + 1 (the next pivot row) is published element by element: A[k + 1][j] sits in slot (k + 1) >> 1 of the threads with
    // h == (k + 1) & 1 -- a STATIC register index because the sweeps are unrolled by template recursion (112 sweeps x 56 FMAs:
    // ~80 KB of code, streamed once).  One 8-byte LDS store per thread and sweep; the owner of the pivot stores d - 1 in its
    // place and d beside it.  The readers fetch rk[2 sl + h]: the same address in every lane, a broadcast.
    if (h == 0) {
        rk[0][j] = (j == 0) ? diag - 1.0 : a[0];
        if (j == 0) dval[0] = diag;
    }
    if (!(CS_SKIP & 1)) cs_sweeps<0>(a, diag, rk, dval, red, p, tid, j, h);
    // ---- H^-1 = -A (the slot of the diagonal element is stale: the pivot register holds it)
    double xs = 0.0;
#pragma unroll
    for (int sl = 0; sl < CS_NS; ++sl) {
        const int i = 2 * sl + h;
        if (col && i < p) {
            const double v = (i == j) ? -diag : -a[sl];
            Hinv[(int64_t)i * p + j] = v;
            xs = fma(v, gv[i], xs);              // x_j = sum_i H^-1[i][j] rhs[i]  (column j = row j)
        }
    }
    if (tid < 256) xpart[h][j] = xs;
    __syncthreads();
    double mx = 0.0, mr = 0.0;
    int bad = 0;
    if (tid < p) {
        const double v = xpart[0][tid] + xpart[1][tid];
        xout[tid] = v;
        mx = fabs(v);
        if (!isfinite(v)) bad = 1;
        if (ref) mr = fabs(ref[tid]);
    }
    mx = wave_allreduce_max(mx);
    mr = wave_allreduce_max(mr);
    for (int m2 = 32; m2 >= 1; m2 >>= 1) bad |= __shfl_xor(bad, m2, 64);
    if ((tid & 63) == 0) { red[tid >> 6] = mx; red[16 + (tid >> 6)] = mr; red[32 + (tid >> 6)] = (double)bad; }
    __syncthreads();
    if (tid == 0) {
        double m0 = 0.0, m1 = 0.0, c = 0.0;
        for (int q = 0; q < 4; ++q) { m0 = fmax(m0, red[q]); m1 = fmax(m1, red[16 + q]); c = fmax(c, red[32 + q]); }
        stats[0] = m0;
        stats[1] = m1;
        stats[2] = red[48] != 0.0 ? red[48] : (c != 0.0 ? 2.0 : 0.0);
    }
}

bool chol_small_ok(int p) {
    const char* e = kernel_knob("DLSA_CHOL_SMALL");           // 0: always the blocked path (A/B runs)
    return p <= CS_MAXP && (!e || atoi(e) != 0);
}

// H^-1 (p <= 112) into Hinv and x = H^-1 rhs in one launch; stats as launch_chol_solve.  No factor is produced: the callers
// (irls.hip) only take this path when their later steps use H^-1 (the fused quasi-Newton kernel).
int launch_chol_small(const double* A, int64_t lda, int p, const double* rhs, const double* ref, double* Hinv, double* xout,
                      double* stats, hipStream_t s) {
    DLSA_REQUIRE(p > 0 && p <= CS_MAXP, "spd_inverse_small: p=%d", p);
    hipLaunchKernelGGL(spd_inverse_small_kernel, dim3(1), dim3(256), 0, s, A, lda, p, rhs, ref, Hinv, xout, stats, CsBatch{0, 0, 0, 0, nullptr});
    DLSA_HIP_CHECK(hipGetLastError());
    return DLSA_OK;
}

// the same for `count` matrices in one launch: matrix b = A + b sA (row pitch lda), rhs / ref / xout + b sV, Hinv + b sH (p x p), stats + b sS
int launch_chol_small_batched(int count, const double* A, int64_t lda, int64_t sA, int p, const double* rhs, const double* ref, int64_t sV,
                              double* Hinv, int64_t sH, double* xout, double* stats, int64_t sS, const int* active, hipStream_t s) {
    DLSA_REQUIRE(p > 0 && p <= CS_MAXP && count > 0, "spd_inverse_small (batched): p=%d count=%d", p, count);
    hipLaunchKernelGGL(spd_inverse_small_kernel, dim3(count), dim3(256), 0, s, A, lda, p, rhs, ref, Hinv, xout, stats, CsBatch{sA, sV, sH, sS, active});
    DLSA_HIP_CHECK(hipGetLastError());
    return DLSA_OK;
}

int launch_tri_inverse(const double* L, int p, double* Linv, hipStream_t s) {
    DLSA_HIP_CHECK(hipMemsetAsync(Linv, 0, (size_t)p * p * sizeof(double), s));
    const int nblk = (p + NB - 1) / NB;
    for (int bi = 0; bi < nblk; ++bi)
        hipLaunchKernelGGL(tri_inverse_row_kernel, dim3(bi + 1), dim3(256), 0, s, L, p, bi, Linv);
    DLSA_HIP_CHECK(hipGetLastError());
    return DLSA_OK;
}

int launch_inv_apply(const double* Linv, int p, const double* rhs, const double* ref, double* xout, double* stats,
                     hipStream_t s) {
    const size_t shm = ((size_t)4 * p + 48) * sizeof(double);
    DLSA_REQUIRE(shm <= 64 * 1024, "inverse apply: p=%d too large", p);
    hipLaunchKernelGGL(inv_apply_kernel, dim3(1), dim3(p <= 64 ? 256 : 1024), shm, s, Linv, p, rhs, ref, xout, stats);
    DLSA_HIP_CHECK(hipGetLastError());
    return DLSA_OK;
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
