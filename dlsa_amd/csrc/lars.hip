// LARS / lasso path for the least-squares approximation on the device
// (reference: lars_lsa and its helpers, dlsa/lsa.py:8-212; selection by AIC/BIC, dlsa/dlsa.py:87-105).
//
// The path is strictly sequential over steps, so it runs as ONE persistent workgroup
// (1024 threads = 16 waves on one CU) with the p x p matrices in HBM/L2, the short vectors of the inner
// loops in LDS and no host round trips inside the path.  A step is a chain of dependent phases, each moving
// only a few hundred KB, so what matters is the number of loads in flight per phase and the number of
// phases, not bytes.  Differences in *method* (not in result) from the reference:
//   * the reference keeps the Cholesky factor R of Sigma[active,active] and does two triangular solves per
//     step (lsa.py:151); triangular solves are k dependent steps, so the kernel keeps R^{-1} (row-major, and
//     its transpose, so both R^{-1}v and R^{-T}v read contiguous rows) and appends a column with two
//     mat-vecs (lsa.py:12-32 updateR);
//   * Gi1 = R^{-1} R^{-T} s (lsa.py:151-153) is carried along: appending column c = R^{-1}[:, k] with sign s_k
//     adds one entry t_k = (s_k - r.t)/r_kk to t = R^{-T} s and the rank-one term c t_k to Gi1 -- O(k)
//     per step instead of two O(k^2) mat-vecs;
//   * a = w Sigma[active, inactive] (lsa.py:157-160) and Sigma[:,active] w (lsa.py:177) are the
//     same vector u by symmetry and are computed once;
//   * RSS_k = (b-beta_k)' Sigma (b-beta_k) (lsa.py:190-192) equals (b-beta_k).Cvec_k because
//     Cvec_k = Sigma (b - beta_k) is carried along the path -- O(p) per step instead of O(p^2);
//   * a lasso drop (lsa.py:179-186) re-appends the positions from the first dropped one on (the factor of the ordered
//     set before it is unchanged) instead of Givens-downdating R (lsa.py:35-80); the factor of the remaining ordered
//     set is unique.
#include "common.h"
#include "lars.h"
#include "options.h"
#include <math.h>
#include <stdlib.h>
#include <algorithm>
#include <atomic>
#include <mutex>

namespace dlsa {

[[maybe_unused]] constexpr int LARS_SLACK = 256;                // zeroed elements after each factor matrix: the last rows' trips run into them
constexpr int LARS_MAX_WGS = 32;               // workgroups of the grid kernel (the barrier costs ~35 ns per workgroup beyond 16)
#ifndef DLSA_LARS_SECONDARY
std::mutex g_lars_grid_mu;                                     // serialises this process's grid-kernel launches (dlsa_lars_lsa_f64)
std::atomic<long long> g_lars_barrier_timeout_ticks{25000000ll};    // 0.25 s of the 100 MHz wall clock (a barrier wait is microseconds)
std::atomic<int> g_lars_grid_aborts{0};
#endif


// The kernels are compiled TWICE (Makefile: lars.hip and lars_t512 from the same source): workgroups of 1024 threads for wide
// paths, of 512 for p <= 768 -- every phase of a step ends in a workgroup barrier or a block reduction, and those cost by the
// number of waves (same box, 1024 -> 512 threads: p = 50 0.58 -> 0.44 ms, p = 100 1.10 -> 0.89, p = 260 3.91 -> 3.62, p = 500
// 8.40 -> 7.90; p = 1000 equal; p = 2000 73.7 -> 99.7: there the mat-vecs want the waves).
#ifndef DLSA_LARS_THREADS
#define DLSA_LARS_THREADS 1024
#endif
#if DLSA_LARS_THREADS == 512
#define LARS_NS lars_t512
#else
#define LARS_NS lars_t1024
#endif
namespace LARS_NS {
constexpr int LARS_THREADS = DLSA_LARS_THREADS;
constexpr int LARS_WAVES = LARS_THREADS / 64;
constexpr int LARS_ROWGROUPS = LARS_THREADS / 16;
#ifndef DLSA_LARS_TRIP
#define DLSA_LARS_TRIP 2
#endif
constexpr int LARS_TRIP = DLSA_LARS_TRIP;      // loads per row, lane and trip in the triangular mat-vecs (four rows at a time)


// Optional phase timer (-DDLSA_LARS_PROF): thread 0 accumulates wall-clock ticks (100 MHz) per phase and prints them.
#ifdef DLSA_LARS_PROF
__shared__ long long lars_prof_t[16];
__shared__ long long lars_prof_last;
#define LARS_PROF_DECL do { if (threadIdx.x == 0) { for (int q_ = 0; q_ < 16; ++q_) lars_prof_t[q_] = 0; lars_prof_last = wall_clock64(); } } while (0)
#define LARS_TICK(i) do { if (threadIdx.x == 0) { const long long now_ = wall_clock64(); lars_prof_t[i] += now_ - lars_prof_last; lars_prof_last = now_; } } while (0)
#else
#define LARS_PROF_DECL
#define LARS_TICK(i)
#endif

// LDS-free wave reductions (common.h): this kernel is one long chain of dependent reductions
__device__ __forceinline__ double wave_sum(double v) { return wave_allreduce_sum(v); }
__device__ __forceinline__ double wave_max(double v) { return wave_allreduce_max(v); }
__device__ __forceinline__ double wave_min(double v) { return wave_allreduce_min(v); }
__device__ __forceinline__ double row16_sum(double s) {      // over the 16 lanes of one DPP row, result in all of them
    s += dpp_xor_f64<8>(s);
    s += dpp_xor_f64<4>(s);
    s += dpp_xor_f64<2>(s);
    s += dpp_xor_f64<1>(s);
    return s;
}

// block-wide reductions of NV values at once; `red` is LDS scratch of 4*LARS_WAVES doubles.  Result to all threads.
template <int NV, typename Op>
__device__ __forceinline__ void block_reduce(double (&v)[NV], double* red, Op op) {
    static_assert(NV <= 3, "red holds three values per wave");
#pragma unroll
    for (int q = 0; q < NV; ++q) v[q] = wave_allreduce(v[q], op);
    __syncthreads();
    if ((threadIdx.x & 63) == 0)
#pragma unroll
        for (int q = 0; q < NV; ++q) red[q * LARS_WAVES + (threadIdx.x >> 6)] = v[q];
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NV; ++q) {
        double s = red[q * LARS_WAVES];
        for (int k = 1; k < LARS_WAVES; ++k) s = op(s, red[q * LARS_WAVES + k]);
        v[q] = s;
    }
}
// two groups of values with different operators in one pass (NA + NB <= 4)
template <int NA, typename OpA, int NB, typename OpB>
__device__ __forceinline__ void block_reduce2(double (&va)[NA], OpA opa, double (&vb)[NB], OpB opb, double* red) {
    static_assert(NA + NB <= 4, "red holds four values per wave");
#pragma unroll
    for (int q = 0; q < NA; ++q) va[q] = wave_allreduce(va[q], opa);
#pragma unroll
    for (int q = 0; q < NB; ++q) vb[q] = wave_allreduce(vb[q], opb);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int q = 0; q < NA; ++q) red[q * LARS_WAVES + (threadIdx.x >> 6)] = va[q];
#pragma unroll
        for (int q = 0; q < NB; ++q) red[(NA + q) * LARS_WAVES + (threadIdx.x >> 6)] = vb[q];
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NA; ++q) {
        double s = red[q * LARS_WAVES];
        for (int k = 1; k < LARS_WAVES; ++k) s = opa(s, red[q * LARS_WAVES + k]);
        va[q] = s;
    }
#pragma unroll
    for (int q = 0; q < NB; ++q) {
        double s = red[(NA + q) * LARS_WAVES];
        for (int k = 1; k < LARS_WAVES; ++k) s = opb(s, red[(NA + q) * LARS_WAVES + k]);
        vb[q] = s;
    }
}
__device__ double block_sum(double v, double* red) { double a[1] = {v}; block_reduce(a, red, WaveOpSum()); return a[0]; }

// y_i = sum_l M[i][l] x[l] over the stored part of row i of a triangular matrix: l in [0, i] (LOWER) or [i, n).
// The other triangle of M is zero (and so are LARS_SLACK elements after the matrix), so nothing but x is masked:
// a 16-lane group owns four consecutive rows at a time, forms four row pointers once and issues 4 x LARS_TRIP
// loads per lane and trip at constant offsets from them.  x is in LDS and shared by the four rows;
// emit(i, y_i) runs on one lane.
template <bool LOWER, typename Emit>
__device__ __forceinline__ void tri_matvec(const double* __restrict__ M, int ld, int n, const double* xs, Emit&& emit) {
    const int grp = threadIdx.x >> 4, l16 = threadIdx.x & 15;
    for (int base = 4 * grp; base < n; base += 4 * LARS_ROWGROUPS) {
        const int lo = (LOWER ? 0 : base) + l16, hi = LOWER ? min(base + 4, n) : n;
        const double* __restrict__ r0 = M + (int64_t)base * ld + lo;
        const double* __restrict__ r1 = M + (int64_t)min(base + 1, n - 1) * ld + lo;
        const double* __restrict__ r2 = M + (int64_t)min(base + 2, n - 1) * ld + lo;
        const double* __restrict__ r3 = M + (int64_t)min(base + 3, n - 1) * ld + lo;
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        for (int l = lo, o = 0; l < hi; l += 16 * LARS_TRIP, o += 16 * LARS_TRIP) {
            double m0[LARS_TRIP], m1[LARS_TRIP], m2[LARS_TRIP], m3[LARS_TRIP];
#pragma unroll
            for (int c = 0; c < LARS_TRIP; ++c) {
                m0[c] = r0[o + 16 * c]; m1[c] = r1[o + 16 * c]; m2[c] = r2[o + 16 * c]; m3[c] = r3[o + 16 * c];
            }
#pragma unroll
            for (int c = 0; c < LARS_TRIP; ++c) {
                const int lc = l + 16 * c;
                const double x = lc < hi ? xs[min(lc, hi - 1)] : 0.0;
                s0 = fma(m0[c], x, s0); s1 = fma(m1[c], x, s1); s2 = fma(m2[c], x, s2); s3 = fma(m3[c], x, s3);
            }
        }
        s0 = row16_sum(s0); s1 = row16_sum(s1); s2 = row16_sum(s2); s3 = row16_sum(s3);
        if (l16 == 0) {
            emit(base, s0);
            if (base + 1 < n) emit(base + 1, s1);
            if (base + 2 < n) emit(base + 2, s2);
            if (base + 3 < n) emit(base + 3, s3);
        }
    }
}

// out[j] = sum_{i<na} wv[i] S[act[i]][j] for all j < m (rows of the symmetric S, so this is S[:,act] wv).
// A thread owns two adjacent columns (16-byte loads); the i range is split over G groups of JT2 threads with
// eight independent loads in flight per thread; partial sums meet in LDS.
__device__ void sym_matvec(const double* __restrict__ S, int ld, int m, int na, const double* wv, const int* act,
                           double* __restrict__ out, double2* part, int JT2, int G) {
    const int jx = threadIdx.x % JT2, g = threadIdx.x / JT2;
    for (int j0 = 0; j0 < m; j0 += 2 * JT2) {
        const int j = j0 + 2 * jx;
        double2 a0 = {0.0, 0.0}, a1 = a0, a2 = a0, a3 = a0;
        if (j < m) {
            const double* __restrict__ col = S + j;
            int i = g;
            for (; i + 7 * G < na; i += 8 * G) {
                double2 v[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] = *reinterpret_cast<const double2*>(col + (int64_t)act[i + q * G] * ld);
#pragma unroll
                for (int q = 0; q < 8; q += 4) {
                    const double w0 = wv[i + q * G], w1 = wv[i + (q + 1) * G], w2 = wv[i + (q + 2) * G], w3 = wv[i + (q + 3) * G];
                    a0.x = fma(w0, v[q].x, a0.x); a0.y = fma(w0, v[q].y, a0.y);
                    a1.x = fma(w1, v[q + 1].x, a1.x); a1.y = fma(w1, v[q + 1].y, a1.y);
                    a2.x = fma(w2, v[q + 2].x, a2.x); a2.y = fma(w2, v[q + 2].y, a2.y);
                    a3.x = fma(w3, v[q + 3].x, a3.x); a3.y = fma(w3, v[q + 3].y, a3.y);
                }
            }
            for (; i < na; i += G) {
                const double2 v = *reinterpret_cast<const double2*>(col + (int64_t)act[i] * ld);
                const double w0 = wv[i];
                a0.x = fma(w0, v.x, a0.x); a0.y = fma(w0, v.y, a0.y);
            }
        }
        part[g * JT2 + jx] = double2{(a0.x + a1.x) + (a2.x + a3.x), (a0.y + a1.y) + (a2.y + a3.y)};
        __syncthreads();
        if (g == 0 && j < m) {
            double2 t = part[jx];
            for (int q = 1; q < G; ++q) { t.x += part[q * JT2 + jx].x; t.y += part[q * JT2 + jx].y; }
            out[j] = t.x;
            if (j + 1 < m) out[j + 1] = t.y;
        }
        __syncthreads();
    }
}

// State of the factor of Sigma[active,active] that the appends maintain.
struct LarsFactor {
    const double* __restrict__ S;
    double* __restrict__ Rinv;
    double* __restrict__ RinvT;
    int ld;
    int* active;  int* sh_act;         // active list (variable ids): global copy and LDS mirror
    double* sgn;                       // sign of the correlation at entry, by active position
    double* t;                         // R^{-T} sgn (LDS)
    double* gi1;                       // R^{-1} R^{-T} sgn (LDS)
    double* sh_x; double* sh_r;        // LDS scratch, m doubles each
    double* red;
};

// Append variable `inew` with sign `sg` at position na (lsa.py:12-32 updateR, on R^{-1}) and extend t and Gi1;
// tsq accumulates |t|^2 = sgn' Gi1 = 1/A^2.
// Returns (to all threads) 1 if the rank grew, 0 if the column is machine-singular (nothing is modified then).
__device__ int append_column(const LarsFactor& f, int na, int inew, double sg, double eps, double& tsq) {
    const int tid = threadIdx.x;
    const double* __restrict__ srow = f.S + (int64_t)inew * f.ld;
    LARS_TICK(2);
    for (int i = tid; i < na; i += LARS_THREADS) f.sh_x[i] = srow[f.sh_act[i]];
    __syncthreads();
    LARS_TICK(8);
    // r = R^{-T} x
    tri_matvec<true>(f.RinvT, f.ld, na, f.sh_x, [&](int i, double s) { f.sh_r[i] = s; });
    __syncthreads();
    LARS_TICK(9);
    double acc[2] = {0.0, 0.0};
    for (int i = tid; i < na; i += LARS_THREADS) {
        const double ri = f.sh_r[i];
        acc[0] = fma(ri, ri, acc[0]);
        acc[1] = fma(ri, f.t[i], acc[1]);
    }
    block_reduce(acc, f.red, WaveOpSum());
    LARS_TICK(10);
    double rpp = srow[inew] - acc[0];
    if (na > 0 && rpp <= eps) return 0;  // rank did not grow: caller records an "ignore"
    rpp = sqrt(rpp);
    const double tn = (sg - acc[1]) / rpp;      // new entry of R^{-T} sgn
    tsq = fma(tn, tn, tsq);
    // new column of R^{-1}: c = [-R^{-1} r / rpp ; 1/rpp];  Gi1 += c tn
    double* __restrict__ Rinv = f.Rinv;
    double* __restrict__ RinvT = f.RinvT;
    const int ld = f.ld;
    tri_matvec<false>(Rinv, ld, na, f.sh_r, [&](int i, double s) {
        const double c = -s / rpp;
        Rinv[(int64_t)i * ld + na] = c;
        RinvT[(int64_t)na * ld + i] = c;
        f.gi1[i] = fma(c, tn, f.gi1[i]);
    });
    LARS_TICK(11);
    if (tid == 0) {
        const double c = 1.0 / rpp;
        Rinv[(int64_t)na * ld + na] = c;
        RinvT[(int64_t)na * ld + na] = c;
        f.gi1[na] = c * tn;
        f.t[na] = tn;
        f.sgn[na] = sg;
        f.active[na] = inew;
        f.sh_act[na] = inew;
    }
    __syncthreads();
    LARS_TICK(12);
    return 1;
}

__global__ __launch_bounds__(LARS_THREADS) void lars_kernel(LarsArgs a) {
    __shared__ double red[4 * LARS_WAVES];
    __shared__ int sh_i[4];
    extern __shared__ __attribute__((aligned(16))) double dyn[];     // sh_part[2*LARS_THREADS] | sh_w, sh_x, sh_r, sh_t, sh_gi1 [m] | sh_act[m]
    const int tid = threadIdx.x;
    const int p = a.p;
    const int off = a.intercept ? 1 : 0;
    const int m = p - off;
    const int ld = (m + 1) & ~1;
    const double eps = a.eps;
    double2* sh_part = reinterpret_cast<double2*>(dyn);
    double* sh_w = dyn + 2 * LARS_THREADS;
    double* sh_x = sh_w + m;
    double* sh_r = sh_x + m;
    double* sh_t = sh_r + m;       // R^{-T} sgn
    double* sh_gi1 = sh_t + m;     // R^{-1} R^{-T} sgn
    int* sh_act = reinterpret_cast<int*>(sh_gi1 + m);
    // thread groups for the mat-vec over the active set: JT2 threads along j (two columns each), G groups along i
    int JT2 = 32;
    while (2 * JT2 < m && JT2 < LARS_THREADS) JT2 *= 2;
    const int G = LARS_THREADS / JT2;
    double* __restrict__ S = a.S;
    double* b = a.vec + 0 * (int64_t)m;       // sign(b0)
    double* absb = a.vec + 1 * (int64_t)m;    // |b0|
    double* Cvec = a.vec + 2 * (int64_t)m;
    double* beta = a.vec + 3 * (int64_t)m;    // current (scaled) coefficients
    double* u = a.vec + 4 * (int64_t)m;       // Sigma[:,active] w
    double* zt = a.vec + 5 * (int64_t)m;      // lasso crossing distances, by active position
    double* sgn = a.vec + 6 * (int64_t)m;     // by active position
    double* a12 = a.vec + 9 * (int64_t)m;
    int* active = a.ivec + 0 * (int64_t)m;    // active list (variable ids)
    int* state = a.ivec + 1 * (int64_t)m;     // 0 inactive, 1 active, 2 ignored
    int* dropf = a.ivec + 2 * (int64_t)m;     // by active position
    const LarsFactor fac{S, a.Rinv, a.RinvT, ld, active, sh_act, sgn, sh_t, sh_gi1, sh_x, sh_r, red};

    LARS_PROF_DECL;
    // ---- prologue: intercept Schur complement (lsa.py:98-104) and rescaling (lsa.py:108-109)
    double a11 = 1.0, beta0c = 0.0;
    if (a.intercept) {
        a11 = a.Sigma0[0];
        for (int j = tid; j < m; j += LARS_THREADS) a12[j] = a.Sigma0[(int64_t)(j + 1) * a.lds0];
    }
    for (int j = tid; j < m; j += LARS_THREADS) {
        const double v = a.b0[j + off];
        absb[j] = fabs(v);
        const double sb = (v > 0.0) ? 1.0 : ((v < 0.0) ? -1.0 : 0.0);
        b[j] = sb;
        sh_w[j] = sb;
        sh_act[j] = j;
        beta[j] = 0.0;
        state[j] = 0;
    }
    __syncthreads();
    if (a.intercept) {
        double part = 0.0;
        for (int j = tid; j < m; j += LARS_THREADS) part += a12[j] * a.b0[j + 1];
        beta0c = block_sum(part, red) / a11;
    }
    for (int64_t e = tid; e < (int64_t)m * ld; e += LARS_THREADS) {
        const int i = (int)(e / ld), j = (int)(e % ld);
        double v = 0.0;
        if (j < m) {
            v = a.Sigma0[(int64_t)(i + off) * a.lds0 + (j + off)];
            if (a.intercept) v -= a12[i] * a12[j] / a11;
            v = absb[i] * v * absb[j];
        }
        S[e] = v;
    }
    __syncthreads();
    // Cvec = b' Sigma  (lsa.py:114): S is symmetric up to rounding, so use rows: Cvec[j] = sum_i b[i] S[i][j]
    sym_matvec(S, ld, m, m, sh_w, sh_act, Cvec, sh_part, JT2, G);
    int max_steps = a.max_steps > 0 ? a.max_steps : 8 * m;
    // path row 0, and Cmax of the first step (lsa.py:128-129: max |Cvec| over the non-active variables)
    double Cmax;
    {
        double rss[1] = {0.0}, cm[1] = {0.0};
        for (int j = tid; j < m; j += LARS_THREADS) {
            a.beta_path[j] = 0.0;
            const double c = Cvec[j];
            rss[0] += b[j] * c;
            cm[0] = fmax(cm[0], fabs(c));
        }
        if (tid == 0) { sh_i[0] = m; sh_i[2] = 0; }
        block_reduce2(rss, WaveOpSum(), cm, WaveOpMax(), red);
        Cmax = cm[0];
        if (tid == 0) {
            a.aic[0] = rss[0]; a.bic[0] = rss[0];
            a.beta0[0] = a.intercept ? beta0c : 0.0;
        }
    }
    int na = 0, k = 0;
    bool had_drops = false;
    double tsq = 0.0;         // |R^{-T} sgn|^2 = sgn' Gi1 = 1/A^2, carried with the factor
    LARS_TICK(0);
    while (k < max_steps && na < m) {
        ++k;
        if (!had_drops) {
            // ---- new variables, in increasing index order (lsa.py:130-149).  One scan finds the first candidate
            // and counts them all (sh_i[0] = m, sh_i[2] = 0 on entry); further scans only when there are ties.
            int start = 0;
            while (true) {
                int cand = m, ncand = 0;
                for (int j = start + tid; j < m; j += LARS_THREADS)
                    if (state[j] == 0 && fabs(Cvec[j]) >= Cmax - eps) { cand = min(cand, j); ++ncand; }
                if (ncand > 0) { atomicMin(&sh_i[0], cand); atomicAdd(&sh_i[2], ncand); }
                __syncthreads();
                const int inew = sh_i[0], left = sh_i[2] - 1;
                if (inew >= m) break;
                const double c = Cvec[inew];
                const int grew = append_column(fac, na, inew, (c > 0.0) ? 1.0 : ((c < 0.0) ? -1.0 : 0.0), eps, tsq);
                if (tid == 0) {
                    state[inew] = grew ? 1 : 2;      // 2: machine-singular, ignored (lsa.py:139-144)
                    sh_i[0] = m; sh_i[2] = 0;
                }
                if (grew) ++na;
                start = inew + 1;
                __syncthreads();
                if (left <= 0) break;
            }
        }
        if (na == 0) break;   // nothing could enter (degenerate input)
        LARS_TICK(2);
        // ---- equiangular direction from the carried Gi1 = R^{-1} R^{-T} Sign (lsa.py:151-153)
        const double A = 1.0 / sqrt(tsq);
        for (int i = tid; i < na; i += LARS_THREADS) sh_w[i] = A * sh_gi1[i];
        __syncthreads();
        LARS_TICK(3);
        // ---- u = Sigma[:,active] w
        sym_matvec(S, ld, m, na, sh_w, sh_act, u, sh_part, JT2, G);
        LARS_TICK(4);
        // ---- step length (lsa.py:154-162) and lasso modification (lsa.py:164-173)
        double gamhat = Cmax / A;
        double mins[2] = {INFINITY, INFINITY};
        if (na < m) {
            for (int j = tid; j < m; j += LARS_THREADS) {
                if (state[j] != 0) continue;
                const double c = Cvec[j], aj = u[j];
                const double g1 = (Cmax - c) / (A - aj);
                const double g2 = (Cmax + c) / (A + aj);
                if (g1 > eps) mins[0] = fmin(mins[0], g1);
                if (g2 > eps) mins[0] = fmin(mins[0], g2);
            }
        }
        if (a.type == 1) {
            for (int i = tid; i < na; i += LARS_THREADS) {
                const double z = -beta[sh_act[i]] / sh_w[i];
                zt[i] = z;
                if (z > eps) mins[1] = fmin(mins[1], z);
            }
        }
        block_reduce(mins, red, WaveOpMin());
        gamhat = fmin(mins[0], gamhat);
        had_drops = false;
        if (a.type == 1 && mins[1] < gamhat) {
            gamhat = mins[1];
            had_drops = true;
            for (int i = tid; i < na; i += LARS_THREADS) dropf[i] = (zt[i] == gamhat) ? 1 : 0;
        }
        // ---- move (lsa.py:175-177)
        for (int i = tid; i < na; i += LARS_THREADS) beta[sh_act[i]] += gamhat * sh_w[i];
        for (int j = tid; j < m; j += LARS_THREADS) Cvec[j] -= gamhat * u[j];
        __syncthreads();
        LARS_TICK(5);
        // ---- drops (lsa.py:179-186)
        if (had_drops) {
            for (int i = tid; i < na; i += LARS_THREADS)
                if (dropf[i]) { beta[sh_act[i]] = 0.0; state[sh_act[i]] = 0; }
            __syncthreads();
            if (tid == 0) {
                int q = 0, first = na;
                for (int i = 0; i < na; ++i) {
                    if (!dropf[i]) { active[q] = active[i]; sgn[q] = sgn[i]; ++q; }
                    else if (first == na) first = i;
                }
                sh_i[1] = q; sh_i[3] = first;
            }
            __syncthreads();
            const int keep = sh_i[1], first = sh_i[3];
            // The factor of the positions before the first dropped one is unchanged: they keep their rows of R^{-1} and
            // their entries of t; Gi1 there loses the later positions' terms (one triangular mat-vec over [i, first)).
            // The positions from `first` on are appended again.
            double tpart = 0.0;
            for (int i = tid; i < first; i += LARS_THREADS) tpart = fma(fac.t[i], fac.t[i], tpart);
            tsq = block_sum(tpart, red);
            tri_matvec<false>(fac.Rinv, ld, first, fac.t, [&](int i, double sum) { fac.gi1[i] = sum; });
            __syncthreads();
            for (int i = first; i < keep; ++i) append_column(fac, i, active[i], sgn[i], 0.0, tsq);
            na = keep;
        }
        LARS_TICK(6);
        // ---- record the path point: un-scaled beta (lsa.py:194-201), RSS, dof, AIC/BIC (:190-210)
        // and Cmax of the next step (lsa.py:128-129)
        double rec[3] = {0.0, 0.0, 0.0}, cm[1] = {0.0};      // RSS, dof, a12 . beta
        for (int j = tid; j < m; j += LARS_THREADS) {
            const double bj = beta[j], cj = Cvec[j];
            const double ub = absb[j] * bj;
            a.beta_path[(int64_t)k * m + j] = ub;
            rec[0] += (b[j] - bj) * cj;
            if (fabs(ub) > eps) rec[1] += 1.0;
            if (a.intercept) rec[2] += a12[j] * ub;
            if (state[j] != 1) cm[0] = fmax(cm[0], fabs(cj));
        }
        block_reduce2(rec, WaveOpSum(), cm, WaveOpMax(), red);
        Cmax = cm[0];
        if (tid == 0) {
            a.aic[k] = rec[0] + 2.0 * rec[1];
            a.bic[k] = rec[0] + log(a.n) * rec[1];
            a.beta0[k] = a.intercept ? beta0c - rec[2] / a11 : 0.0;
        }
        __syncthreads();
        LARS_TICK(7);
    }
#ifdef DLSA_LARS_PROF
    if (tid == 0) {
        printf("LARS_PROF p=%d steps=%d us: prologue %.0f select %.0f direction %.0f u %.0f step %.0f drops %.0f record %.0f |",
               p, k, lars_prof_t[0] * 0.01, lars_prof_t[2] * 0.01, lars_prof_t[3] * 0.01, lars_prof_t[4] * 0.01, lars_prof_t[5] * 0.01,
               lars_prof_t[6] * 0.01, lars_prof_t[7] * 0.01);
        printf(" append: gather %.0f r %.0f reduce %.0f column %.0f tail %.0f\n", lars_prof_t[8] * 0.01, lars_prof_t[9] * 0.01,
               lars_prof_t[10] * 0.01, lars_prof_t[11] * 0.01, lars_prof_t[12] * 0.01);
    }
#endif
    if (tid == 0) *a.n_steps = k;
}


// ------------------------------------------------------------------------------------------------------------
// Multi-workgroup variant.  One CU streams ~110 GB/s from L2, which bounds the single-workgroup kernel from
// p ~ 500 up; here G workgroups (one per CU, cooperative launch) share the O(k^2) and O(kp) mat-vecs of a step and
// replicate everything that is O(p): every workgroup keeps Cvec, beta, t, the active list ... in its own LDS, runs the
// same instructions on the same numbers and so takes the same decisions -- no broadcast of scalars is ever needed.
// Position i of the active list (row i of R^{-1} and of its transpose, Gi1[i], w[i]) belongs to workgroup i % G.
// A step has two grid barriers (~1.3 us for 16 workgroups on gfx950, agent-scope release/acquire on a counter):
//   append:  x = Sigma[new, active] (replicated gather);  r_i = (R^{-T} x)_i by the row owners -> rbuf
//            -- barrier 1 --   everyone reads r; r.r and r.t give r_kk and t_k (replicated)
//            c_i = -(R^{-1} r)_i / r_kk, Gi1_i += c_i t_k, w_i = A Gi1_i by the row owners -> wbuf
//   u:       partial Sigma[own rows of the active set, :]' w_own -> upart[g][:]
//            -- barrier 2 --   everyone reads w and sums the G partial vectors in a fixed order
//   step length, lasso crossing, move, drops, Cmax: replicated in LDS;  workgroup 0 writes the path record.
// A further append without an intervening barrier (ties, the rebuild after a lasso drop) first waits for the
// column writes of the previous one.  The workgroups are left where the dispatcher puts them (round-robin over the
// XCDs): confining them to one XCD makes the exchanged vectors L2 hits but was measured slower overall
// (p=1000, 16 workgroups: 29.9 vs 22.2 ms) -- eight L2s hold more of Sigma and R^{-1} than one.

// bar[0] counts arrivals, bar[1] is the ABORT word.  The barrier is hand-rolled and the launch is a plain one (lars_run), so nothing
// guarantees that all G workgroups are resident together: with other kernels holding CUs (more concurrent grid-LARS calls than the
// chip has room for, a CU-masked or shared device) a resident workgroup could wait for one that is queued behind it.  Every wait
// is therefore BOUNDED: a workgroup that has waited `timeout` ticks of the 100 MHz wall clock sets bar[1] and leaves; every
// workgroup that sees bar[1] leaves (a late starter at its first barrier); the kernel reports n_steps = -1 and the host reruns the
// path on the single-workgroup kernel (dlsa_lars_lsa_f64).  Polling is a relaxed agent-scope load + s_sleep with ONE acquire
// after the match (an acquire per poll invalidates the CU's L1 every iteration: 2-3x slower per hop, guide G16).
__device__ __forceinline__ bool grid_barrier(unsigned* bar, unsigned nwg, unsigned& phase, long long timeout) {
    __shared__ int gb_ok;
    __syncthreads();                 // this workgroup's global stores have been issued by every wave
    if (threadIdx.x == 0) {
        ++phase;
        const unsigned target = phase * nwg;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (the write-back has completed before the arrival is visible)
        __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        bool ok = true;
        const long long t0 = wall_clock64();
        unsigned spins = 0;
        while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if ((spins++ & 255u) == 0u) {        // (checked at the first failed poll too: a timeout of one tick aborts at once -- the tests' way in)
                if (__hip_atomic_load(bar + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { ok = false; break; }
                if (wall_clock64() - t0 > timeout) {
                    __hip_atomic_store(bar + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ok = false;
                    break;
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        gb_ok = ok ? 1 : 0;
    }
    __syncthreads();
    return gb_ok != 0;
}

// y_i over the stored part of row i for the rows i = first, first + stride, ... < n this workgroup owns: one wave per
// row, four loads per lane and trip at constant offsets (zero triangle + slack: only x is masked).
template <bool LOWER, typename Emit>
__device__ __forceinline__ void tri_matvec_rows(const double* __restrict__ M, int ld, int n, const double* xs, int first,
                                                int stride, Emit&& emit) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int i = first + wave * stride; i < n; i += LARS_WAVES * stride) {
        const int lo = (LOWER ? 0 : i) + lane, hi = LOWER ? i + 1 : n;
        const double* __restrict__ row = M + (int64_t)i * ld + lo;
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        for (int l = lo, o = 0; l < hi; l += 256, o += 256) {
            const double m0 = row[o], m1 = row[o + 64], m2 = row[o + 128], m3 = row[o + 192];
            const double x0 = xs[l];
            const double x1 = l + 64 < hi ? xs[min(l + 64, hi - 1)] : 0.0;
            const double x2 = l + 128 < hi ? xs[min(l + 128, hi - 1)] : 0.0;
            const double x3 = l + 192 < hi ? xs[min(l + 192, hi - 1)] : 0.0;
            s0 = fma(m0, x0, s0); s1 = fma(m1, x1, s1); s2 = fma(m2, x2, s2); s3 = fma(m3, x3, s3);
        }
        const double s = wave_sum((s0 + s1) + (s2 + s3));
        if (lane == 0) emit(i, s);
    }
}

struct LarsGridState {
    const double* __restrict__ S;
    double* __restrict__ Rinv;
    double* __restrict__ RinvT;
    double* rbuf;
    unsigned* bar;
    long long bar_timeout;
    int ld, g, G;
    // LDS, replicated in every workgroup
    int* act; float* sgn; double* t; double* xu; double* rw;
    double* gi1;                       // LDS; entries of the positions this workgroup owns
    double* red;
};

// Append variable `inew` with sign `sg` at position na.  All workgroups call it with the same arguments and get the
// same answer (1 = appended, 0 = rank did not grow, -1 = the launch was aborted at a grid barrier); `dirty` says that column writes
// of an earlier append have not been followed by a grid barrier yet.
__device__ int append_column_grid(const LarsGridState& f, int na, int inew, double sg, double eps, double& tsq,
                                  unsigned& phase, bool& dirty) {
    const int tid = threadIdx.x;
    LARS_TICK(0);
    if (dirty) { if (!grid_barrier(f.bar, f.G, phase, f.bar_timeout)) return -1; dirty = false; }
    const double* __restrict__ srow = f.S + (int64_t)inew * f.ld;
    for (int i = tid; i < na; i += LARS_THREADS) f.xu[i] = srow[f.act[i]];
    __syncthreads();
    LARS_TICK(1);
    double* __restrict__ rbuf = f.rbuf;
    tri_matvec_rows<true>(f.RinvT, f.ld, na, f.xu, f.g, f.G, [&](int i, double s) { rbuf[i] = s; });
    LARS_TICK(2);
    if (!grid_barrier(f.bar, f.G, phase, f.bar_timeout)) return -1;      // aborted launch (see grid_barrier)
    LARS_TICK(3);
    double acc[2] = {0.0, 0.0};
    for (int i = tid; i < na; i += LARS_THREADS) {
        const double ri = rbuf[i];
        f.rw[i] = ri;
        acc[0] = fma(ri, ri, acc[0]);
        acc[1] = fma(ri, f.t[i], acc[1]);
    }
    block_reduce(acc, f.red, WaveOpSum());          // (its barriers also publish rw)
    LARS_TICK(4);
    double rpp = srow[inew] - acc[0];
    if (na > 0 && rpp <= eps) return 0;  // rank did not grow: caller records an "ignore"
    rpp = sqrt(rpp);
    const double tn = (sg - acc[1]) / rpp;
    tsq = fma(tn, tn, tsq);
    double* __restrict__ Rinv = f.Rinv;
    double* __restrict__ RinvT = f.RinvT;
    const int ld = f.ld;
    tri_matvec_rows<false>(Rinv, ld, na, f.rw, f.g, f.G, [&](int i, double s) {
        const double c = -s / rpp;
        Rinv[(int64_t)i * ld + na] = c;
        RinvT[(int64_t)na * ld + i] = c;
        f.gi1[i] = fma(c, tn, f.gi1[i]);
    });
    if (tid == 0) {
        const double c = 1.0 / rpp;
        if (na % f.G == f.g) {
            Rinv[(int64_t)na * ld + na] = c;
            RinvT[(int64_t)na * ld + na] = c;
            f.gi1[na] = c * tn;
        }
        f.t[na] = tn;
        f.sgn[na] = (float)sg;
        f.act[na] = inew;
    }
    dirty = true;
    __syncthreads();
    LARS_TICK(5);
    return 1;
}

// an aborted launch (grid_barrier timed out somewhere in the grid): every workgroup reports -1 steps and leaves
#define LARS_GRID_ABORT() do { if (threadIdx.x == 0) *a.n_steps = -1; return; } while (0)
#define LARS_GRID_SYNC() do { if (!grid_barrier(a.bar, G, phase, a.bar_timeout)) LARS_GRID_ABORT(); } while (0)
__global__ __launch_bounds__(LARS_THREADS) void lars_grid_kernel(LarsArgs a) {
    __shared__ double red[4 * LARS_WAVES];
    __shared__ int sh_i[4];
    extern __shared__ __attribute__((aligned(16))) double dyn[];
    const int tid = threadIdx.x;
    const int g = blockIdx.x, G = gridDim.x;
    const int p = a.p;
    const int off = a.intercept ? 1 : 0;
    const int m = p - off;
    const int ld = (m + 1) & ~1;
    const int mo = m / G + 2;                        // own-list capacity
    const double eps = a.eps;
    // LDS: sh_part[2 T] | Cvec, beta, xu, rw, t, gi1 [m] | own_w [mo] | sgn(f32), state, act [m] | own_act [mo]
    double2* sh_part = reinterpret_cast<double2*>(dyn);
    double* Cvec = dyn + 2 * LARS_THREADS;
    double* beta = Cvec + m;
    double* xu = beta + m;          // x of the append, then u of the step
    double* rw = xu + m;            // r of the append, then w of the step
    double* tv = rw + m;
    double* gi1 = tv + m;
    double* own_w = gi1 + m;
    float* sgn = reinterpret_cast<float*>(own_w + mo);
    int* state = reinterpret_cast<int*>(sgn + m);
    int* act = state + m;
    int* own_act = act + m;
    int JT2 = 32;
    while (2 * JT2 < m && JT2 < LARS_THREADS) JT2 *= 2;
    const int GS = LARS_THREADS / JT2;
    double* __restrict__ S = a.S;
    double* b = a.vec + 0 * (int64_t)m;       // sign(b0)      (global: written and read by workgroup 0 only)
    double* absb = a.vec + 1 * (int64_t)m;    // |b0|
    double* a12 = a.vec + 9 * (int64_t)m;
    double* upart = a.upart + (int64_t)g * ld;
    unsigned phase = 0;
    bool dirty = false;
    LARS_PROF_DECL;
    const LarsGridState fac{S, a.Rinv, a.RinvT, a.rbuf, a.bar, a.bar_timeout, ld, g, G, act, sgn, tv, xu, rw, gi1, red};

    // ---- prologue: intercept Schur complement (lsa.py:98-104) and rescaling (lsa.py:108-109).
    // |b0|, a12 and sign(b0) are staged in LDS (rw, tv, beta) for the build of S; workgroup 0 keeps global copies.
    double a11 = 1.0, beta0c = 0.0;
    if (a.intercept) a11 = a.Sigma0[0];
    for (int j = tid; j < m; j += LARS_THREADS) {
        const double v = a.b0[j + off];
        const double sb = (v > 0.0) ? 1.0 : ((v < 0.0) ? -1.0 : 0.0);
        const double a12j = a.intercept ? a.Sigma0[(int64_t)(j + 1) * a.lds0] : 0.0;
        rw[j] = fabs(v); tv[j] = a12j; beta[j] = sb;
        if (g == 0) { absb[j] = fabs(v); b[j] = sb; a12[j] = a12j; }
    }
    __syncthreads();
    if (a.intercept) {
        double part = 0.0;
        for (int j = tid; j < m; j += LARS_THREADS) part += tv[j] * a.b0[j + 1];
        beta0c = block_sum(part, red) / a11;
    }
    int nown = 0;
    for (int i = g; i < m; i += G, ++nown) {
        for (int j = tid; j < ld; j += LARS_THREADS) {
            double v = 0.0;
            if (j < m) {
                v = a.Sigma0[(int64_t)(i + off) * a.lds0 + (j + off)];
                if (a.intercept) v -= tv[i] * tv[j] / a11;
                v = rw[i] * v * rw[j];
            }
            S[(int64_t)i * ld + j] = v;
        }
        if (tid == 0) { own_act[nown] = i; own_w[nown] = beta[i]; }
    }
    __syncthreads();
    // Cvec = b' Sigma (lsa.py:114) from the rows each workgroup just wrote
    sym_matvec(S, ld, m, nown, own_w, own_act, upart, sh_part, JT2, GS);
    LARS_GRID_SYNC();
    for (int j = tid; j < m; j += LARS_THREADS) {
        double s = 0.0;
        for (int q = 0; q < G; ++q) s += a.upart[(int64_t)q * ld + j];
        Cvec[j] = s;
    }
    __syncthreads();
    int max_steps = a.max_steps > 0 ? a.max_steps : 8 * m;
    // path row 0, and Cmax of the first step (lsa.py:128-129: max |Cvec| over the non-active variables)
    double Cmax;
    {
        double rss[1] = {0.0}, cm[1] = {0.0};
        for (int j = tid; j < m; j += LARS_THREADS) {
            const double c = Cvec[j];
            rss[0] += beta[j] * c;                   // beta still holds sign(b0)
            cm[0] = fmax(cm[0], fabs(c));
        }
        if (tid == 0) { sh_i[0] = m; sh_i[2] = 0; }
        block_reduce2(rss, WaveOpSum(), cm, WaveOpMax(), red);
        Cmax = cm[0];
        for (int j = tid; j < m; j += LARS_THREADS) {
            beta[j] = 0.0;
            state[j] = 0;
            if (g == 0) a.beta_path[j] = 0.0;
        }
        if (g == 0 && tid == 0) {
            a.aic[0] = rss[0]; a.bic[0] = rss[0];
            a.beta0[0] = a.intercept ? beta0c : 0.0;
        }
        __syncthreads();
    }
    int na = 0, k = 0;
    bool had_drops = false;
    double tsq = 0.0;
    while (k < max_steps && na < m) {
        ++k;
        if (!had_drops) {
            // ---- new variables, in increasing index order (lsa.py:130-149)
            int start = 0;
            while (true) {
                int cand = m, ncand = 0;
                for (int j = start + tid; j < m; j += LARS_THREADS)
                    if (state[j] == 0 && fabs(Cvec[j]) >= Cmax - eps) { cand = min(cand, j); ++ncand; }
                if (ncand > 0) { atomicMin(&sh_i[0], cand); atomicAdd(&sh_i[2], ncand); }
                __syncthreads();
                const int inew = sh_i[0], left = sh_i[2] - 1;
                if (inew >= m) break;
                const double c = Cvec[inew];
                const int grew = append_column_grid(fac, na, inew, (c > 0.0) ? 1.0 : ((c < 0.0) ? -1.0 : 0.0), eps, tsq, phase, dirty);
                if (grew < 0) LARS_GRID_ABORT();
                if (tid == 0) {
                    state[inew] = grew ? 1 : 2;      // 2: machine-singular, ignored (lsa.py:139-144)
                    sh_i[0] = m; sh_i[2] = 0;
                }
                if (grew) ++na;
                start = inew + 1;
                __syncthreads();
                if (left <= 0) break;
            }
        }
        if (na == 0) break;   // nothing could enter (degenerate input)
        LARS_TICK(0);
        // ---- equiangular direction (lsa.py:151-153): w_i = A Gi1_i for the positions this workgroup owns
        const double A = 1.0 / sqrt(tsq);
        nown = (na > g) ? (na - g + G - 1) / G : 0;
        for (int q = tid; q < nown; q += LARS_THREADS) {
            const int i = g + q * G;
            const double wi = A * gi1[i];
            own_w[q] = wi;
            own_act[q] = act[i];
            a.wbuf[i] = wi;
        }
        __syncthreads();
        // ---- partial u = Sigma[own active rows, :]' w_own
        sym_matvec(S, ld, m, nown, own_w, own_act, upart, sh_part, JT2, GS);
        LARS_TICK(6);
        LARS_GRID_SYNC();
        LARS_TICK(7);
        dirty = false;
        for (int i = tid; i < na; i += LARS_THREADS) rw[i] = a.wbuf[i];
        for (int j2 = tid; 2 * j2 < m; j2 += LARS_THREADS) {
            const double2* src = reinterpret_cast<const double2*>(a.upart + 2 * j2);
            double2 sum = {0.0, 0.0};
            // (eight loads in flight per trip: as a plain loop every partial waited for its own L2 round trip -- 32 in a row at
            // p = 2000, 10 of a step's 37 us; the additions keep their order)
            for (int q0 = 0; q0 < G; q0 += 8) {
                double2 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = (q0 + u < G) ? src[(int64_t)(q0 + u) * (ld / 2)] : double2{0.0, 0.0};
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (q0 + u < G) { sum.x += v[u].x; sum.y += v[u].y; }
            }
            xu[2 * j2] = sum.x;
            if (2 * j2 + 1 < m) xu[2 * j2 + 1] = sum.y;
        }
        __syncthreads();
        LARS_TICK(8);
        // ---- step length (lsa.py:154-162) and lasso modification (lsa.py:164-173)
        double gamhat = Cmax / A;
        double mins[2] = {INFINITY, INFINITY};
        if (na < m) {
            for (int j = tid; j < m; j += LARS_THREADS) {
                if (state[j] != 0) continue;
                const double c = Cvec[j], aj = xu[j];
                const double g1 = (Cmax - c) / (A - aj);
                const double g2 = (Cmax + c) / (A + aj);
                if (g1 > eps) mins[0] = fmin(mins[0], g1);
                if (g2 > eps) mins[0] = fmin(mins[0], g2);
            }
        }
        if (a.type == 1) {
            for (int i = tid; i < na; i += LARS_THREADS) {
                const double z = -beta[act[i]] / rw[i];
                if (z > eps) mins[1] = fmin(mins[1], z);
            }
        }
        block_reduce(mins, red, WaveOpMin());
        gamhat = fmin(mins[0], gamhat);
        had_drops = false;
        if (a.type == 1 && mins[1] < gamhat) {
            gamhat = mins[1];
            had_drops = true;
        }
        // ---- move (lsa.py:175-177); a crossing variable (lsa.py:179-186) leaves: beta = 0, state = inactive
        for (int i = tid; i < na; i += LARS_THREADS) {
            const int v = act[i];
            const double z = -beta[v] / rw[i];
            if (had_drops && z == gamhat) { beta[v] = 0.0; state[v] = 0; act[i] = -1 - v; }
            else beta[v] += gamhat * rw[i];
        }
        for (int j = tid; j < m; j += LARS_THREADS) Cvec[j] -= gamhat * xu[j];
        __syncthreads();
        if (had_drops) {
            if (tid == 0) {
                int q = 0, first = na;
                for (int i = 0; i < na; ++i) {
                    if (act[i] >= 0) { act[q] = act[i]; sgn[q] = sgn[i]; ++q; }
                    else if (first == na) first = i;
                }
                sh_i[1] = q; sh_i[3] = first;
            }
            __syncthreads();
            const int keep = sh_i[1], first = sh_i[3];
            // positions before the first dropped one keep their rows of R^{-1} and their entries of t (their Gi1 loses the
            // later positions' terms: one triangular mat-vec by the row owners); the positions from `first` on are appended again
            double part = 0.0;
            for (int i = tid; i < first; i += LARS_THREADS) part = fma(tv[i], tv[i], part);
            tsq = block_sum(part, red);
            tri_matvec_rows<false>(a.Rinv, ld, first, tv, g, G, [&](int i, double sum) { gi1[i] = sum; });
            __syncthreads();
            for (int i = first; i < keep; ++i)
                if (append_column_grid(fac, i, act[i], (double)sgn[i], 0.0, tsq, phase, dirty) < 0) LARS_GRID_ABORT();
            na = keep;
        }
        LARS_TICK(9);
        // ---- Cmax of the next step; workgroup 0 records the path point: un-scaled beta (lsa.py:194-201), RSS, dof,
        // AIC/BIC (:190-210)
        double rec[3] = {0.0, 0.0, 0.0}, cm[1] = {0.0};      // RSS, dof, a12 . beta
        for (int j = tid; j < m; j += LARS_THREADS) {
            const double cj = Cvec[j];
            if (state[j] != 1) cm[0] = fmax(cm[0], fabs(cj));
            if (g == 0) {
                const double bj = beta[j];
                const double ub = absb[j] * bj;
                a.beta_path[(int64_t)k * m + j] = ub;
                rec[0] += (b[j] - bj) * cj;
                if (fabs(ub) > eps) rec[1] += 1.0;
                if (a.intercept) rec[2] += a12[j] * ub;
            }
        }
        block_reduce2(rec, WaveOpSum(), cm, WaveOpMax(), red);
        Cmax = cm[0];
        if (g == 0 && tid == 0) {
            a.aic[k] = rec[0] + 2.0 * rec[1];
            a.bic[k] = rec[0] + log(a.n) * rec[1];
            a.beta0[k] = a.intercept ? beta0c - rec[2] / a11 : 0.0;
        }
        __syncthreads();
        LARS_TICK(10);
    }
#ifdef DLSA_LARS_PROF
    if (g == 0 && tid == 0)
        printf("LARS_GRID_PROF p=%d G=%d steps=%d us: select %.0f gather %.0f r %.0f barrier1 %.0f reduce %.0f column %.0f | w+upartial %.0f barrier2 %.0f ureduce %.0f step %.0f record %.0f\n",
               p, G, k, lars_prof_t[0] * 0.01, lars_prof_t[1] * 0.01, lars_prof_t[2] * 0.01, lars_prof_t[3] * 0.01, lars_prof_t[4] * 0.01,
               lars_prof_t[5] * 0.01, lars_prof_t[6] * 0.01, lars_prof_t[7] * 0.01, lars_prof_t[8] * 0.01, lars_prof_t[9] * 0.01,
               lars_prof_t[10] * 0.01);
#endif
    if (g == 0 && tid == 0) *a.n_steps = k;
}

// Workgroups for the path at width p (measured: p=200 one workgroup 2.6 ms / four 2.9; p=300 5.1 / 4.7; p=500 8.3 ms
// at 8; p=1000 22 ms at 16; p=2000 74 ms at 32 (round 5: 66, see the reduce of u); the grid kernel with ONE workgroup is slower than lars_kernel:
// 4.0 vs 2.6 ms at p=200).  DLSA_LARS_WGS overrides, 1..LARS_MAX_WGS.
static int lars_workgroups(int p) {
    int wgs = p < 256 ? 1 : (p < 384 ? (LARS_THREADS == 512 ? 8 : 4) : (p < 768 ? 8 : (p < 1536 ? 16 : 32)));   // (512-thread build, p = 260: 3.78 ms at 4, 3.59 at 8)
    if (const char* e = kernel_knob("DLSA_LARS_WGS")) wgs = atoi(e);
    return std::max(1, std::min(wgs, LARS_MAX_WGS));
}


// launch of this build's kernels (the C ABI entry below picks the build by p)
int lars_run(LarsArgs& a, int p, int intercept, hipStream_t s, int max_wgs, int* wgs_used) {
    const size_t mm = (size_t)(p - (intercept ? 1 : 0));
    int wgs = std::min(lars_workgroups(p), std::max(1, max_wgs));
    if (wgs > kNumCU) wgs = 1;                   // (the spinning barrier needs a CU per workgroup)
    // the grid kernel replicates more vectors in LDS; beyond its limit (m ~ 2440) the single-workgroup kernel still fits
    if (wgs > 1 && (size_t)LARS_THREADS * 16 + mm * 60 + (mm / wgs + 2) * 12 + 64 > (size_t)kLdsBytes) wgs = 1;
    if (wgs > 1) {
        const size_t shm = (size_t)LARS_THREADS * 16 + mm * 60 + (mm / wgs + 2) * 12 + 64;
        if (shm > 48 * 1024)
            DLSA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(lars_grid_kernel),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
        DLSA_HIP_CHECK(hipMemsetAsync(a.bar, 0, 256, s));
        // A PLAIN launch: at most 32 workgroups on 256 CUs are resident together as soon as the CUs they need are free (kernels of
        // other streams end without waiting for this one).  hipLaunchCooperativeKernel would check co-residency at launch, but
        // (measured) every HIP stream CREATED after a process's first cooperative launch is serialised with the others -- the
        // partition chains of a later fit (irls.hip) then lose their overlap (config 4 structured: 14.4 -> 22.2 ms) -- and it
        // costs 15-19 us of host time per call.  What the plain launch cannot promise -- several grid kernels each partly
        // resident and waiting for their queued workgroups -- is covered twice: grid launches of one process are serialised
        // (g_lars_grid_mu in dlsa_lars_lsa_f64) and every barrier wait is bounded (grid_barrier: abort + single-workgroup rerun).
        // (dlsa_kernel_options.cooperative = 1 asks for the cooperative launch all the same)
        void* kargs[] = {(void*)&a};
        if (launch_cooperative(reinterpret_cast<const void*>(lars_grid_kernel), dim3(wgs), dim3(LARS_THREADS), kargs, shm, s) != hipSuccess)
            hipLaunchKernelGGL(lars_grid_kernel, dim3(wgs), dim3(LARS_THREADS), shm, s, a);
        DLSA_HIP_CHECK(hipGetLastError());
    } else {
        const size_t shm = (size_t)LARS_THREADS * 16 + mm * 44 + 64;
        DLSA_REQUIRE(shm <= (size_t)kLdsBytes, "lars_lsa: p=%d needs %zu bytes of LDS (limit %d)", p, shm, kLdsBytes);
        if (shm > 48 * 1024)
            DLSA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(lars_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
        hipLaunchKernelGGL(lars_kernel, dim3(1), dim3(LARS_THREADS), shm, s, a);
        DLSA_HIP_CHECK(hipGetLastError());
    }
    if (wgs_used) *wgs_used = wgs;
    return DLSA_OK;
}

}  // namespace LARS_NS
namespace lars_t512 { int lars_run(LarsArgs& a, int p, int intercept, hipStream_t s, int max_wgs, int* wgs_used); }
namespace lars_t1024 { int lars_run(LarsArgs& a, int p, int intercept, hipStream_t s, int max_wgs, int* wgs_used); }

}  // namespace dlsa

#ifndef DLSA_LARS_SECONDARY
extern "C" {

size_t dlsa_lars_workspace_bytes(int p) {
    if (p <= 0) return 0;
    const size_t m = (size_t)p, ld = (m + 1) & ~(size_t)1;
    return dlsa::align_up(m * ld * 8, 256) + dlsa::align_up((m * ld + dlsa::LARS_SLACK) * 8, 256) * 2 + dlsa::align_up(12 * m * 8, 256) +
           dlsa::align_up(m * 8, 256) * 2 + dlsa::align_up(dlsa::LARS_MAX_WGS * ld * 8, 256) + 256 + dlsa::align_up(4 * m * 4, 256) + 512;
}

int dlsa_lars_lsa_f64(const double* Sigma0, int64_t lds, const double* b0, int p, int intercept, double n,
                      int type, double eps, int max_steps, double* beta_path, double* beta0, double* aic,
                      double* bic, int* n_steps_host, void* ws, size_t ws_bytes, void* stream) {
    using namespace dlsa;
    DLSA_REQUIRE(Sigma0 && b0 && beta_path && beta0 && aic && bic, "lars_lsa: null argument");
    DLSA_REQUIRE(p > (intercept ? 1 : 0) && lds >= p, "lars_lsa: bad shape p=%d lds=%lld", p, (long long)lds);
    DLSA_REQUIRE(type == 0 || type == 1, "lars_lsa: type must be 0 ('lar') or 1 ('lasso')");
    DLSA_REQUIRE(n > 0, "lars_lsa: sample size must be positive");
    if (!ws || ws_bytes < dlsa_lars_workspace_bytes(p) || ((uintptr_t)ws & 255)) {
        set_error("lars_lsa: workspace %zu bytes needed (256-aligned), got %zu", dlsa_lars_workspace_bytes(p), ws_bytes);
        return DLSA_ERR_WORKSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    Arena ar(ws, ws_bytes);
    const size_t m = (size_t)p, ld = (m + 1) & ~(size_t)1;
    LarsArgs a;
    a.Sigma0 = Sigma0; a.b0 = b0; a.lds0 = lds; a.p = p; a.intercept = intercept ? 1 : 0; a.type = type;
    a.max_steps = max_steps; a.n = n; a.eps = eps;
    a.S = (double*)ar.take(m * ld * 8);
    a.Rinv = (double*)ar.take((m * ld + LARS_SLACK) * 8);
    a.RinvT = (double*)ar.take((m * ld + LARS_SLACK) * 8);
    a.vec = (double*)ar.take(12 * m * 8);
    a.ivec = (int*)ar.take(4 * m * 4);
    a.n_steps = (int*)ar.take(256);
    a.rbuf = (double*)ar.take(m * 8);
    a.wbuf = (double*)ar.take(m * 8);
    a.upart = (double*)ar.take((size_t)LARS_MAX_WGS * ld * 8);
    a.bar = (unsigned*)ar.take(256);
    a.beta_path = beta_path; a.beta0 = beta0; a.aic = aic; a.bic = bic;
    // 512-thread workgroups up to p = 768, 1024-thread ones beyond (DLSA_LARS_THREADS=512|1024 forces a build)
    bool small_wg = p <= 768;
    if (const char* e = kernel_knob("DLSA_LARS_THREADS")) small_wg = atoi(e) == 512;
    // ticks of the 100 MHz wall clock a workgroup of the grid kernel waits at one grid barrier (normally microseconds) before it
    // gives the launch up: 0.25 s
    a.bar_timeout = g_lars_barrier_timeout_ticks.load();
    int steps = 0, wgs_used = 1;
    const bool use_c = lars_c_eligible(p, intercept);
    if (!use_c && lars_q_eligible(p, intercept)) {
        // up to 448 variables (1020 when the column-split kernel is switched off): the carried Cholesky rows (lars_q.hip) -- one workgroup, or a few that share the fused pass and meet
        // at a bounded grid barrier (a launch that gave up there is rerun on one workgroup).  Every entry of its matrices is written
        // before it is read, so nothing is cleared.
        for (int attempt = 0; attempt < 2; ++attempt) {
            std::unique_lock<std::mutex> grid_lock(g_lars_grid_mu, std::defer_lock);
            const int max_wgs = attempt == 0 ? LARS_MAX_WGS : 1;
            if (max_wgs > 1 && p - (intercept ? 1 : 0) > 108) grid_lock.lock();
            const int rc = lars_q_run(a, p, intercept, s, max_wgs, &wgs_used);
            if (rc) return rc;
            DLSA_HIP_CHECK(hipMemcpyAsync(&steps, a.n_steps, sizeof(int), hipMemcpyDeviceToHost, s));
            DLSA_HIP_CHECK(hipStreamSynchronize(s));
            if (steps >= 0) break;
            g_lars_grid_aborts.fetch_add(1);
            if (attempt == 1 || wgs_used == 1) {
                set_error("lars_lsa: the path kernel aborted (grid barrier timeout with %d workgroups)", wgs_used);
                return DLSA_ERR_HIP;
            }
        }
        if (n_steps_host) *n_steps_host = steps;
        return DLSA_OK;
    }
    int first_attempt = 0;
    if (use_c) {
        // 449 .. 2044 variables: the carried rows with the pass split by columns over up to 64 workgroups (lars_c.hip); a launch that gave
        // up at its bounded grid barrier is rerun on the single-workgroup kernel below
        std::unique_lock<std::mutex> grid_lock(g_lars_grid_mu);
        const int rc = lars_c_run(a, p, intercept, s, &wgs_used);
        if (rc) return rc;
        DLSA_HIP_CHECK(hipMemcpyAsync(&steps, a.n_steps, sizeof(int), hipMemcpyDeviceToHost, s));
        DLSA_HIP_CHECK(hipStreamSynchronize(s));
        if (steps >= 0) {
            if (n_steps_host) *n_steps_host = steps;
            return DLSA_OK;
        }
        g_lars_grid_aborts.fetch_add(1);
        first_attempt = 1;
    }
    for (int attempt = first_attempt; attempt < 2; ++attempt) {
        // the triangular mat-vecs rely on zeros in the unused triangles and in the slack
        DLSA_HIP_CHECK(hipMemsetAsync(a.Rinv, 0, (m * ld + LARS_SLACK) * 8, s));
        DLSA_HIP_CHECK(hipMemsetAsync(a.RinvT, 0, (m * ld + LARS_SLACK) * 8, s));
        // grid launches of this process run one at a time (launch .. completion): two of them could each hold part of the CUs the
        // other's queued workgroups need.  Other processes on the same GPU are covered by the barrier's timeout alone.
        std::unique_lock<std::mutex> grid_lock(g_lars_grid_mu, std::defer_lock);
        const int max_wgs = attempt == 0 ? LARS_MAX_WGS : 1;
        if (max_wgs > 1 && p >= 256) grid_lock.lock();
        const int rc = small_wg ? lars_t512::lars_run(a, p, intercept, s, max_wgs, &wgs_used) : lars_t1024::lars_run(a, p, intercept, s, max_wgs, &wgs_used);
        if (rc) return rc;
        DLSA_HIP_CHECK(hipMemcpyAsync(&steps, a.n_steps, sizeof(int), hipMemcpyDeviceToHost, s));
        DLSA_HIP_CHECK(hipStreamSynchronize(s));
        if (steps >= 0) break;
        // the grid kernel gave up at a barrier (its workgroups were not all resident within the timeout): the single-workgroup
        // kernel needs no co-residency -- same path, slower
        g_lars_grid_aborts.fetch_add(1);
        if (attempt == 1 || wgs_used == 1) {
            set_error("lars_lsa: the path kernel aborted (grid barrier timeout with %d workgroups)", wgs_used);
            return DLSA_ERR_HIP;
        }
    }
    if (n_steps_host) *n_steps_host = steps;
    return DLSA_OK;
}

// Test / diagnostics hook for the bounded grid barrier of lars_grid_kernel (not a reference entry: the reference's lars_lsa,
// dlsa/lsa.py:90-212, is host numpy and has no such state): sets the barrier timeout in seconds (<= 0 restores the 0.25 s default)
// and returns how many grid launches of this process have aborted and been rerun on the single-workgroup kernel.
int dlsa_lars_grid_barrier_timeout(double seconds) {
    dlsa::g_lars_barrier_timeout_ticks.store(seconds > 0.0 ? (long long)(seconds * 1e8) + 1 : 25000000ll);
    return dlsa::g_lars_grid_aborts.load();
}

}  // extern "C"
#endif  // DLSA_LARS_SECONDARY
