// LARS / lasso path for the least-squares approximation on the device
// (reference: lars_lsa and its helpers, dlsa/lsa.py:8-212; selection by AIC/BIC, dlsa/dlsa.py:87-105).
//
// The path is strictly sequential over steps, so it runs as ONE persistent workgroup
// (1024 threads = 16 waves on one CU) with the p x p matrices in HBM/L2 and all vectors in a
// global scratch area; there are no host round trips inside the path.  Differences in
// *method* (not in result) from the reference:
//   * the reference keeps the Cholesky factor R of Sigma[active,active] and does two
//     triangular solves per step (lsa.py:151); triangular solves are k dependent steps, so the
//     kernel keeps R^{-1} instead (appending a column is two mat-vecs, lsa.py:12-32) and gets
//     Gi1 = R^{-1} R^{-T} s from two mat-vecs;
//   * a = w Sigma[active, inactive] (lsa.py:157-160) and Sigma[:,active] w (lsa.py:177) are the
//     same vector u by symmetry and are computed once;
//   * RSS_k = (b-beta_k)' Sigma (b-beta_k) (lsa.py:190-192) equals (b-beta_k).Cvec_k because
//     Cvec_k = Sigma (b - beta_k) is carried along the path -- O(p) per step instead of O(p^2);
//   * a lasso drop (lsa.py:179-186) rebuilds R^{-1} for the remaining active set instead of
//     Givens-downdating R (lsa.py:35-80); the factor of the remaining ordered set is unique.
#include "common.h"
#include <math.h>

namespace dlsa {

#ifndef DLSA_LARS_THREADS
#define DLSA_LARS_THREADS 1024
#endif
constexpr int LARS_THREADS = DLSA_LARS_THREADS;
constexpr int LARS_WAVES = LARS_THREADS / 64;

struct LarsArgs {
    const double* Sigma0;   // p x p
    const double* b0;       // p
    int64_t lds0;
    int p, intercept, type, max_steps;
    double n, eps;
    // workspace
    double* S;        // m x m scaled Sigma
    double* Rinv;     // m x m upper triangular inverse factor (active order)
    double* vec;      // 12 vectors of length m (see kernel)
    int* ivec;        // 4 int vectors of length m
    // outputs
    double* beta_path; double* beta0; double* aic; double* bic;
    int* n_steps;     // device scalar
};

// LDS-free wave reductions (common.h): this kernel is one long chain of dependent reductions
__device__ __forceinline__ double wave_sum(double v) { return wave_allreduce_sum(v); }
__device__ __forceinline__ double wave_max(double v) { return wave_allreduce_max(v); }
__device__ __forceinline__ double wave_min(double v) { return wave_allreduce_min(v); }

// block-wide reductions; `red` is LDS scratch of LARS_WAVES+1 doubles.  Result to all threads.
__device__ double block_sum(double v, double* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double s = 0.0;
    for (int k = 0; k < LARS_WAVES; ++k) s += red[k];
    return s;
}
__device__ double block_max(double v, double* red) {
    v = wave_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double s = red[0];
    for (int k = 1; k < LARS_WAVES; ++k) s = fmax(s, red[k]);
    return s;
}
__device__ double block_min(double v, double* red) {
    v = wave_min(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double s = red[0];
    for (int k = 1; k < LARS_WAVES; ++k) s = fmin(s, red[k]);
    return s;
}

// Append variable `inew` to the factor (lsa.py:12-32 updateR, on R^{-1}).
// Returns (to all threads) 1 if the rank grew, 0 if the column is machine-singular.
__device__ int append_column(const double* __restrict__ S, double* __restrict__ Rinv, int m, int na,
                             int inew, const int* __restrict__ active, double eps,
                             double* __restrict__ xold, double* __restrict__ r, double* red) {
    const int tid = threadIdx.x;
    if (na == 0) {
        __syncthreads();
        const double d = S[(int64_t)inew * m + inew];
        if (tid == 0) Rinv[0] = 1.0 / sqrt(d);
        __syncthreads();
        return 1;
    }
    for (int i = tid; i < na; i += LARS_THREADS) xold[i] = S[(int64_t)inew * m + active[i]];
    __syncthreads();
    // r = R^{-T} xold :  r[i] = sum_{l<=i} Rinv[l][i] xold[l]   (coalesced over i)
    for (int i = tid; i < na; i += LARS_THREADS) {
        double s = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        int l = 0;
        for (; l + 3 <= i; l += 4) {
            s = fma(Rinv[(int64_t)l * m + i], xold[l], s);
            s1 = fma(Rinv[(int64_t)(l + 1) * m + i], xold[l + 1], s1);
            s2 = fma(Rinv[(int64_t)(l + 2) * m + i], xold[l + 2], s2);
            s3 = fma(Rinv[(int64_t)(l + 3) * m + i], xold[l + 3], s3);
        }
        for (; l <= i; ++l) s = fma(Rinv[(int64_t)l * m + i], xold[l], s);
        r[i] = (s + s1) + (s2 + s3);
    }
    __syncthreads();
    double part = 0.0;
    for (int i = tid; i < na; i += LARS_THREADS) part += r[i] * r[i];
    const double rr = block_sum(part, red);
    double rpp = S[(int64_t)inew * m + inew] - rr;
    if (rpp <= eps) return 0;            // rank did not grow: caller records an "ignore"
    rpp = sqrt(rpp);
    // new column of R^{-1}: [-R^{-1} r / rpp ; 1/rpp]   (one wave per row, coalesced over l)
    const int wave = tid >> 6, lane = tid & 63;
    for (int i = wave; i < na; i += LARS_WAVES) {
        double s = 0.0, s1 = 0.0;
        int l = i + lane;
        for (; l + 64 < na; l += 128) {
            s = fma(Rinv[(int64_t)i * m + l], r[l], s);
            s1 = fma(Rinv[(int64_t)i * m + l + 64], r[l + 64], s1);
        }
        if (l < na) s = fma(Rinv[(int64_t)i * m + l], r[l], s);
        s = wave_sum(s + s1);
        if (lane == 0) Rinv[(int64_t)i * m + na] = -s / rpp;
    }
    if (tid == 0) Rinv[(int64_t)na * m + na] = 1.0 / rpp;
    __syncthreads();
    return 1;
}

__global__ __launch_bounds__(LARS_THREADS) void lars_kernel(LarsArgs a) {
    __shared__ double red[LARS_WAVES + 1];
    __shared__ int sh_i[4];
    extern __shared__ __attribute__((aligned(16))) double dyn[];     // sh_w[m] | sh_part[LARS_THREADS] | sh_act[m]
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int p = a.p;
    const int off = a.intercept ? 1 : 0;
    const int m = p - off;
    const double eps = a.eps;
    double* sh_w = dyn;
    double* sh_part = dyn + m;
    int* sh_act = reinterpret_cast<int*>(dyn + m + LARS_THREADS);
    // thread groups for the mat-vec over the active set: JT threads along j, G groups along i
    int JT = 64;
    while (JT < m && JT < LARS_THREADS) JT *= 2;
    const int G = LARS_THREADS / JT;
    double* __restrict__ S = a.S;
    double* __restrict__ Rinv = a.Rinv;
    double* b = a.vec + 0 * (int64_t)m;       // sign(b0)
    double* absb = a.vec + 1 * (int64_t)m;    // |b0|
    double* Cvec = a.vec + 2 * (int64_t)m;
    double* beta = a.vec + 3 * (int64_t)m;    // current (scaled) coefficients
    double* u = a.vec + 4 * (int64_t)m;       // Sigma[:,active] w
    double* w = a.vec + 5 * (int64_t)m;       // by active position
    double* sgn = a.vec + 6 * (int64_t)m;     // by active position
    double* t1 = a.vec + 7 * (int64_t)m;
    double* t2 = a.vec + 8 * (int64_t)m;
    double* a12 = a.vec + 9 * (int64_t)m;
    int* active = a.ivec + 0 * (int64_t)m;    // active list (variable ids)
    int* state = a.ivec + 1 * (int64_t)m;     // 0 inactive, 1 active, 2 ignored
    int* dropf = a.ivec + 2 * (int64_t)m;     // by active position

    // ---- prologue: intercept Schur complement (lsa.py:98-104) and rescaling (lsa.py:108-109)
    double a11 = 1.0, beta0c = 0.0;
    if (a.intercept) {
        a11 = a.Sigma0[0];
        for (int j = tid; j < m; j += LARS_THREADS) a12[j] = a.Sigma0[(int64_t)(j + 1) * a.lds0];
    }
    for (int j = tid; j < m; j += LARS_THREADS) {
        const double v = a.b0[j + off];
        absb[j] = fabs(v);
        b[j] = (v > 0.0) ? 1.0 : ((v < 0.0) ? -1.0 : 0.0);
        beta[j] = 0.0;
        state[j] = 0;
    }
    __syncthreads();
    if (a.intercept) {
        double part = 0.0;
        for (int j = tid; j < m; j += LARS_THREADS) part += a12[j] * a.b0[j + 1];
        beta0c = block_sum(part, red) / a11;
    }
    for (int64_t e = tid; e < (int64_t)m * m; e += LARS_THREADS) {
        const int i = (int)(e / m), j = (int)(e % m);
        double v = a.Sigma0[(int64_t)(i + off) * a.lds0 + (j + off)];
        if (a.intercept) v -= a12[i] * a12[j] / a11;
        S[e] = absb[i] * v * absb[j];
    }
    __syncthreads();
    // Cvec = b' Sigma  (lsa.py:114); one wave per column block would be strided, S is symmetric
    // up to rounding, so use rows: Cvec[j] = sum_i b[i] S[i][j]
    for (int j = tid; j < m; j += LARS_THREADS) {
        double s = 0.0;
        for (int i = 0; i < m; ++i) s = fma(b[i], S[(int64_t)i * m + j], s);
        Cvec[j] = s;
    }
    __syncthreads();
    int max_steps = a.max_steps > 0 ? a.max_steps : 8 * m;
    // path row 0
    {
        double part = 0.0;
        for (int j = tid; j < m; j += LARS_THREADS) {
            a.beta_path[j] = 0.0;
            part += b[j] * Cvec[j];
        }
        const double rss = block_sum(part, red);
        if (tid == 0) {
            a.aic[0] = rss; a.bic[0] = rss;
            a.beta0[0] = a.intercept ? beta0c : 0.0;
        }
    }
    int na = 0, k = 0;
    bool had_drops = false;
    while (k < max_steps && na < m) {
        ++k;
        // ---- Cmax over the non-active variables (lsa.py:128-129)
        double part = 0.0;
        for (int j = tid; j < m; j += LARS_THREADS) if (state[j] != 1) part = fmax(part, fabs(Cvec[j]));
        const double Cmax = block_max(part, red);
        if (!had_drops) {
            // ---- new variables, in increasing index order (lsa.py:130-149)
            int start = 0;
            while (true) {
                __syncthreads();
                if (tid == 0) sh_i[0] = m;
                __syncthreads();
                int cand = m;
                for (int j = start + tid; j < m; j += LARS_THREADS)
                    if (state[j] == 0 && fabs(Cvec[j]) >= Cmax - eps) { cand = j; break; }
                if (cand < m) atomicMin(&sh_i[0], cand);
                __syncthreads();
                const int inew = sh_i[0];
                if (inew >= m) break;
                const int grew = append_column(S, Rinv, m, na, inew, active, eps, t1, t2, red);
                if (tid == 0) {
                    if (grew) {
                        active[na] = inew;
                        const double c = Cvec[inew];
                        sgn[na] = (c > 0.0) ? 1.0 : ((c < 0.0) ? -1.0 : 0.0);
                        state[inew] = 1;
                    } else {
                        state[inew] = 2;      // machine-singular: ignore (lsa.py:139-144)
                    }
                }
                if (grew) ++na;
                start = inew + 1;
                __syncthreads();
            }
        }
        if (na == 0) break;   // nothing could enter (degenerate input)
        // ---- equiangular direction: Gi1 = R^{-1} R^{-T} Sign (lsa.py:151-153)
        for (int i = tid; i < na; i += LARS_THREADS) {
            double s = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
            int l = 0;
            for (; l + 3 <= i; l += 4) {
                s = fma(Rinv[(int64_t)l * m + i], sgn[l], s);
                s1 = fma(Rinv[(int64_t)(l + 1) * m + i], sgn[l + 1], s1);
                s2 = fma(Rinv[(int64_t)(l + 2) * m + i], sgn[l + 2], s2);
                s3 = fma(Rinv[(int64_t)(l + 3) * m + i], sgn[l + 3], s3);
            }
            for (; l <= i; ++l) s = fma(Rinv[(int64_t)l * m + i], sgn[l], s);
            t1[i] = (s + s1) + (s2 + s3);
        }
        __syncthreads();
        for (int i = wave; i < na; i += LARS_WAVES) {
            double s = 0.0, s1 = 0.0;
            int l = i + lane;
            for (; l + 64 < na; l += 128) {
                s = fma(Rinv[(int64_t)i * m + l], t1[l], s);
                s1 = fma(Rinv[(int64_t)i * m + l + 64], t1[l + 64], s1);
            }
            if (l < na) s = fma(Rinv[(int64_t)i * m + l], t1[l], s);
            s = wave_sum(s + s1);
            if (lane == 0) t2[i] = s;
        }
        __syncthreads();
        part = 0.0;
        for (int i = tid; i < na; i += LARS_THREADS) part += t2[i] * sgn[i];
        const double A = 1.0 / sqrt(block_sum(part, red));
        for (int i = tid; i < na; i += LARS_THREADS) w[i] = A * t2[i];
        __syncthreads();
        // ---- u = Sigma[:,active] w  (rows of S, coalesced over j).  w and the active list are cached in
        // LDS (no dependent global loads); the i range is split over G thread groups of JT threads.
        for (int i = tid; i < na; i += LARS_THREADS) { sh_w[i] = w[i]; sh_act[i] = active[i]; }
        __syncthreads();
        {
            const int jx = tid % JT, g = tid / JT;
            for (int j0 = 0; j0 < m; j0 += JT) {
                const int j = j0 + jx;
                double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
                if (j < m) {
                    int i = g;
                    for (; i + 3 * G < na; i += 4 * G) {
                        s0 = fma(sh_w[i], S[(int64_t)sh_act[i] * m + j], s0);
                        s1 = fma(sh_w[i + G], S[(int64_t)sh_act[i + G] * m + j], s1);
                        s2 = fma(sh_w[i + 2 * G], S[(int64_t)sh_act[i + 2 * G] * m + j], s2);
                        s3 = fma(sh_w[i + 3 * G], S[(int64_t)sh_act[i + 3 * G] * m + j], s3);
                    }
                    for (; i < na; i += G) s0 = fma(sh_w[i], S[(int64_t)sh_act[i] * m + j], s0);
                }
                sh_part[g * JT + jx] = (s0 + s1) + (s2 + s3);
                __syncthreads();
                if (g == 0 && j < m) {
                    double t = sh_part[jx];
                    for (int q = 1; q < G; ++q) t += sh_part[q * JT + jx];
                    u[j] = t;
                }
                __syncthreads();
            }
        }
        // ---- step length (lsa.py:154-162)
        double gamhat = Cmax / A;
        if (na < m) {
            double gm = INFINITY;
            for (int j = tid; j < m; j += LARS_THREADS) {
                if (state[j] != 0) continue;
                const double c = Cvec[j], aj = u[j];
                const double g1 = (Cmax - c) / (A - aj);
                const double g2 = (Cmax + c) / (A + aj);
                if (g1 > eps) gm = fmin(gm, g1);
                if (g2 > eps) gm = fmin(gm, g2);
            }
            gm = block_min(gm, red);
            gamhat = fmin(gm, gamhat);
        }
        // ---- lasso modification (lsa.py:164-173)
        had_drops = false;
        if (a.type == 1) {
            double zm = INFINITY;
            for (int i = tid; i < na; i += LARS_THREADS) {
                const double z = -beta[active[i]] / w[i];
                t1[i] = z;
                if (z > eps) zm = fmin(zm, z);
            }
            zm = block_min(zm, red);
            if (zm < gamhat) {
                gamhat = zm;
                had_drops = true;
                for (int i = tid; i < na; i += LARS_THREADS) dropf[i] = (t1[i] == zm) ? 1 : 0;
            }
            __syncthreads();
        }
        // ---- move (lsa.py:175-177)
        for (int i = tid; i < na; i += LARS_THREADS) beta[active[i]] += gamhat * w[i];
        for (int j = tid; j < m; j += LARS_THREADS) Cvec[j] -= gamhat * u[j];
        __syncthreads();
        // ---- drops (lsa.py:179-186)
        if (had_drops) {
            for (int i = tid; i < na; i += LARS_THREADS)
                if (dropf[i]) { beta[active[i]] = 0.0; state[active[i]] = 0; }
            __syncthreads();
            if (tid == 0) {
                int q = 0;
                for (int i = 0; i < na; ++i)
                    if (!dropf[i]) { active[q] = active[i]; sgn[q] = sgn[i]; ++q; }
                sh_i[1] = q;
            }
            __syncthreads();
            const int keep = sh_i[1];
            // rebuild R^{-1} for the remaining ordered active set
            int nb = 0;
            for (int i = 0; i < keep; ++i) {
                append_column(S, Rinv, m, nb, active[i], active, 0.0, t1, t2, red);
                ++nb;
                __syncthreads();
            }
            na = keep;
        }
        // ---- record the path point: un-scaled beta (lsa.py:194-201), RSS, dof, AIC/BIC (:190-210)
        double prss = 0.0, pdof = 0.0, pb0 = 0.0;
        for (int j = tid; j < m; j += LARS_THREADS) {
            const double bj = beta[j];
            const double ub = absb[j] * bj;
            a.beta_path[(int64_t)k * m + j] = ub;
            prss += (b[j] - bj) * Cvec[j];
            if (fabs(ub) > eps) pdof += 1.0;
            if (a.intercept) pb0 += a12[j] * ub;
        }
        const double rss = block_sum(prss, red);
        const double dof = block_sum(pdof, red);
        double b0k = 0.0;
        if (a.intercept) b0k = beta0c - block_sum(pb0, red) / a11;
        if (tid == 0) {
            a.aic[k] = rss + 2.0 * dof;
            a.bic[k] = rss + log(a.n) * dof;
            a.beta0[k] = b0k;
        }
        __syncthreads();
    }
    if (tid == 0) *a.n_steps = k;
}

}  // namespace dlsa

extern "C" {

size_t dlsa_lars_workspace_bytes(int p) {
    if (p <= 0) return 0;
    const size_t m = (size_t)p;
    return dlsa::align_up(m * m * 8, 256) * 2 + dlsa::align_up(12 * m * 8, 256) + dlsa::align_up(4 * m * 4, 256) + 512;
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
    const size_t m = (size_t)p;
    LarsArgs a;
    a.Sigma0 = Sigma0; a.b0 = b0; a.lds0 = lds; a.p = p; a.intercept = intercept ? 1 : 0; a.type = type;
    a.max_steps = max_steps; a.n = n; a.eps = eps;
    a.S = (double*)ar.take(m * m * 8);
    a.Rinv = (double*)ar.take(m * m * 8);
    a.vec = (double*)ar.take(12 * m * 8);
    a.ivec = (int*)ar.take(4 * m * 4);
    a.n_steps = (int*)ar.take(256);
    a.beta_path = beta_path; a.beta0 = beta0; a.aic = aic; a.bic = bic;
    DLSA_HIP_CHECK(hipMemsetAsync(a.Rinv, 0, m * m * 8, s));
    const size_t shm = ((size_t)(p - (intercept ? 1 : 0)) * 12 + (size_t)LARS_THREADS * 8 + 64);
    hipLaunchKernelGGL(lars_kernel, dim3(1), dim3(LARS_THREADS), shm, s, a);
    DLSA_HIP_CHECK(hipGetLastError());
    int steps = 0;
    DLSA_HIP_CHECK(hipMemcpyAsync(&steps, a.n_steps, sizeof(int), hipMemcpyDeviceToHost, s));
    DLSA_HIP_CHECK(hipStreamSynchronize(s));
    if (n_steps_host) *n_steps_host = steps;
    return DLSA_OK;
}

}  // extern "C"
