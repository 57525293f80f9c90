// Minimum-norm least-squares solve of a symmetric p x p system on the device: the rank-deficient case of the WLS combine.
//
// Reference: dlsa/dlsa.py:48-49 -- beta_byOLS = np.linalg.lstsq(Sig_inv_sum, Sig_invMcoef_sum, rcond=None)[0].  For an SPD
// sum that is a plain solve (chol.hip).  When a coefficient's column is zero in every block (a dummy level that occurs in
// no partition, models.py:84-91) or two columns coincide, the sum is singular and lstsq returns the MINIMUM-NORM solution:
// singular values <= rcond * sigma_max are treated as zero, rcond = eps * p.  For a symmetric matrix the singular values
// are |eigenvalues| and the pseudo-inverse is V diag(1 / lambda_i : |lambda_i| > cut) V', so an eigendecomposition is all
// it takes.
//
// Kernel: parallel two-sided Jacobi.  A round-robin schedule pairs the m = p (+1 if odd) indices into m/2 disjoint pairs,
// m - 1 rounds per sweep.  Within a round the rotations commute on disjoint 2 x 2 blocks: thread (a, b) owns the block
// (pair a) x (pair b) of A and of V, computes the two rotations from the diagonal blocks of the INPUT copy and writes the
// rotated block to the OUTPUT copy (ping-pong, so no thread reads what another writes).  One launch per round; the matrices
// (2 MB at p = 500) stay in L2.  A sweep ends with a reduction of the off-diagonal mass; 6-10 sweeps reach 1e-30 relative.
// Latency-bound and rarely taken (only when the Cholesky of the combine fails or is numerically rank-deficient).
#include "common.h"
#include <algorithm>
#include <math.h>
#include <vector>

namespace dlsa {

// round-robin 1-factorisation of K_m (m even): round r, slot k -> (i, j), i < j
__device__ __forceinline__ void rr_pair(int m, int r, int k, int& i, int& j) {
    const int q = m - 1;
    int a, b;
    if (k == 0) { a = q; b = r % q; }
    else { a = (r + k) % q; b = (r - k + q) % q; }
    i = a < b ? a : b;
    j = a < b ? b : a;
}

// rotation that annihilates a_ij:  J = [[c, s], [-s, c]] on (i, j);  columns: x_i' = c x_i - s x_j, x_j' = s x_i + c x_j
__device__ __forceinline__ void jacobi_cs(double aii, double ajj, double aij, double& c, double& s) {
    if (aij == 0.0 || !(fabs(aij) > 1e-300)) { c = 1.0; s = 0.0; return; }
    const double tau = (ajj - aii) / (2.0 * aij);
    const double t = (tau >= 0.0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
    c = 1.0 / sqrt(1.0 + t * t);
    s = t * c;
}

__global__ void jacobi_round_kernel(const double* __restrict__ Ain, const double* __restrict__ Vin,
                                    double* __restrict__ Aout, double* __restrict__ Vout, int m, int r) {
    const int half = m >> 1;
    const int a = blockIdx.y * blockDim.y + threadIdx.y;
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= half || b >= half) return;
    int ia, ja, ib, jb;
    rr_pair(m, r, a, ia, ja);
    rr_pair(m, r, b, ib, jb);
    double ca, sa, cb, sb;
    jacobi_cs(Ain[(int64_t)ia * m + ia], Ain[(int64_t)ja * m + ja], Ain[(int64_t)ia * m + ja], ca, sa);
    jacobi_cs(Ain[(int64_t)ib * m + ib], Ain[(int64_t)jb * m + jb], Ain[(int64_t)ib * m + jb], cb, sb);
    // A' = Ja' A Jb on the 2 x 2 block
    const double x00 = Ain[(int64_t)ia * m + ib], x01 = Ain[(int64_t)ia * m + jb];
    const double x10 = Ain[(int64_t)ja * m + ib], x11 = Ain[(int64_t)ja * m + jb];
    const double y00 = ca * x00 - sa * x10, y01 = ca * x01 - sa * x11;      // rows rotated by pair a
    const double y10 = sa * x00 + ca * x10, y11 = sa * x01 + ca * x11;
    double z00 = cb * y00 - sb * y01, z01 = sb * y00 + cb * y01;            // columns rotated by pair b
    double z10 = cb * y10 - sb * y11, z11 = sb * y10 + cb * y11;
    if (a == b) { z01 = 0.0; z10 = 0.0; }                                    // the annihilated entry, exactly
    Aout[(int64_t)ia * m + ib] = z00; Aout[(int64_t)ia * m + jb] = z01;
    Aout[(int64_t)ja * m + ib] = z10; Aout[(int64_t)ja * m + jb] = z11;
    // V' = V Jb: rows ia, ja of V, columns of pair b
    const double v00 = Vin[(int64_t)ia * m + ib], v01 = Vin[(int64_t)ia * m + jb];
    const double v10 = Vin[(int64_t)ja * m + ib], v11 = Vin[(int64_t)ja * m + jb];
    Vout[(int64_t)ia * m + ib] = cb * v00 - sb * v01; Vout[(int64_t)ia * m + jb] = sb * v00 + cb * v01;
    Vout[(int64_t)ja * m + ib] = cb * v10 - sb * v11; Vout[(int64_t)ja * m + jb] = sb * v10 + cb * v11;
}

// A (m x m, zero padded) <- symmetrised S;  V <- I;  acc[0..2] <- 0
__global__ void jacobi_init_kernel(const double* __restrict__ S, int64_t lds, int p, int m, double* __restrict__ A,
                                   double* __restrict__ V, double* __restrict__ acc) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e == 0) { acc[0] = 0.0; acc[1] = 0.0; acc[2] = 0.0; }
    if (e >= (int64_t)m * m) return;
    const int i = (int)(e / m), j = (int)(e % m);
    A[e] = (i < p && j < p) ? 0.5 * (S[(int64_t)i * lds + j] + S[(int64_t)j * lds + i]) : 0.0;
    V[e] = i == j ? 1.0 : 0.0;
}

// acc[0] += sum of squares off the diagonal, acc[1] += sum of squares on it, acc[2] = NaN flag
__global__ void jacobi_offnorm_kernel(const double* __restrict__ A, int m, double* __restrict__ acc) {
    const int64_t e0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    double off = 0.0, dia = 0.0, bad = 0.0;
    for (int64_t e = e0; e < (int64_t)m * m; e += (int64_t)gridDim.x * blockDim.x) {
        const double v = A[e];
        if (!isfinite(v)) bad = 1.0;
        if (e / m == e % m) dia += v * v; else off += v * v;
    }
    off = wave_allreduce_sum(off); dia = wave_allreduce_sum(dia); bad = wave_allreduce_max(bad);
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(acc + 0, off);
        atomicAdd(acc + 1, dia);
        if (bad != 0.0) acc[2] = 1.0;
    }
}

// t_i = (v_i . rhs) / lambda_i for |lambda_i| > cut, else 0;  lam_out[i] = lambda_i;  one thread per eigenpair (coalesced in i)
__global__ void pinv_project_kernel(const double* __restrict__ A, const double* __restrict__ V, int m, int p,
                                    const double* __restrict__ rhs, double rcond, double* __restrict__ t,
                                    double* __restrict__ lam_out, int* __restrict__ rank) {
    __shared__ double lmax_s;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    // largest |eigenvalue| (every block recomputes it: m <= 2048 values)
    double lm = 0.0;
    for (int k = threadIdx.x; k < m; k += blockDim.x) lm = fmax(lm, fabs(A[(int64_t)k * m + k]));
    lm = wave_allreduce_max(lm);
    if (threadIdx.x == 0) lmax_s = 0.0;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) atomicMax((unsigned long long*)&lmax_s, (unsigned long long)__double_as_longlong(lm));   // non-negative doubles order as integers
    __syncthreads();
    const double cut = rcond * lmax_s;
    if (i >= m) return;
    const double lam = A[(int64_t)i * m + i];
    double dot = 0.0;
    for (int r = 0; r < p; ++r) dot = fma(V[(int64_t)r * m + i], rhs[r], dot);
    const bool keep = fabs(lam) > cut;
    t[i] = keep ? dot / lam : 0.0;
    lam_out[i] = lam;
    if (keep) atomicAdd(rank, 1);
}

// theta_r = sum_i V[r][i] t_i   (one wave per row)
__global__ void pinv_expand_kernel(const double* __restrict__ V, int m, int p, const double* __restrict__ t,
                                   double* __restrict__ theta) {
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= p) return;
    double s = 0.0;
    for (int k = lane; k < m; k += 64) s = fma(V[(int64_t)row * m + k], t[k], s);
    s = wave_allreduce_sum(s);
    if (lane == 0) theta[row] = s;
}

struct PinvLayout { size_t A0, A1, V0, V1, t, lam, acc, rank, total; int m; };

static PinvLayout pinv_layout(int p) {
    PinvLayout l;
    l.m = p + (p & 1);
    const size_t mm = align_up((size_t)l.m * l.m * sizeof(double), 256), mv = align_up((size_t)l.m * sizeof(double), 256);
    size_t o = 0;
    l.A0 = o; o += mm; l.A1 = o; o += mm; l.V0 = o; o += mm; l.V1 = o; o += mm;
    l.t = o; o += mv; l.lam = o; o += mv; l.acc = o; o += 256; l.rank = o; o += 256;
    l.total = o;
    return l;
}

size_t sym_pinv_workspace_bytes_impl(int p) { return pinv_layout(p).total; }

// theta = pinv(S) v with the lstsq(rcond) cut; eig_host (nullable, p values) receives the eigenvalues, rank_host the rank
int sym_pinv_solve_impl(const double* S, int64_t lds, const double* v, int p, double rcond, double* theta,
                        int* rank_host, double* eig_host, int* sweeps_host, void* ws, size_t ws_bytes, hipStream_t s) {
    const PinvLayout l = pinv_layout(p);
    if (!ws || ws_bytes < l.total || ((uintptr_t)ws & 255)) {
        set_error("sym_pinv_solve: workspace %zu bytes needed (256-aligned), got %zu", l.total, ws_bytes);
        return DLSA_ERR_WORKSPACE;
    }
    char* base = (char*)ws;
    double* A[2] = {(double*)(base + l.A0), (double*)(base + l.A1)};
    double* V[2] = {(double*)(base + l.V0), (double*)(base + l.V1)};
    double* t = (double*)(base + l.t);
    double* lam = (double*)(base + l.lam);
    double* acc = (double*)(base + l.acc);
    int* rank = (int*)(base + l.rank);
    const int m = l.m, half = m / 2;
    const int64_t mm = (int64_t)m * m;
    hipLaunchKernelGGL(jacobi_init_kernel, dim3((unsigned)((mm + 255) / 256)), dim3(256), 0, s, S, lds, p, m, A[0], V[0], acc);
    int cur = 0, sweeps = 0;
    bool settled = m <= 1;
    const dim3 blk(32, 8), grd((half + 31) / 32, (half + 7) / 8);
    const int red_blocks = (int)std::min<int64_t>(256, (mm + 255) / 256);
    for (int sweep = 0; sweep < 40 && m > 1; ++sweep) {
        for (int r = 0; r < m - 1; ++r) {
            hipLaunchKernelGGL(jacobi_round_kernel, grd, blk, 0, s, (const double*)A[cur], (const double*)V[cur], A[cur ^ 1], V[cur ^ 1], m, r);
            cur ^= 1;
        }
        ++sweeps;
        DLSA_HIP_CHECK(hipMemsetAsync(acc, 0, 3 * sizeof(double), s));
        hipLaunchKernelGGL(jacobi_offnorm_kernel, dim3(red_blocks), dim3(256), 0, s, (const double*)A[cur], m, acc);
        double h[3];
        DLSA_HIP_CHECK(hipMemcpyAsync(h, acc, sizeof(h), hipMemcpyDeviceToHost, s));
        DLSA_HIP_CHECK(hipStreamSynchronize(s));
        if (h[2] != 0.0 || !isfinite(h[0]) || !isfinite(h[1])) { set_error("sym_pinv_solve: NaN/Inf in the system"); return DLSA_ERR_NAN; }
        if (h[0] <= 1e-30 * (h[0] + h[1])) { settled = true; break; }      // off-diagonal mass below 1e-15 of the Frobenius norm
    }
    if (!settled) {          // 40 sweeps without reaching the target: say so instead of returning a half-diagonalised spectrum
        set_error("sym_pinv_solve: the Jacobi iteration did not reach its off-diagonal target in %d sweeps (p = %d)", sweeps, p);
        return DLSA_ERR_NOT_CONVERGED;
    }
    DLSA_HIP_CHECK(hipMemsetAsync(rank, 0, sizeof(int), s));
    hipLaunchKernelGGL(pinv_project_kernel, dim3((m + 255) / 256), dim3(256), 0, s, (const double*)A[cur], (const double*)V[cur], m, p, v,
                       rcond, t, lam, rank);
    hipLaunchKernelGGL(pinv_expand_kernel, dim3((p + 3) / 4), dim3(256), 0, s, (const double*)V[cur], m, p, (const double*)t, theta);
    DLSA_HIP_CHECK(hipGetLastError());
    int rk = 0;
    DLSA_HIP_CHECK(hipMemcpyAsync(&rk, rank, sizeof(int), hipMemcpyDeviceToHost, s));
    if (eig_host) DLSA_HIP_CHECK(hipMemcpyAsync(eig_host, lam, (size_t)p * sizeof(double), hipMemcpyDeviceToHost, s));
    DLSA_HIP_CHECK(hipStreamSynchronize(s));
    if (rank_host) *rank_host = rk;
    if (sweeps_host) *sweeps_host = sweeps;
    return DLSA_OK;
}

int launch_chol_solve(const double* A, int64_t lda, int64_t strideA, const double* rhs, int64_t stride_rhs,
                      const double* ref, int64_t stride_ref, int p, int nsys, double* Lws, double* xout,
                      int64_t stride_x, double* stats, int64_t stride_stats, hipStream_t s, int reuse_factor);   // chol.hip

}  // namespace dlsa

extern "C" {

size_t dlsa_sym_pinv_workspace_bytes(int p) {
    if (p <= 0 || p > 2048) return 0;
    return dlsa::sym_pinv_workspace_bytes_impl(p);
}

int dlsa_sym_pinv_solve_f64(const double* S, int64_t lds, const double* v, int p, double rcond, double* theta,
                            int* rank_host, double* eig_host, void* ws, size_t ws_bytes, void* stream) {
    using namespace dlsa;
    DLSA_REQUIRE(S && v && theta, "sym_pinv_solve: null argument");
    DLSA_REQUIRE(p > 0 && p <= 2048 && lds >= p, "sym_pinv_solve: bad shape p=%d lds=%lld", p, (long long)lds);
    if (!(rcond >= 0.0)) rcond = 2.220446049250313e-16 * p;          // lstsq(rcond=None): eps * max(M, N)
    return sym_pinv_solve_impl(S, lds, v, p, rcond, theta, rank_host, eig_host, nullptr, ws, ws_bytes, (hipStream_t)stream);
}

size_t dlsa_wls_solve_workspace_bytes(int p) {
    if (p <= 0 || p > 2048) return 0;
    return std::max(dlsa::sym_pinv_workspace_bytes_impl(p), dlsa_solve_workspace_bytes(p) + (size_t)p * sizeof(double) + 512);
}

int dlsa_wls_solve_f64(const double* S, int64_t lds, const double* v, int p, double* theta, int* rank_host,
                       void* ws, size_t ws_bytes, void* stream) {
    using namespace dlsa;
    DLSA_REQUIRE(S && v && theta, "wls_solve: null argument");
    DLSA_REQUIRE(p > 0 && p <= 2048 && lds >= p, "wls_solve: bad shape p=%d lds=%lld", p, (long long)lds);
    const size_t need = dlsa_wls_solve_workspace_bytes(p);
    if (!ws || ws_bytes < need || ((uintptr_t)ws & 255)) {
        set_error("wls_solve: workspace %zu bytes needed (256-aligned), got %zu", need, ws_bytes);
        return DLSA_ERR_WORKSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    // 1. the SPD case: blocked Cholesky.  Accept it only when the factor is comfortably full rank: a singular sum can slip
    //    through with a roundoff-sized positive pivot, so compare the pivots L_ii^2 with the diagonal they came from.
    Arena ar(ws, ws_bytes);
    double* L = (double*)ar.take((size_t)p * p * sizeof(double));
    double* stats = (double*)ar.take(4 * sizeof(double));
    int rc = launch_chol_solve(S, lds, 0, v, 0, nullptr, 0, p, 1, L, theta, 0, stats, 0, s, 0);
    if (rc) return rc;
    std::vector<double> dl((size_t)p), ds((size_t)p);
    double h[3];
    DLSA_HIP_CHECK(hipMemcpyAsync(h, stats, sizeof(h), hipMemcpyDeviceToHost, s));
    DLSA_HIP_CHECK(hipMemcpy2DAsync(dl.data(), sizeof(double), L, (size_t)(p + 1) * sizeof(double), sizeof(double), (size_t)p, hipMemcpyDeviceToHost, s));
    DLSA_HIP_CHECK(hipMemcpy2DAsync(ds.data(), sizeof(double), S, (size_t)(lds + 1) * sizeof(double), sizeof(double), (size_t)p, hipMemcpyDeviceToHost, s));
    DLSA_HIP_CHECK(hipStreamSynchronize(s));
    if (h[2] == 2.0) { set_error("wls_solve: NaN/Inf in the system"); return DLSA_ERR_NAN; }
    bool full = (h[2] == 0.0);
    const double thresh = 8.0 * 2.220446049250313e-16 * p;
    double lmin = INFINITY, smax = 0.0;
    for (int i = 0; i < p && full; ++i) {
        if (!(dl[i] * dl[i] > thresh * fabs(ds[i])) || !isfinite(dl[i])) full = false;
        lmin = fmin(lmin, dl[i] * dl[i]);
        smax = fmax(smax, fabs(ds[i]));
    }
    // lstsq(rcond=None) cuts singular values relative to the LARGEST one (eps p sigma_max), not relative to a pivot's own diagonal
    // entry: a badly scaled SPD sum -- diag(1, 1e-20), unstandardised columns with cond > 1 / (eps p) -- passes the test above
    // and would return S^-1 v where numpy returns the truncated minimum-norm solution.  min L_ii^2 <= lambda_min-ish and
    // max S_ii <= lambda_max <= p max S_ii bracket the condition number cheaply; below the cut the spectral path decides.
    if (full && !(lmin > 2.220446049250313e-16 * p * smax)) full = false;
    if (full) { if (rank_host) *rank_host = p; return DLSA_OK; }
    // 2. rank-deficient (or indefinite): minimum-norm least-squares solution, lstsq(rcond=None) semantics
    return sym_pinv_solve_impl(S, lds, v, p, 2.220446049250313e-16 * p, theta, rank_host, nullptr, nullptr, ws, ws_bytes, s);
}

}  // extern "C"
