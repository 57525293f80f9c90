// The map step for MANY SMALL partitions (dlsa/models.py:110-131 per partition; config 1: 20 partitions of 5 000 x 50 rows,
// projects/logistic_dlsa.py:89-92).
//
// irls.hip drives one partition at a time from the host: a Newton iteration is a dozen launches and one host round trip,
// ~0.1 ms, whatever the partition's size -- 0.6 ms per 5 000-row partition whose arithmetic takes microseconds.  Partitions
// are independent, so here ONE launch fits them all: a workgroup per partition runs the whole Newton iteration to
// convergence on the device -- no host synchronisation, no inter-workgroup communication.
//   per iteration:  rows -> eta, mu, w, residual, loglik (the 8 waves deal the 4-row k-steps, 4 at a time with all their loads
//                   in flight first -- a 2 MB partition's pass is latency-bound: 0.78 -> 0.1 ms; the 16 lanes of a row
//                   reduce eta with DPP);  g += residual x;  H += x' w x on v_mfma_f64_16x16x4_f64 (upper-triangle tiles,
//                   fragments loaded straight from global memory: the partition lives in L2 / Infinity Cache after the
//                   first pass);  the waves' partials meet in LDS in a fixed order (deterministic);
//                   wave 0: Cholesky of H in LDS (lane = row), two triangular solves, step, stopping rule.
// Stopping rule, step halving and the returned (coef, H at coef) are the oracle's / irls.hip's: |delta|_inf <= tol max(1, |beta|_inf).
// CLUSTERS (round 4): with fewer partitions than CUs, C workgroups (on C CUs) share a partition: they deal its row batches, write
// their partial H, g, loglik to global scratch and meet at a per-partition barrier (a counter with agent-scope release / acquire, as
// the LARS grid kernel's; bounded: a timeout aborts the launch and the host reruns it with C = 1); workgroup 0 of the cluster
// sums the partials in a fixed order, runs the Cholesky / step / stopping rule and publishes beta and the loop state before a
// second barrier.  Config 1 (20 x 5 000 x 50) uses 160 CUs instead of 20.
// Width: p + intercept <= 64 columns (4 tiles per side, 10 accumulator tiles per wave); the implicit intercept is the LAST
// column inside the kernel and the FIRST one in the outputs (models.py:136-142).
#include "common.h"
#include "options.h"
#include <algorithm>
#include <atomic>
#include <mutex>
#include <math.h>
#include <vector>

namespace dlsa {

constexpr int SM_MAXP = 64;
constexpr int SM_LD = SM_MAXP + 1;            // LDS row pitch of the p x p matrices
constexpr int SM_THREADS = 512;               // 8 waves: two per SIMD, 256 registers each
constexpr int SM_WAVES = SM_THREADS / 64;
constexpr int SM_U = 4;                       // k-steps per batch: their loads are in flight together, their 16 rows' transcendentals run once

struct SmallArgs {
    const double* X;
    const double* y;
    const int64_t* first;     // [K] device: first row of partition k
    const int64_t* rows;      // [K] device: rows of partition k
    int64_t ldx, step;
    int p, icpt, pe, max_iter;
    double tol;
    double* coef;             // [K][pe]
    double* sig;              // [K][pe][pe]
    double* smc;              // [K][pe]
    int* n_iter;              // [K] device
    int* status;              // [K] device
    double* loglik;           // [K] device
    // clusters (C > 1)
    int C;                    // workgroups per partition
    double* scratch;          // [K][C][SM_SLOT]: partial H (rows < pe, pitch SM_LD), g, loglik
    double* bcast;            // [K][SM_BSLOT]: beta, loop state, status, loglik from the cluster's workgroup 0
    unsigned* bar;            // [K][16]: [0] arrivals; bar_abort: one word for the launch
    unsigned* bar_abort;
    long long bar_timeout;    // ticks of the 100 MHz wall clock
};
constexpr int SM_SLOT = SM_MAXP * SM_LD + SM_MAXP + 8;
constexpr int SM_BSLOT = SM_MAXP + 8;

// Barrier of the C workgroups of one partition (lars.hip's grid_barrier on a per-partition counter): relaxed agent-scope polling
// with ONE acquire after the match; a wait longer than the timeout sets the launch's abort word and every workgroup leaves.
__device__ __forceinline__ bool cluster_barrier(unsigned* bar, unsigned* abort_word, unsigned nwg, unsigned& phase, long long timeout) {
    __shared__ int cb_ok;
    __syncthreads();                 // this workgroup's global stores have been issued by every wave
    if (threadIdx.x == 0) {
        ++phase;
        const unsigned target = phase * nwg;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        bool ok = true;
        const long long t0 = wall_clock64();
        unsigned spins = 0;
        while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if ((spins++ & 255u) == 0u) {
                if (__hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { ok = false; break; }
                if (wall_clock64() - t0 > timeout) {
                    __hip_atomic_store(abort_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ok = false;
                    break;
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        cb_ok = ok ? 1 : 0;
    }
    __syncthreads();
    return cb_ok != 0;
}

__device__ __forceinline__ double row16_sum(double v) {      // sum over the 16 lanes of a row group
    v += dpp_xor_f64<1>(v);
    v += dpp_xor_f64<2>(v);
    v += dpp_xor_f64<4>(v);
    v += dpp_xor_f64<8>(v);
    return v;
}

template <int NT>
__global__ __launch_bounds__(SM_THREADS) void irls_small_kernel(SmallArgs a) {
    constexpr int NTRI = NT * (NT + 1) / 2;
    typedef double acc_t __attribute__((ext_vector_type(4)));
    __shared__ double Hs[SM_MAXP * SM_LD];      // the Hessian (full, symmetric)
    __shared__ double Ls[SM_MAXP * SM_LD];      // its Cholesky factor (lower)
    __shared__ double gs[SM_MAXP], beta[SM_MAXP], prev[SM_MAXP], stepv[SM_MAXP];
    __shared__ double sc[4];                    // [0] loglik  [1] loop state  [2] status
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int C = a.C;
    const int k = blockIdx.x / C, cpart = blockIdx.x - k * C;
    const bool leader = cpart == 0;                 // workgroup 0 of the partition's cluster: sums, factorises, decides, writes the results
    unsigned bphase = 0;
    unsigned* const cbar = a.bar ? a.bar + 16 * k : nullptr;
    const int pe = a.pe, p = a.p;
    const int64_t nk = a.rows[k], r0 = a.first[k], pitch = a.ldx * a.step;
    const double* __restrict__ Xk = a.X + r0 * a.ldx;
    const double* __restrict__ yk = a.y + r0;
    const int kg = lane >> 4, cl = lane & 15;

    if (nk == 0) {      // empty partition: the reference's zero block (models.py:84-91)
        if (!leader) return;
        for (int e = tid; e < pe * pe; e += SM_THREADS) a.sig[(int64_t)k * pe * pe + e] = 0.0;
        for (int e = tid; e < pe; e += SM_THREADS) { a.coef[(int64_t)k * pe + e] = 0.0; a.smc[(int64_t)k * pe + e] = 0.0; }
        if (tid == 0) { a.n_iter[k] = 0; a.status[k] = DLSA_PART_EMPTY; a.loglik[k] = 0.0; }
        return;
    }
    for (int e = tid; e < SM_MAXP; e += SM_THREADS) { beta[e] = 0.0; prev[e] = 0.0; stepv[e] = 0.0; }
    __syncthreads();

    double ll_prev = -INFINITY;
    bool have_prev = false, last_pass = false;
    int halvings = 0, iters = 0, evals = 0, status = DLSA_PART_NOT_CONVERGED;
    double ll = 0.0;
    for (;;) {
        ++evals;
        // ---- one pass over the rows at the current beta
        double bl[NT], g[NT];
        acc_t acc[NTRI];
#pragma unroll
        for (int t = 0; t < NT; ++t) { bl[t] = beta[16 * t + cl]; g[t] = 0.0; }
#pragma unroll
        for (int t = 0; t < NTRI; ++t) acc[t] = acc_t{0, 0, 0, 0};
        double llw = 0.0;
        const int64_t nks = (nk + 3) / 4;
        for (int64_t ks0 = (int64_t)(cpart * SM_WAVES + wave) * SM_U; ks0 < nks; ks0 += (int64_t)C * SM_WAVES * SM_U) {
            // all loads of SM_U k-steps first (a k-step = 4 rows x 16 NT columns in the MFMA fragment layout), then the arithmetic
            double xs[SM_U][NT], ys[SM_U];
            bool vs[SM_U];
#pragma unroll
            for (int u = 0; u < SM_U; ++u) {
                const int64_t r = (ks0 + u) * 4 + kg;
                vs[u] = r < nk;
                const double* rowp = Xk + (vs[u] ? r : 0) * pitch;
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const int col = 16 * t + cl;
                    xs[u][t] = rowp[col < p ? col : 0];                 // clamped, masked below: the load stays unconditional
                }
                ys[u] = yk[(vs[u] ? r : 0) * a.step];
            }
            // eta of the batch's 4 SM_U rows, then the transcendentals ONCE per row: lane (kg, cl = u) evaluates row (u, kg) -- every
            // lane of a row group would otherwise run the same exp / log1p sequence for the same four rows
            double eta_u[SM_U];
#pragma unroll
            for (int u = 0; u < SM_U; ++u) {
                double part = 0.0;
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const int col = 16 * t + cl;
                    const double v = !vs[u] ? 0.0 : (col < p ? xs[u][t] : ((a.icpt && col == p) ? 1.0 : 0.0));
                    xs[u][t] = v;
                    part = fma(v, bl[t], part);
                }
                eta_u[u] = row16_sum(part);
            }
            double my_eta = eta_u[0], my_y = ys[0];
            bool my_valid = vs[0];
#pragma unroll
            for (int u = 1; u < SM_U; ++u)
                if ((cl & (SM_U - 1)) == u) { my_eta = eta_u[u]; my_y = ys[u]; my_valid = vs[u]; }
            const double e = exp(-fabs(my_eta));
            const double inv = 1.0 / (1.0 + e);
            const double my_mu = my_eta >= 0.0 ? inv : e * inv;
            const double my_w = my_valid ? e * inv * inv : 0.0;
            const double my_res = my_valid ? my_y - my_mu : 0.0;
            if (my_valid && cl < SM_U) llw += my_y * my_eta - (fmax(my_eta, 0.0) + log1p(e));     // y eta - softplus(eta), once per row
#pragma unroll
            for (int u = 0; u < SM_U; ++u) {
                const int src = (lane & 48) | u;                    // the lane of this row group that evaluated row u
                const double wv = __shfl(my_w, src, 64), resid = __shfl(my_res, src, 64);
                double bw[NT];
#pragma unroll
                for (int t = 0; t < NT; ++t) { g[t] = fma(resid, xs[u][t], g[t]); bw[t] = xs[u][t] * wv; }
#pragma unroll
                for (int tj = 0; tj < NT; ++tj)
#pragma unroll
                    for (int ti = 0; ti <= tj; ++ti)
                        acc[tj * (tj + 1) / 2 + ti] = __builtin_amdgcn_mfma_f64_16x16x4f64(xs[u][ti], bw[tj], acc[tj * (tj + 1) / 2 + ti], 0, 0, 0);
            }
        }
        // ---- the waves meet in LDS, one after the other (fixed order)
#pragma unroll
        for (int t = 0; t < NT; ++t) { g[t] += __shfl_xor(g[t], 16, 64); g[t] += __shfl_xor(g[t], 32, 64); }
        llw = wave_allreduce_sum(llw);
        for (int wv2 = 0; wv2 < SM_WAVES; ++wv2) {
            if (wave == wv2) {
#pragma unroll
                for (int tj = 0; tj < NT; ++tj)
#pragma unroll
                    for (int ti = 0; ti <= tj; ++ti)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {      // C/D register q of lane l = C[4q + (l >> 4)][l & 15]
                            double* d = Hs + (16 * ti + 4 * q + kg) * SM_LD + 16 * tj + cl;
                            const double v = acc[tj * (tj + 1) / 2 + ti][q];
                            if (wv2 == 0) *d = v; else *d += v;
                        }
                if (kg == 0) {
#pragma unroll
                    for (int t = 0; t < NT; ++t) { if (wv2 == 0) gs[16 * t + cl] = g[t]; else gs[16 * t + cl] += g[t]; }
                }
                if (lane == 0) { if (wv2 == 0) sc[0] = llw; else sc[0] += llw; }
            }
            __syncthreads();
        }
        if (C > 1) {
            // the cluster meets: partials to global scratch, barrier, workgroup 0 adds the others' in the order 1 .. C - 1
            double* __restrict__ slot = a.scratch + ((int64_t)k * C + cpart) * SM_SLOT;
            if (!leader) {
                for (int e = tid; e < pe * SM_LD; e += SM_THREADS) slot[e] = Hs[e];
                if (tid < pe) slot[SM_MAXP * SM_LD + tid] = gs[tid];
                if (tid == 0) slot[SM_MAXP * SM_LD + SM_MAXP] = sc[0];
            }
            if (!cluster_barrier(cbar, a.bar_abort, (unsigned)C, bphase, a.bar_timeout)) return;
            if (leader) {
                // element by element: the C - 1 loads of an element go out together (one L2 round trip per element, not one per
                // partial), the additions keep the order 1 .. C - 1
                const double* __restrict__ src0 = a.scratch + (int64_t)k * C * SM_SLOT;
                auto gather_add = [&](int e, double acc) {
                    double v[15];
#pragma unroll
                    for (int c = 1; c < 16; ++c) v[c - 1] = c < C ? src0[(int64_t)c * SM_SLOT + e] : 0.0;
#pragma unroll
                    for (int c = 1; c < 16; ++c) if (c < C) acc += v[c - 1];
                    return acc;
                };
                if (C >= 4) {
                    for (int e = tid; e < pe * SM_LD; e += SM_THREADS) Hs[e] = gather_add(e, Hs[e]);
                    if (tid < pe) gs[tid] = gather_add(SM_MAXP * SM_LD + tid, gs[tid]);
                    if (tid == 0) sc[0] = gather_add(SM_MAXP * SM_LD + SM_MAXP, sc[0]);
                } else {
                    for (int c = 1; c < C; ++c) {
                        const double* __restrict__ src = src0 + (int64_t)c * SM_SLOT;
                        for (int e = tid; e < pe * SM_LD; e += SM_THREADS) Hs[e] += src[e];
                        if (tid < pe) gs[tid] += src[SM_MAXP * SM_LD + tid];
                        if (tid == 0) sc[0] += src[SM_MAXP * SM_LD + SM_MAXP];
                    }
                }
                __syncthreads();
            }
        }
        if (leader) {
            for (int e = tid; e < pe * pe; e += SM_THREADS) {          // mirror: the lower triangle is the transpose of the upper
                const int i = e / pe, j = e - i * pe;
                if (i > j) Hs[i * SM_LD + j] = Hs[j * SM_LD + i];
            }
            __syncthreads();
            ll = sc[0];
        }
        if (last_pass) break;                               // max_iter reached: H, loglik are those of the last iterate
        // ---- wave 0: safeguard, Cholesky, solve, step, stopping rule
        if (leader && wave == 0) {
            const int j = lane;
            int state = 0;                                  // 0 continue, 1 converged, 2 failed (status in sc[2]), 3 step halved
            if (!isfinite(ll)) { state = 2; if (lane == 0) sc[2] = DLSA_PART_NAN; }
            else if (have_prev && ll < ll_prev - 1e-12 * fabs(ll_prev) && halvings < 30) {
                // the previous step overshot: halve it and evaluate again (oracle irls_logistic / irls.hip)
                if (j < pe) { stepv[j] *= 0.5; beta[j] = prev[j] + stepv[j]; }
                state = 3;
            } else {
                // One wave, lane = row, data passed between lanes through LDS: every store that another lane reads next is followed by
                // a wavefront-scope release / acquire + wave barrier -- lock-step execution alone is not part of the memory model, and
                // nothing else stops the compiler from hoisting the next column's loads above these stores.
                auto wave_sync = [&]() {
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                };
                // Cholesky by columns, left-looking: lane j forms s_j = H[j][c] - sum_{k<c} L[j][k] L[c][k] with independent,
                // pipelined LDS reads (its own row, pitch 65: conflict-free; row c: a broadcast) and four accumulators; the pivot
                // s_c reaches every lane through v_readlane.  (Until round 4 this was the right-looking form: per column every
                // lane ran a serial read-modify-write loop over its row in LDS -- p^2 / 2 dependent round trips, a fifth of the
                // time an iteration of a 5 000 x 50 partition took.)
                const int jj = min(j, pe - 1);
                const double* __restrict__ rj = Ls + jj * SM_LD;
                auto bcast = [&](double v, int src) {      // lane src's value to every lane (src uniform)
                    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), src), __builtin_amdgcn_readlane(__double2loint(v), src));
                };
                bool ok = true;
                double dg = 1.0;                            // L[j][j]
                for (int c = 0; c < pe; ++c) {
                    const double* __restrict__ rc = Ls + c * SM_LD;
                    double s0 = Hs[jj * SM_LD + c], s1 = 0.0, s2 = 0.0, s3 = 0.0;
                    int kk = 0;
                    for (; kk + 15 < c; kk += 16) {          // sixteen pairs of loads in flight: a trip costs one LDS round trip, not four
                        double a[16], b[16];
#pragma unroll
                        for (int q = 0; q < 16; ++q) { a[q] = rj[kk + q]; b[q] = rc[kk + q]; }
#pragma unroll
                        for (int q = 0; q < 16; q += 4) {
                            s0 = fma(-a[q], b[q], s0);
                            s1 = fma(-a[q + 1], b[q + 1], s1);
                            s2 = fma(-a[q + 2], b[q + 2], s2);
                            s3 = fma(-a[q + 3], b[q + 3], s3);
                        }
                    }
                    for (; kk + 3 < c; kk += 4) {
                        s0 = fma(-rj[kk], rc[kk], s0);
                        s1 = fma(-rj[kk + 1], rc[kk + 1], s1);
                        s2 = fma(-rj[kk + 2], rc[kk + 2], s2);
                        s3 = fma(-rj[kk + 3], rc[kk + 3], s3);
                    }
                    for (; kk < c; ++kk) s0 = fma(-rj[kk], rc[kk], s0);
                    const double sj = (s0 + s1) + (s2 + s3);
                    const double d = bcast(sj, c);
                    if (!(d > 0.0) || !isfinite(d)) { ok = false; break; }
                    const double sd = sqrt(d);
                    if (j == c) dg = sd;
                    if (j >= c && j < pe) Ls[j * SM_LD + c] = (j == c) ? sd : sj / sd;
                    wave_sync();                            // column c is in LDS: the next columns read it (this wave's LDS operations run in order)
                }
                if (!ok) { state = 2; if (lane == 0) sc[2] = DLSA_PART_NOT_SPD; }
                else {
                    // the two triangular solves with the right-hand side in registers (lane j holds entry j), the entry being
                    // eliminated broadcast by v_readlane: no LDS write, no synchronisation inside the loops
                    const double rdg = 1.0 / dg;
                    double bv = j < pe ? gs[j] : 0.0;
                    for (int c = 0; c < pe; ++c) {          // L z = g
                        const double z = bcast(bv, c) * bcast(rdg, c);
                        const double ljc = rj[c];
                        if (j == c) bv = z; else if (j > c) bv = fma(-ljc, z, bv);
                    }
                    for (int c = pe - 1; c >= 0; --c) {     // L' delta = z
                        const double z = bcast(bv, c) * bcast(rdg, c);
                        const double lcj = Ls[c * SM_LD + jj];
                        if (j == c) bv = z; else if (j < c) bv = fma(-lcj, z, bv);
                    }
                    const double dj = j < pe ? bv : 0.0, bj = j < pe ? beta[j] : 0.0;
                    const double dmax = wave_allreduce_max(fabs(dj)), bmax = wave_allreduce_max(fabs(bj));
                    if (!isfinite(dmax)) { state = 2; if (lane == 0) sc[2] = DLSA_PART_NAN; }
                    else if (dmax <= a.tol * fmax(1.0, bmax)) state = 1;
                    else if (j < pe) { prev[j] = bj; stepv[j] = dj; beta[j] = bj + dj; }
                }
            }
            if (lane == 0) sc[1] = (double)state;
        }
        __syncthreads();
        if (C > 1) {
            // workgroup 0 publishes the new beta and the loop state; the others pick them up after the second barrier
            double* __restrict__ bs = a.bcast + (int64_t)k * SM_BSLOT;
            if (leader) {
                if (tid < pe) bs[tid] = beta[tid];
                if (tid == 0) { bs[SM_MAXP] = sc[1]; bs[SM_MAXP + 1] = sc[2]; bs[SM_MAXP + 2] = ll; }
            }
            if (!cluster_barrier(cbar, a.bar_abort, (unsigned)C, bphase, a.bar_timeout)) return;
            if (!leader) {
                if (tid < pe) beta[tid] = bs[tid];
                if (tid == 0) { sc[1] = bs[SM_MAXP]; sc[2] = bs[SM_MAXP + 1]; }
                ll = bs[SM_MAXP + 2];
                __syncthreads();
            }
        }
        const int state = (int)sc[1];
        if (state == 3) {                                   // re-evaluate at the halved step (not a new iteration)
            ++halvings;
            if (evals > 2 * a.max_iter + 64) break;
            continue;
        }
        ++iters;
        if (state == 1) { status = DLSA_PART_OK; break; }
        if (state == 2) { status = (int)sc[2]; break; }
        ll_prev = ll; have_prev = true; halvings = 0;
        if (iters >= a.max_iter) last_pass = true;
    }
    if (!leader) return;
    // outputs: intercept first (models.py:136-142)
    auto omap = [&](int o) { return a.icpt ? (o == 0 ? pe - 1 : o - 1) : o; };
    for (int e = tid; e < pe * pe; e += SM_THREADS) {
        const int i = e / pe, j = e - i * pe;
        a.sig[(int64_t)k * pe * pe + e] = Hs[omap(i) * SM_LD + omap(j)];
    }
    for (int o = tid; o < pe; o += SM_THREADS) {
        const int i = omap(o);
        a.coef[(int64_t)k * pe + o] = beta[i];
        double s = 0.0;
        for (int j = 0; j < pe; ++j) s = fma(Hs[i * SM_LD + j], beta[j], s);      // Sig_inv . coef (models.py:131)
        a.smc[(int64_t)k * pe + o] = s;
    }
    if (tid == 0) { a.n_iter[k] = iters; a.status[k] = status; a.loglik[k] = ll; }
}

std::atomic<int> g_small_cluster_aborts{0};
std::mutex g_small_cluster_mu;                                         // serialises this process's clustered launches (launch .. completion)
std::atomic<long long> g_small_cluster_timeout_ticks{25000000ll};       // 0.25 s of the 100 MHz wall clock (a barrier wait is microseconds; cooperative launches cannot miss co-residency)

// Workgroups per partition: as many as leave every CU at most one, while a workgroup keeps >= 512 rows (below that the two cluster
// barriers and the partial sums cost what the shorter pass saves).  DLSA_IRLS_SMALL_CLUSTER = C forces it (1 .. 16).
static int small_cluster_count(int K, int64_t nmax) {
    int C = (int)std::max<int64_t>(1, std::min<int64_t>(std::min(8, kNumCU / std::max(K, 1)), nmax / 512));
    // sixteen only for long partitions: workgroup 0 adds 15 partials of 33 KB and the barriers span 16 CUs (K = 4: 5000 rows 0.72 ms at
    // C = 8 / 1.02 at 16, 20000 rows 1.08 / 1.18, 60000 rows 2.01 / 1.64)
    if (C == 8 && kNumCU / std::max(K, 1) >= 16 && nmax >= 30000) C = 16;
    if (const char* e = knob("DLSA_IRLS_SMALL_CLUSTER")) C = std::max(1, std::min(16, atoi(e)));
    if ((int64_t)K * C > kNumCU) C = 1;                  // (co-residency: a CU takes one workgroup of this kernel -- 256 registers x 512 threads)
    return C;
}

bool irls_small_enabled() {
    const char* e = knob("DLSA_IRLS_SMALL");       // 0: always the host-driven path (valid results, A/B runs)
    return e ? atoi(e) != 0 : true;
}

// One launch with a workgroup per partition, or the host-driven path that gives every partition the whole GPU in turn?
// Measured (ms, p ~ 50): this kernel 1.0 + 1.9e-4 n_k whatever K (up to one workgroup per CU), the host-driven path 0.55 K
// (per-iteration launch + synchronisation latency): K = 20 x 5 000 rows 1.9 vs 12.9, K = 200 x 5 000 2.6 vs 125,
// K = 20 x 50 000 10.5 vs 11.3, K = 8 x 20 000 5.4 vs 5.4, K = 4 x 60 000 7.1 vs 2.3.
// est_ms (nullable): the cost model's estimate for this kernel, 0 when forced -- the caller weighs it against the lock step's
bool irls_small_eligible(const int64_t* rows_host, int K, int pe, double* est_ms) {
    if (est_ms) *est_ms = 0.0;
    if (!irls_small_enabled() || pe > SM_MAXP || K < 2) return false;
    int64_t nmax = 0;
    for (int k = 0; k < K; ++k) nmax = std::max(nmax, rows_host[k]);
    if (nmax > 65536) return false;
    if (const char* f = knob("DLSA_IRLS_SMALL")) if (atoi(f) == 2) return true;      // 2: this kernel whatever the cost model says (A/B runs)
    const int C = small_cluster_count(K, nmax);
    const double rounds = (double)(((int64_t)K * C + kNumCU - 1) / kNumCU);
    // the host-driven path fits such partitions on up to four concurrent chains (irls.hip, irls_fit_core): 0.55 ms per partition on one
    // chain, 0.23 on four (K = 20, p = 50, 10000 .. 60000 rows: 4.5 ms whatever the row count).  This kernel (round 4: clusters,
    // left-looking Cholesky; bench/small_ab.py, p = 50, ms): 0.8 + 1.9e-4 n_k / C -- K = 20: 1.01 / 1.28 / 2.05 at 5000 / 20000 /
    // 50000 rows (C = 8), K = 100: 1.36 / 2.73 / 6.66 at 5000 / 20000 / 60000 (C = 2), K = 200 x 5000: 1.72 (C = 1)
    const char* e = knob("DLSA_IRLS_CHAINS");
    const int cap = e ? std::min(8, std::max(1, atoi(e))) : 4;
    const int S = std::max(1, std::min(cap, (K - 1) / 2));
    static const double per_partition_ms[5] = {0.55, 0.55, 0.37, 0.29, 0.23};
    const double t_small = ((C > 8 ? 1.0 : 0.8) + 1.9e-4 * ((double)nmax / C) * std::max(0.5, pe / 50.0)) * rounds, t_host = per_partition_ms[std::min(S, 4)] * K;
    if (est_ms) *est_ms = t_small;
    return t_small < t_host;
}

size_t irls_small_workspace_bytes(int K) { return align_up((size_t)K * (2 * sizeof(int64_t) + 2 * sizeof(int) + sizeof(double)), 256) + 256; }

int irls_small_fit(const double* X, int64_t ldx, const double* y, const int64_t* first_host, const int64_t* rows_host,
                   int64_t step, int K, int p, int intercept, double tol, int max_iter, double* coef, double* Sig_inv,
                   double* Sig_invMcoef, int* n_iter_host, int* status_host, double* loglik_host, void* ws, size_t ws_bytes,
                   hipStream_t s) {
    if (!ws || ws_bytes < irls_small_workspace_bytes(K) || ((uintptr_t)ws & 255)) {
        set_error("irls_fit: workspace %zu bytes needed (256-aligned), got %zu", irls_small_workspace_bytes(K), ws_bytes);
        return DLSA_ERR_WORKSPACE;
    }
    char* base = (char*)ws;
    int64_t* d_first = (int64_t*)base;
    int64_t* d_rows = d_first + K;
    double* d_ll = (double*)(d_rows + K);
    int* d_iter = (int*)(d_ll + K);
    int* d_status = d_iter + K;
    DLSA_HIP_CHECK(hipMemcpyAsync(d_first, first_host, (size_t)K * sizeof(int64_t), hipMemcpyHostToDevice, s));
    DLSA_HIP_CHECK(hipMemcpyAsync(d_rows, rows_host, (size_t)K * sizeof(int64_t), hipMemcpyHostToDevice, s));
    SmallArgs a;
    a.X = X; a.y = y; a.first = d_first; a.rows = d_rows; a.ldx = ldx; a.step = step; a.p = p; a.icpt = intercept ? 1 : 0;
    a.pe = p + a.icpt; a.max_iter = max_iter; a.tol = tol; a.coef = coef; a.sig = Sig_inv; a.smc = Sig_invMcoef;
    a.n_iter = d_iter; a.status = d_status; a.loglik = d_ll;
    const int nt = (a.pe + 15) / 16;
    int64_t nmax = 0;
    for (int k = 0; k < K; ++k) nmax = std::max(nmax, rows_host[k]);
    int C = small_cluster_count(K, nmax);
    PoolBlock blk;                 // (released on every return below, the error ones too)
    for (int attempt = 0; attempt < 2; ++attempt) {
        a.C = C; a.scratch = nullptr; a.bcast = nullptr; a.bar = nullptr; a.bar_abort = nullptr; a.bar_timeout = g_small_cluster_timeout_ticks.load();
        if (C > 1) {
            const size_t b_scr = align_up((size_t)K * C * SM_SLOT * sizeof(double), 256), b_bc = align_up((size_t)K * SM_BSLOT * sizeof(double), 256),
                         b_bar = align_up(((size_t)K * 16 + 16) * sizeof(unsigned), 256);
            DLSA_HIP_CHECK(blk.alloc(b_scr + b_bc + b_bar, s));
            char* pool = blk.p;
            a.scratch = (double*)pool; a.bcast = (double*)(pool + b_scr); a.bar = (unsigned*)(pool + b_scr + b_bc); a.bar_abort = a.bar + (size_t)K * 16;
            DLSA_HIP_CHECK(hipMemsetAsync(a.bar, 0, b_bar, s));
        }
        // clustered launches of this process run one at a time: two of them could each hold part of the CUs the other's queued
        // workgroups need (a CU takes one workgroup of this kernel).  Other processes on the GPU are covered by the timeout alone.
        std::unique_lock<std::mutex> cluster_lock(g_small_cluster_mu, std::defer_lock);
        if (C > 1) cluster_lock.lock();
        const dim3 grid((unsigned)(K * C));
        // clusters meet at device barriers: the plain launch with its bounded barrier (rerun on one workgroup per partition when a wait
        // times out), or -- dlsa_kernel_options.cooperative = 1 -- a cooperative launch (every workgroup resident, or a clean refusal)
        const void* fn = nt == 1 ? (const void*)irls_small_kernel<1> : nt == 2 ? (const void*)irls_small_kernel<2> :
                         nt == 3 ? (const void*)irls_small_kernel<3> : (const void*)irls_small_kernel<4>;
        void* kargs[] = {(void*)&a};
        if (C == 1 || launch_cooperative(fn, grid, dim3(SM_THREADS), kargs, 0, s) != hipSuccess) {
            switch (nt) {
                case 1: hipLaunchKernelGGL(irls_small_kernel<1>, grid, dim3(SM_THREADS), 0, s, a); break;
                case 2: hipLaunchKernelGGL(irls_small_kernel<2>, grid, dim3(SM_THREADS), 0, s, a); break;
                case 3: hipLaunchKernelGGL(irls_small_kernel<3>, grid, dim3(SM_THREADS), 0, s, a); break;
                default: hipLaunchKernelGGL(irls_small_kernel<4>, grid, dim3(SM_THREADS), 0, s, a); break;
            }
        }
        DLSA_HIP_CHECK(hipGetLastError());
        if (C == 1) break;
        // a cluster barrier that timed out (the workgroups of a partition were not all resident) set the abort word: nothing
        // was written; run again with one workgroup per partition, which needs no co-residency
        unsigned aborted = 0;
        DLSA_HIP_CHECK(hipMemcpyAsync(&aborted, a.bar_abort, sizeof(unsigned), hipMemcpyDeviceToHost, s));
        DLSA_HIP_CHECK(hipStreamSynchronize(s));
        blk.release();
        if (!aborted) break;
        g_small_cluster_aborts.fetch_add(1);
        C = 1;
    }
    std::vector<int> hi((size_t)K), hs((size_t)K);
    std::vector<double> hl((size_t)K);
    DLSA_HIP_CHECK(hipMemcpyAsync(hi.data(), d_iter, (size_t)K * sizeof(int), hipMemcpyDeviceToHost, s));
    DLSA_HIP_CHECK(hipMemcpyAsync(hs.data(), d_status, (size_t)K * sizeof(int), hipMemcpyDeviceToHost, s));
    DLSA_HIP_CHECK(hipMemcpyAsync(hl.data(), d_ll, (size_t)K * sizeof(double), hipMemcpyDeviceToHost, s));
    DLSA_HIP_CHECK(hipStreamSynchronize(s));
    int overall = DLSA_OK;
    for (int k = 0; k < K; ++k) {
        if (n_iter_host) n_iter_host[k] = hi[k];
        if (status_host) status_host[k] = hs[k];
        if (loglik_host) loglik_host[k] = hl[k];
        if (hs[k] == DLSA_PART_NOT_CONVERGED && overall == DLSA_OK) overall = DLSA_ERR_NOT_CONVERGED;
        if (hs[k] == DLSA_PART_NOT_SPD && overall == DLSA_OK) overall = DLSA_ERR_NOT_SPD;
        if (hs[k] == DLSA_PART_NAN && overall == DLSA_OK) overall = DLSA_ERR_NAN;
    }
    if (overall == DLSA_ERR_NOT_CONVERGED) set_error("irls_fit: at least one partition hit max_iter");
    if (overall == DLSA_ERR_NOT_SPD) set_error("irls_fit: a partition's Hessian is not positive definite");
    if (overall == DLSA_ERR_NAN) set_error("irls_fit: NaN/Inf in a partition's fit");
    return overall;
}

}  // namespace dlsa

extern "C" {

// Test / diagnostics hook for the bounded cluster barrier of irls_small_kernel (no reference counterpart: the reference fits its
// partitions in Spark tasks, dlsa/models.py:110-131): sets the barrier timeout in seconds (<= 0 restores the 0.25 s default) and
// returns how many clustered launches of this process have been given up and rerun with one workgroup per partition.
int dlsa_irls_small_cluster_timeout(double seconds) {
    dlsa::g_small_cluster_timeout_ticks.store(seconds > 0.0 ? (long long)(seconds * 1e8) + 1 : 25000000ll);
    return dlsa::g_small_cluster_aborts.load();
}

}  // extern "C"
