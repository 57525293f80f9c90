// LARS / lasso path of the least-squares approximation up to LARS_Q_MAX_M penalised variables (m = p - intercept): one workgroup (or a few),
// the Cholesky rows of ALL variables carried along the path (reference: lars_lsa and updateR, dlsa/lsa.py:12-32, 90-212).
//
// lars.hip keeps R^{-1} and, per step, gathers x = Sigma[active, new], forms r = R^{-T} x, the new column of R^{-1} and
// u = Sigma[:, active] w: three dependent mat-vecs and ~12 workgroup barriers per step, 8.9 us per step at p = 100.  Here row i of
//     Q = R^{-T} Sigma[active, :]            (na x m; row i is written when position i is appended)
// is the forward substitution of lsa.py:17 done for EVERY variable j at once:
//     Q[na][j] = (Sigma[new][j] - sum_{i<na} Q[i][new] Q[i][j]) / r_pp,     r_pp^2 = Sigma[new][new] - |Q[:, new]|^2   (lsa.py:18-19)
// so that r = R^{-T} x of the next append is a COLUMN of Q (no mat-vec), and
//     Sigma[:, active] w = A Q' t,   t = R^{-T} sign                                   (lsa.py:151-153, 157-160, 177)
// is carried as v = Q' t (v += Q[na][:] t_na on every append): O(m) per step.  The direction itself needs R^{-1}: its new column
// c = -R^{-1} r / r_pp (lars.hip's rule) is the same kind of mat-vec as the new row of Q -- rows of RT = (R^{-1})' times r -- so
// both run as ONE pass with one pair of barriers: a thread owns two adjacent columns of [Q | RT], the rows are split over thread
// groups, partial sums meet in LDS in a fixed order (deterministic).  A step is then
//     select (every wave scans Cvec in LDS redundantly: no barrier) -> gather r, |r|^2 and r.t per wave (1 barrier) ->
//     the fused pass (2) -> step length and lasso crossing (one min-reduction, 2) -> move + path record + next Cmax (one reduction, 2).
// For m <= ~108 Q and the packed RT live in LDS (160 KB); wider problems keep them in global memory (L2) with 8 row loads in
// flight per thread, and beyond m = 200 a few workgroups (CLUSTERS) deal the row groups of the pass among them: one bounded grid
// barrier per append, partial sums through global memory added in a fixed order, everything else replicated (QGrid, fused_mv).  A lasso drop (lsa.py:179-186) truncates Q / RT to the positions before the first dropped one, recomputes v and
// Gi1 from them with the same fused pass and appends the kept positions again (the factor of an ordered set is unique), as lars.hip.
#include "common.h"
#include "lars.h"
#include "options.h"
#include <math.h>
#include <stdlib.h>
#include <algorithm>

namespace dlsa {
namespace {

// Optional phase timer (-DDLSA_LARS_PROF): thread 0 accumulates wall-clock ticks (100 MHz) per phase and prints them.
#ifdef DLSA_LARS_PROF
__shared__ long long q_prof_t[16];
__shared__ long long q_prof_last;
__shared__ long long q_prof_c0, q_prof_w0;
#define QPROF_DECL do { if (threadIdx.x == 0) { for (int q_ = 0; q_ < 16; ++q_) q_prof_t[q_] = 0; q_prof_last = wall_clock64(); q_prof_w0 = q_prof_last; q_prof_c0 = clock64(); } } while (0)
#define QTICK(i) do { if (threadIdx.x == 0) { const long long now_ = wall_clock64(); q_prof_t[i] += now_ - q_prof_last; q_prof_last = now_; } } while (0)
#else
#define QPROF_DECL
#define QTICK(i)
#endif

// Workgroup barrier for data exchanged through LDS.  __syncthreads() also drains the wave's global loads and stores (vmcnt(0)):
// with it every step waited for its path-record stores and for the prefetched row of S at the next barrier (~1.5 us per step).
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// The workgroups that share a fused pass (clusters, m > 108): lars.hip's bounded grid barrier -- a monotonic counter with agent-scope
// release / acquire, relaxed polling with ONE acquire after the match; a wait longer than the timeout sets the abort word and every
// workgroup leaves (the host then runs the path on one workgroup).
struct QGrid {
    int nwg, wg, xld;
    double* xbuf;            // [2][nwg][xld]: the workgroups' partial sums of a pass, two buffers used alternately
    unsigned* bar;
    long long timeout;
    unsigned phase;          // barrier count of this launch
    int passes;              // exchanged passes so far (buffer parity)
    bool failed;
};
__device__ __forceinline__ bool q_grid_barrier(QGrid& gr) {
    __shared__ int gb_ok;
    __syncthreads();                 // this workgroup's global stores have been issued by every wave
    if (threadIdx.x == 0) {
        ++gr.phase;
        const unsigned target = gr.phase * (unsigned)gr.nwg;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(gr.bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        bool ok = true;
        const long long t0 = wall_clock64();
        unsigned spins = 0;
        while (__hip_atomic_load(gr.bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if ((spins++ & 255u) == 0u) {
                if (__hip_atomic_load(gr.bar + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { ok = false; break; }
                if (wall_clock64() - t0 > gr.timeout) {
                    __hip_atomic_store(gr.bar + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ok = false;
                    break;
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        gb_ok = ok ? 1 : 0;
    } else {
        ++gr.phase;
    }
    __syncthreads();
    return gb_ok != 0;
}

// Values every lane holds alike (sums reduced over the wave, LDS words read by all lanes) that steer control flow or count loops:
// moved to a scalar register, or the compiler predicates every loop and branch they touch with lane masks.
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ bool uni(bool b) { return __builtin_amdgcn_readfirstlane((int)b) != 0; }

// offset of row l of the packed RT (row l holds columns 0 .. l, padded to an even count)
__host__ __device__ __forceinline__ int rt_off_packed(int l) { const int h = l >> 1; return 2 * h * (h + 1) + (l & 1) * (2 * h + 2); }

template <int T>
struct QBlock {
    static constexpr int WAVES = T / 64;
    // block-wide reduction of NA values with opa and NB values with opb (NA + NB <= 4); result to all threads.  `red` holds two
    // buffers of 4 * WAVES doubles used alternately (`phase` counts the calls, the same in every thread): a call writes the buffer
    // that was last read two calls ago, before the barrier of the call in between -- one barrier per reduction.
    template <int NA, typename OpA, int NB, typename OpB>
    static __device__ __forceinline__ void reduce2(double (&va)[NA], OpA opa, double (&vb)[NB], OpB opb, double* red, int& phase) {
        static_assert(NA + NB <= 4, "a buffer holds four values per wave");
#pragma unroll
        for (int q = 0; q < NA; ++q) va[q] = wave_allreduce(va[q], opa);
#pragma unroll
        for (int q = 0; q < NB; ++q) vb[q] = wave_allreduce(vb[q], opb);
        double* r = red + (phase & 1) * 4 * WAVES;
        ++phase;
        if ((threadIdx.x & 63) == 0) {
#pragma unroll
            for (int q = 0; q < NA; ++q) r[q * WAVES + (threadIdx.x >> 6)] = va[q];
#pragma unroll
            for (int q = 0; q < NB; ++q) r[(NA + q) * WAVES + (threadIdx.x >> 6)] = vb[q];
        }
        lds_barrier();
#pragma unroll
        for (int q = 0; q < NA; ++q) {
            double s = r[q * WAVES];
            for (int k = 1; k < WAVES; ++k) s = opa(s, r[q * WAVES + k]);
            va[q] = s;
        }
#pragma unroll
        for (int q = 0; q < NB; ++q) {
            double s = r[(NA + q) * WAVES];
            for (int k = 1; k < WAVES; ++k) s = opb(s, r[(NA + q) * WAVES + k]);
            vb[q] = s;
        }
    }
};

// The cross-wave half of a block reduction: v[q] is already the same in every lane of a wave (reduced there, or a ballot count);
// OPS[q] = 0 sum, 1 min, 2 max.  One barrier, alternating buffers as QBlock::reduce2.
template <int T, int N>
__device__ __forceinline__ void cross_wave(double (&v)[N], const int (&ops)[N], double* red, int& phase) {
    constexpr int WAVES = T / 64;
    static_assert(N <= 4, "a buffer holds four values per wave");
    double* r = red + (phase & 1) * 4 * WAVES;
    ++phase;
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int q = 0; q < N; ++q) r[q * WAVES + (threadIdx.x >> 6)] = v[q];
    }
    lds_barrier();
#pragma unroll
    for (int q = 0; q < N; ++q) {
        double s = r[q * WAVES];
        for (int k = 1; k < WAVES; ++k) {
            const double u = r[q * WAVES + k];
            s = ops[q] == 0 ? s + u : (ops[q] == 1 ? fmin(s, u) : fmax(s, u));
        }
        v[q] = s;
    }
}

// 1 / sqrt(d) to about an ulp: v_rsq_f64 and two Newton steps (sqrt + division are ~45 dependent instructions, and every wave
// of the workgroup evaluates the step's scalars itself)
__device__ __forceinline__ double rsqrt_newton(double d) {
    double y = __builtin_amdgcn_rsq(d);
    double e = fma(-d * y, y, 1.0);
    y = fma(0.5 * y, e, y);
    e = fma(-d * y, y, 1.0);
    return fma(0.5 * y, e, y);
}

// A thread's place in the fused pass, fixed for the whole path: threads [0, NQ * GQ) take the NQ column pairs of Q in GQ row
// groups, the next NR * GR threads the NR column pairs of RT in GR row groups (GR = 1 or even: the packed RT's row offsets then
// advance by a second-order recurrence, no multiplication per row).  RT's rows are half as long on average, so GQ > GR fills the
// workgroup better than one G for both.  i0 = the first row the thread reads (RT's column pair (c2, c2 + 1) exists from row c2
// on), `stride` = the distance between the row groups' partial sums of one column pair in `part` (indexed by thread).
struct QLayout {
    int G, Gl, g, c2, i0, stride;      // G: row stride of a thread's walk (all workgroups' row groups), Gl: this workgroup's row groups
    bool isq, live;
};
__device__ __forceinline__ QLayout q_layout(int T, int tid, int m, bool lds, int nwg, int wg) {
    QLayout L;
    const int NQ = ((m + 1) & ~1) >> 1, NR = (m + 2) >> 1;
    int GQ = 1, GR = 1;
    if (lds) {
        // the split that minimises the longest walk: n / GQ rows for a Q thread, about n / (2 GR) for an RT thread
        float best = 1e30f;
        for (int gr = 1; gr <= 16; gr = gr == 1 ? 2 : gr + 2) {
            const int gq = (T - NR * gr) / NQ;
            if (gq < 1) break;
            const float cost = fmaxf(1.0f / (float)gq, 0.55f / (float)gr);
            if (cost < best) { best = cost; GQ = gq; GR = gr; }
        }
    } else {
        // Q and RT in global memory: the pass is bound by what one CU streams from L2 (~75 GB/s in 16-byte loads); more loading
        // threads only lengthen the slowest walk (p = 260, GQ = 2 / GR = 1: 2.35 ms; one G for both: 2.13 ms) -- more CUs do
        // help: nwg workgroups deal the row groups among them (group wg * Gl + g of nwg * Gl)
        GQ = GR = max(1, T / (NQ + NR));
    }
    const int nq_threads = NQ * GQ;
    L.isq = tid < nq_threads;
    const int u = L.isq ? tid : tid - nq_threads;
    const int width = L.isq ? NQ : NR;
    L.Gl = L.isq ? GQ : GR;
    L.G = L.Gl * nwg;
    L.stride = width;
    L.g = u / width;
    const int jx = u - L.g * width;
    L.live = L.g < L.Gl;
    L.c2 = 2 * jx;
    const int gidx = wg * L.Gl + L.g;
    L.i0 = gidx;
    if (!L.isq && L.c2 > gidx) L.i0 = gidx + (L.c2 - gidx + L.G - 1) / L.G * L.G;
    return L;
}

// out = M' x over the first n rows of two row-major matrices that grow by appended rows: Qm (stride ld, NQ column pairs) and RT
// (row l holds columns 0 .. l; ncolR columns are produced); x is read with stride xs.  eq(j, {sum_j, sum_j+1}) / er(i, {sum_i,
// sum_i+1}) run on one thread per column pair after the partial sums of the row groups have met in LDS.  Two barriers; n = 0 is
// allowed.  Per row a thread spends two pointer additions, two loads and two FMAs (an integer multiplication costs four FMAs'
// issue time with one wave per SIMD): row offsets advance by constants, the packed RT's by a running difference.
template <int T, bool LDSQ, typename EQ, typename ER>
__device__ __forceinline__ void fused_mv(const QLayout& L, QGrid& gr, const double* __restrict__ Qm, int ld, const double* __restrict__ RT, int ncolR,
                                         int n, const double* __restrict__ x, int xs, double2* part, EQ&& eq, ER&& er) {
    const bool mine = L.live && (L.isq || L.c2 < ncolR);
    if (mine) {
        double2 a0 = {0.0, 0.0}, a1 = a0, a2 = a0, a3 = a0;
        // row loads in flight per thread: L2 needs ~64 KB in flight to stream at a CU's rate, and beyond m = 255 a column pair
        // has one thread (G = 1); 128 registers per thread bound the 1024-thread build
#ifndef DLSA_LARS_Q_NB_GLOBAL
#define DLSA_LARS_Q_NB_GLOBAL 8
#endif
        constexpr int NB = T >= 1024 ? 4 : (LDSQ ? 8 : DLSA_LARS_Q_NB_GLOBAL);
        const int G = L.G;
        int i = L.i0;
        const double* __restrict__ xp = x + i * xs;
        const int xstep = G * xs;
        // row pointer and its step: constant for Q and for a square RT; for the packed RT (row l at 2h(h+1) + (l&1)(2h+2), h = l/2)
        // the step from row l to l + G is linear in l, so it advances by G * G per row group (G even) or is len(l) itself (G = 1)
        const bool packed = LDSQ && !L.isq;
        const double* __restrict__ rp = (L.isq ? Qm : RT) + L.c2 + (packed ? rt_off_packed(i) : i * ld);
        int step = packed ? rt_off_packed(i + G) - rt_off_packed(i) : G * ld;
        const int step2 = packed ? (G == 1 ? 0 : G * G) : 0;
        auto next = [&]() {
            rp += step;
            if (packed && G == 1) step = (i + G + 2) & ~1;      // len(row i + 1) once i has advanced; see the callers' order
            else step += step2;
            xp += xstep;
            i += G;
        };
        while (i + (NB - 1) * G < n) {
            double2 q[NB];
            double xv[NB];
#pragma unroll
            for (int k = 0; k < NB; ++k) {
                q[k] = *reinterpret_cast<const double2*>(rp);
                xv[k] = *xp;
                next();
            }
#pragma unroll
            for (int k = 0; k < NB; k += 4) {
                a0.x = fma(xv[k], q[k].x, a0.x); a0.y = fma(xv[k], q[k].y, a0.y);
                a1.x = fma(xv[k + 1], q[k + 1].x, a1.x); a1.y = fma(xv[k + 1], q[k + 1].y, a1.y);
                a2.x = fma(xv[k + 2], q[k + 2].x, a2.x); a2.y = fma(xv[k + 2], q[k + 2].y, a2.y);
                a3.x = fma(xv[k + 3], q[k + 3].x, a3.x); a3.y = fma(xv[k + 3], q[k + 3].y, a3.y);
            }
        }
        if (i < n) {      // the last, partial batch: the same independent loads under a row mask (singly they would each cost a round trip)
            double2 q[NB];
            double xv[NB];
#pragma unroll
            for (int k = 0; k < NB; ++k) {
                q[k] = double2{0.0, 0.0};
                xv[k] = 0.0;
                if (i < n) {
                    q[k] = *reinterpret_cast<const double2*>(rp);
                    xv[k] = *xp;
                    next();
                }
            }
#pragma unroll
            for (int k = 0; k < NB; k += 4) {
                a0.x = fma(xv[k], q[k].x, a0.x); a0.y = fma(xv[k], q[k].y, a0.y);
                a1.x = fma(xv[k + 1], q[k + 1].x, a1.x); a1.y = fma(xv[k + 1], q[k + 1].y, a1.y);
                a2.x = fma(xv[k + 2], q[k + 2].x, a2.x); a2.y = fma(xv[k + 2], q[k + 2].y, a2.y);
                a3.x = fma(xv[k + 3], q[k + 3].x, a3.x); a3.y = fma(xv[k + 3], q[k + 3].y, a3.y);
            }
        }
        part[threadIdx.x] = double2{(a0.x + a1.x) + (a2.x + a3.x), (a0.y + a1.y) + (a2.y + a3.y)};
    }
    QTICK(4);
    lds_barrier();
    QTICK(5);
    double2 t = {0.0, 0.0};
    if (mine && L.g == 0) {
        t = part[threadIdx.x];
        for (int q = 1; q < L.Gl; ++q) { const double2 u = part[threadIdx.x + q * L.stride]; t.x += u.x; t.y += u.y; }
    }
    if (gr.nwg > 1) {
        // the workgroups' partial sums meet in global memory: publish, one grid barrier, add in the order 0 .. nwg - 1 (every
        // workgroup the same numbers: the emits below are replicated)
        const int pos = (L.isq ? 0 : ld) + L.c2;
        double* __restrict__ buf = gr.xbuf + (size_t)(gr.passes & 1) * gr.nwg * gr.xld;
        if (mine && L.g == 0) *reinterpret_cast<double2*>(buf + (size_t)gr.wg * gr.xld + pos) = t;
        ++gr.passes;
        if (!q_grid_barrier(gr)) { gr.failed = true; return; }
        if (mine && L.g == 0) {
            t = *reinterpret_cast<const double2*>(buf + pos);
            for (int w = 1; w < gr.nwg; ++w) { const double2 u = *reinterpret_cast<const double2*>(buf + (size_t)w * gr.xld + pos); t.x += u.x; t.y += u.y; }
        }
    }
    if (mine && L.g == 0) { if (L.isq) eq(L.c2, t); else er(L.c2, t); }
    QTICK(6);
    // the emitted rows are read by other threads in later passes: through LDS, or (global variant) through the CU's L1 / L2, for
    // which the stores must have completed
    if constexpr (LDSQ) lds_barrier(); else __syncthreads();
    QTICK(7);
}

__host__ __device__ inline size_t lars_q_lds_doubles(int m, int T, bool ldsq) {
    const size_t ld = (size_t)((m + 1) & ~1);
    size_t n = 2 * (size_t)T + 11 * ld + 2 * ld;                          // part | eleven double vectors | four int vectors
    if (ldsq) n += (size_t)m * ld + (size_t)rt_off_packed(m) + 16;         // Q | packed RT
    return n;
}

template <int T, bool LDSQ>
__global__ __launch_bounds__(T) void lars_q_kernel(LarsArgs a) {
    using B = QBlock<T>;
    __shared__ double red[2 * 4 * B::WAVES];
    __shared__ int sh_i[4];
    int red_phase = 0;
    extern __shared__ __attribute__((aligned(16))) double dyn[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int p = a.p;
    const int off = a.intercept ? 1 : 0;
    const int m = p - off;
    const int ld = (m + 1) & ~1;
    const int NQ = ld >> 1;
    const double eps = a.eps;
    double2* part = reinterpret_cast<double2*>(dyn);
    double* cvec = dyn + 2 * T;          // Sigma (b - beta), by variable
    double* beta = cvec + ld;            // current (scaled) coefficients
    double* v = beta + ld;               // Q' t
    double* bsgn = v + ld;               // sign(b0)
    double* absb = bsgn + ld;            // |b0|
    double* a12 = absb + ld;             // intercept row of Sigma0
    double* sgn = a12 + ld;              // sign of the correlation at entry, by active position
    double* tv = sgn + ld;               // R^{-T} sgn, by active position
    double* gi1 = tv + ld;               // R^{-1} R^{-T} sgn, by active position
    double* rv = gi1 + ld;               // column `new` of Q
    double* lastq = rv + ld;             // the row of Q the last append made (clusters: not yet visible in global memory to the others)
    int* state = reinterpret_cast<int*>(lastq + ld);   // 0 inactive, 1 active, 2 ignored
    int* act = state + ld;               // active list (variable ids)
    int* dropf = act + ld;               // by active position
    int* pos = dropf + ld;               // position in the active list, m when not active
    double* Q;
    double* RT;
    if constexpr (LDSQ) {
        Q = reinterpret_cast<double*>(pos + ld);
        RT = Q + (size_t)m * ld;
    } else {
        Q = a.RinvT;
        RT = a.Rinv;
    }
    auto rt_off = [&](int l) { return LDSQ ? rt_off_packed(l) : l * ld; };
    QGrid grid;
    grid.nwg = LDSQ ? 1 : max(1, a.nwg);
    grid.wg = LDSQ ? 0 : (int)blockIdx.x;
    grid.xld = 2 * ld;
    grid.xbuf = a.upart;
    grid.bar = a.bar;
    grid.timeout = a.bar_timeout;
    grid.phase = 0; grid.passes = 0; grid.failed = false;
    const bool writer = grid.wg == 0;              // workgroup 0 writes the path (every workgroup computes it)
    const QLayout L = q_layout(T, tid, m, LDSQ, grid.nwg, grid.wg);
    // the workgroup that owns row i of Q / RT (reads it in its passes, so writes it): the one of row group i mod G
    auto owns_row = [&](int i) { return LDSQ || (i % (L.Gl * grid.nwg)) / L.Gl == grid.wg; };
    double* __restrict__ S = a.S;

    QPROF_DECL;
    // ---- prologue: intercept Schur complement (lsa.py:98-104) and rescaling (lsa.py:108-109)
    double a11 = 1.0, beta0c = 0.0;
    if (a.intercept) {
        a11 = a.Sigma0[0];
        for (int j = tid; j < m; j += T) a12[j] = a.Sigma0[(int64_t)(j + 1) * a.lds0];
    }
    for (int j = tid; j < ld; j += T) {
        const double b0 = j < m ? a.b0[j + off] : 0.0;
        absb[j] = fabs(b0);
        bsgn[j] = (b0 > 0.0) ? 1.0 : ((b0 < 0.0) ? -1.0 : 0.0);
        beta[j] = 0.0;
        v[j] = 0.0;
        state[j] = j < m ? 0 : 2;
        pos[j] = m;
        if (j >= m) { a12[j] = 0.0; cvec[j] = 0.0; }
    }
    __syncthreads();
    if (a.intercept) {
        double s[1] = {0.0}, dummy[1] = {0.0};
        for (int j = tid; j < m; j += T) s[0] += a12[j] * a.b0[j + 1];
        B::reduce2(s, WaveOpSum(), dummy, WaveOpMax(), red, red_phase);
        beta0c = s[0] / a11;
    }
    for (int e = tid; e < m * ld; e += T) {
        const int i = e / ld, j = e - i * ld;
        double val = 0.0;
        if (j < m) {
            val = a.Sigma0[(int64_t)(i + off) * a.lds0 + (j + off)];
            if (a.intercept) val -= a12[i] * a12[j] / a11;
            val = absb[i] * val * absb[j];
        }
        S[e] = val;
    }
    __syncthreads();
    // Cvec = b' Sigma (lsa.py:114), rows of the symmetric S
    fused_mv<T, false>(L, grid, S, ld, nullptr, 0, m, bsgn, 1, part,
                       [&](int j, double2 t) { cvec[j] = t.x; cvec[j + 1] = j + 1 < m ? t.y : 0.0; }, [&](int, double2) {});
    if (grid.failed) { if (tid == 0 && writer) *a.n_steps = -1; return; }
    const int max_steps = a.max_steps > 0 ? a.max_steps : 8 * m;
    const double logn = log(a.n);
    double Cmax;
    {
        double rss[1] = {0.0}, cm[1] = {0.0};
        for (int j = tid; j < m; j += T) {
            if (writer) a.beta_path[j] = 0.0;
            const double c = cvec[j];
            rss[0] += bsgn[j] * c;
            cm[0] = fmax(cm[0], fabs(c));
        }
        B::reduce2(rss, WaveOpSum(), cm, WaveOpMax(), red, red_phase);
        Cmax = cm[0];
        if (tid == 0 && writer) {
            a.aic[0] = rss[0]; a.bic[0] = rss[0];
            a.beta0[0] = a.intercept ? beta0c : 0.0;
        }
    }
    QTICK(0);
    int na = 0, k = 0;
    bool had_drops = false;
    double tsq = 0.0;         // |R^{-T} sgn|^2 = sgn' Gi1 = 1/A^2, carried with the factor
    // Thread j owns variable j (T >= m): its coefficient, correlation, state and position live in registers for the whole path;
    // LDS holds what other threads read: ckey[j] = Cvec[j] while j may enter (NaN otherwise: the select's only read), pos, v.
    // A dependent LDS round trip costs ~60 ns with one wave per SIMD, which is what a step is made of.
    const int jv = min(tid, ld - 1);
    const bool own = tid < m;
    const double r_absb = absb[jv], r_bsgn = bsgn[jv], r_a12 = a.intercept ? a12[jv] : 0.0;
    double r_beta = 0.0, r_cvec = cvec[jv];
    int r_state = own ? 0 : 2, r_pos = m;
    double* ckey = beta;      // (the LDS vector `beta` is free: coefficients are in registers)
    lds_barrier();
    if (tid < ld) ckey[tid] = own ? r_cvec : __builtin_nan("");
    lds_barrier();

    int last_pos = -1;         // the position whose row of Q is in `lastq`
    // Append variable `inew` with sign `sg` at position n_at (lsa.py:12-32 on the carried rows); returns 1 if the rank grew, 0 if
    // the column is machine-singular (nothing is modified then).
    auto append = [&](int n_at, int inew, double sg, double eps_rank) -> int {
        // this thread's two entries of row `inew` of S (consumed after the pass) and the diagonal
        const double2 s2 = (tid < NQ) ? *reinterpret_cast<const double2*>(S + (int64_t)inew * ld + 2 * tid) : double2{0.0, 0.0};
        const double sdiag = S[(int64_t)inew * ld + inew];
        QTICK(1);
        // r = column `inew` of Q: read in place when Q is in LDS, gathered once from global memory otherwise
        const double* xr = Q + inew;
        int xs = ld;
        if constexpr (!LDSQ) {
            // (clusters: the row the previous append made was written by its owner after that append's barrier -- not yet visible
            // here; every workgroup kept it in LDS)
            for (int i = tid; i < n_at; i += T) rv[i] = (i == last_pos) ? lastq[inew] : Q[i * ld + inew];
            lds_barrier();
            xr = rv; xs = 1;
        }
        QTICK(2);
        // |r|^2 and r.t: every wave sums all n_at terms in the same order (no barrier; identical in every wave)
        double rr = 0.0, rt = 0.0;
        for (int i0 = 0; i0 < n_at; i0 += 128) {
            const int i = i0 + lane, i2 = i + 64;
            const double ra = xr[min(i, n_at - 1) * xs], ta = tv[min(i, n_at - 1)];
            const double rb = xr[min(i2, n_at - 1) * xs], tb = tv[min(i2, n_at - 1)];
            if (i < n_at) { rr = fma(ra, ra, rr); rt = fma(ra, ta, rt); }
            if (i2 < n_at) { rr = fma(rb, rb, rr); rt = fma(rb, tb, rt); }
        }
        rr = wave_allreduce_sum(rr);
        rt = wave_allreduce_sum(rt);
        const double d = sdiag - rr;                 // r_pp^2 (lsa.py:18)
        if (uni(n_at > 0 && d <= eps_rank)) return 0;
        const double rinv = rsqrt_newton(d);         // 1 / r_pp
        const double tn = (sg - rt) * rinv;          // new entry of R^{-T} sgn
        tsq = fma(tn, tn, tsq);
        QTICK(3);
        double* __restrict__ qrow = Q + n_at * ld;
        double* __restrict__ rrow = RT + rt_off(n_at);
        const bool mine_row = owns_row(n_at);
        fused_mv<T, LDSQ>(L, grid, Q, ld, RT, n_at + 1, n_at, xr, xs, part,
            [&](int j, double2 t) {
                double q0 = (s2.x - t.x) * rinv, q1 = (s2.y - t.y) * rinv;
                const int2 pj = *reinterpret_cast<const int2*>(pos + j);
                double2 vj = *reinterpret_cast<double2*>(v + j);
                if (pj.x < n_at) q0 = 0.0;            // a variable appended earlier: its entry of the Schur complement is zero
                if (pj.y < n_at) q1 = 0.0;
                if (mine_row) *reinterpret_cast<double2*>(qrow + j) = double2{q0, q1};
                if constexpr (!LDSQ) *reinterpret_cast<double2*>(lastq + j) = double2{q0, q1};
                vj.x = fma(q0, tn, vj.x); vj.y = fma(q1, tn, vj.y);
                *reinterpret_cast<double2*>(v + j) = vj;
            },
            [&](int i, double2 t) {
                // new column of R^{-1} (a row of RT): c = [-R^{-1} r / rpp ; 1/rpp];  Gi1 += c tn
                const double c0 = i < n_at ? -t.x * rinv : (i == n_at ? rinv : 0.0);
                const double c1 = i + 1 < n_at ? -t.y * rinv : (i + 1 == n_at ? rinv : 0.0);
                if (mine_row) *reinterpret_cast<double2*>(rrow + i) = double2{c0, c1};
                if (i < n_at) gi1[i] = fma(c0, tn, gi1[i]); else if (i == n_at) gi1[i] = c0 * tn;
                if (i + 1 < n_at) gi1[i + 1] = fma(c1, tn, gi1[i + 1]); else if (i + 1 == n_at) gi1[i + 1] = c1 * tn;
                if (i == n_at || i + 1 == n_at) { tv[n_at] = tn; sgn[n_at] = sg; act[n_at] = inew; pos[inew] = n_at; }
            });
        if (grid.failed) return -1;
        last_pos = n_at;
        return 1;
    };

    int re_i = 0, re_end = 0;      // positions [re_i, re_end) of the active list wait to be appended again (after a lasso drop)
    while (k < max_steps && na < m) {
        ++k;
        // ---- appends: first the positions a lasso drop of the previous step left to be rebuilt (that step admits no new
        // variable, lsa.py:130), else the new variables in increasing index order (lsa.py:130-149): every wave scans ckey (LDS)
        // for the first candidate and the number of candidates (ballots: the candidate mask of 64 variables is a scalar);
        // further scans only when there are ties.
        int start = 0;
        while (true) {
            int inew, n_at, left = 0;
            double sg, eps_rank;
            const bool rebuild = re_i < re_end;
            if (rebuild) {
                n_at = re_i; inew = uni(act[re_i]); sg = sgn[re_i]; eps_rank = 0.0;
            } else if (!had_drops) {
                inew = m;
                int ncand = 0;
                const double thr = Cmax - eps;
                double cnew = 0.0;
                for (int j0 = start & ~127; j0 < m; j0 += 128) {
                    const int ja = j0 + lane, jb = ja + 64;
                    const double ca = ckey[min(ja, ld - 1)], cb = ckey[min(jb, ld - 1)];      // (NaN: not a candidate)
                    const unsigned long long ma = __ballot(ja >= start && ja < m && fabs(ca) >= thr);
                    const unsigned long long mb = __ballot(jb >= start && jb < m && fabs(cb) >= thr);
                    if (inew == m && (ma | mb)) {
                        const int l = ma ? __ffsll((long long)ma) - 1 : __ffsll((long long)mb) - 1;
                        inew = (ma ? j0 : j0 + 64) + l;
                        cnew = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(ma ? ca : cb), l),
                                                __builtin_amdgcn_readlane(__double2loint(ma ? ca : cb), l));
                    }
                    ncand += __popcll(ma) + __popcll(mb);
                }
                left = ncand - 1;
                if (inew >= m) break;
                n_at = na; sg = (cnew > 0.0) ? 1.0 : ((cnew < 0.0) ? -1.0 : 0.0); eps_rank = eps;
            } else {
                break;
            }
            const int grew = append(n_at, inew, sg, eps_rank);
            if (grew < 0) { if (tid == 0 && writer) *a.n_steps = -1; return; }      // a cluster barrier gave up: the host reruns on one workgroup
            if (tid == inew) r_pos = grew ? n_at : m;
            if (rebuild) { ++re_i; continue; }
            if (tid == inew) { r_state = grew ? 1 : 2; ckey[inew] = __builtin_nan(""); }      // 2: machine-singular, ignored (lsa.py:139-144)
            if (grew) ++na;
            start = inew + 1;
            if (left <= 0) break;
            lds_barrier();
        }
        if (na == 0) break;   // nothing could enter (degenerate input)
        QTICK(8);
        // ---- equiangular direction w = A Gi1 (lsa.py:151-153), u = Sigma[:,active] w = A v; step length (lsa.py:154-162) and
        // lasso modification (lsa.py:164-173)
        const double A = rsqrt_newton(tsq);
        double gamhat = Cmax * (tsq * A);              // Cmax / A
        double mins[2] = {INFINITY, INFINITY};
        const double uj = A * v[jv];
        const double wj = A * gi1[min(r_pos, ld - 1)];
        if (r_state == 0) {
            const double g1 = (Cmax - r_cvec) / (A - uj);
            const double g2 = (Cmax + r_cvec) / (A + uj);
            if (g1 > eps) mins[0] = fmin(mins[0], g1);
            if (g2 > eps) mins[0] = fmin(mins[0], g2);
        } else if (r_state == 1 && a.type == 1) {
            const double z = -r_beta / wj;
            if (z > eps) mins[1] = fmin(mins[1], z);
        }
        QTICK(9);
        {
            mins[0] = wave_allreduce_min(mins[0]);
            if (a.type == 1) mins[1] = wave_allreduce_min(mins[1]);      // (uniform: the lasso's crossing distance)
            const int ops[2] = {1, 1};
            cross_wave<T>(mins, ops, red, red_phase);
        }
        QTICK(10);
        gamhat = fmin(mins[0], gamhat);
        had_drops = uni(a.type == 1 && mins[1] < gamhat);
        if (had_drops) gamhat = mins[1];
        // ---- move (lsa.py:175-177), drops (lsa.py:179-186), the path point: un-scaled beta (lsa.py:194-201), RSS, dof, AIC/BIC
        // (:190-210) and Cmax of the next step (lsa.py:128-129): every thread on its own variable
        double rec[4] = {0.0, 0.0, 0.0, 0.0};      // RSS, dof, a12 . beta, max |Cvec| over the variables that are not active
        bool nonzero = false;
        if (own) {
            if (r_state == 1) {
                bool dropped = false;
                if (had_drops) dropped = (-r_beta / wj) == gamhat;
                r_beta = dropped ? 0.0 : r_beta + gamhat * wj;
                dropf[r_pos] = dropped ? 1 : 0;
                if (dropped) { r_state = 0; pos[tid] = m; r_pos = m; }
            }
            r_cvec -= gamhat * uj;
            ckey[tid] = r_state == 0 ? r_cvec : __builtin_nan("");
            const double ub = r_absb * r_beta;
            if (writer) a.beta_path[(int64_t)k * m + tid] = ub;
            rec[0] = (r_bsgn - r_beta) * r_cvec;
            nonzero = fabs(ub) > eps;
            if (a.intercept) rec[2] = r_a12 * ub;
            if (r_state != 1) rec[3] = fabs(r_cvec);
        }
        QTICK(11);
        {
            rec[0] = wave_allreduce_sum(rec[0]);
            rec[1] = (double)__popcll(__ballot(nonzero));                  // dof: a count, no butterfly
            if (a.intercept) rec[2] = wave_allreduce_sum(rec[2]);          // (uniform)
            rec[3] = wave_allreduce_max(rec[3]);
            const int ops[4] = {0, 0, 0, 2};
            cross_wave<T>(rec, ops, red, red_phase);
        }
        QTICK(12);
        Cmax = rec[3];
        if (tid == T - 64 && writer) {      // (a wave that owns no variable: off the path of the waves that do)
            a.aic[k] = rec[0] + 2.0 * rec[1];
            a.bic[k] = rec[0] + logn * rec[1];
            a.beta0[k] = a.intercept ? beta0c - rec[2] / a11 : 0.0;
        }
        if (had_drops) {
            if (tid == 0) {
                int q = 0, first = na;
                for (int i = 0; i < na; ++i) {
                    if (!dropf[i]) { act[q] = act[i]; sgn[q] = sgn[i]; ++q; }
                    else if (first == na) first = i;
                }
                sh_i[1] = q; sh_i[3] = first;
            }
            lds_barrier();
            const int keep = uni(sh_i[1]), first = uni(sh_i[3]);
            for (int i = first + tid; i < keep; i += T) pos[act[i]] = m;      // appended again at the top of the next step
            // The rows before the first dropped position are unchanged; t there too.  v and Gi1 lose the later rows' terms:
            // v = Q[:first]' t, Gi1 = R^{-1}[:first,:first] t.
            double ts = 0.0;
            for (int i = lane; i < first; i += 64) ts = fma(tv[i], tv[i], ts);
            tsq = wave_allreduce_sum(ts);
            fused_mv<T, LDSQ>(L, grid, Q, ld, RT, first, first, tv, 1, part,
                [&](int j, double2 t) { *reinterpret_cast<double2*>(v + j) = t; },
                [&](int i, double2 t) { gi1[i] = t.x; if (i + 1 < first) gi1[i + 1] = t.y; });
            if (grid.failed) { if (tid == 0 && writer) *a.n_steps = -1; return; }
            last_pos = -1;                 // (the barrier of this pass published every row written so far)
            re_i = first; re_end = keep;
            na = keep;
        }
        QTICK(13);
    }
#ifdef DLSA_LARS_PROF
    if (tid == 0) {
        printf("LARS_Q_PROF shader clock %.0f MHz over the kernel\n", (double)(clock64() - q_prof_c0) / ((double)(wall_clock64() - q_prof_w0) * 0.01));
        printf("LARS_Q_PROF T=%d lds=%d p=%d steps=%d us: prologue %.0f | select %.0f gather %.0f sums %.0f | pass: loop %.0f bar %.0f emit %.0f bar %.0f | "
               "post-append %.0f mins %.0f reduce %.0f move+record %.0f reduce %.0f tail %.0f\n", T, (int)LDSQ, p, k, q_prof_t[0] * 0.01, q_prof_t[1] * 0.01,
               q_prof_t[2] * 0.01, q_prof_t[3] * 0.01, q_prof_t[4] * 0.01, q_prof_t[5] * 0.01, q_prof_t[6] * 0.01, q_prof_t[7] * 0.01, q_prof_t[8] * 0.01,
               q_prof_t[9] * 0.01, q_prof_t[10] * 0.01, q_prof_t[11] * 0.01, q_prof_t[12] * 0.01, q_prof_t[13] * 0.01);
    }
#endif
    if (tid == 0 && writer) *a.n_steps = k;
}

template <int T, bool LDSQ>
int launch_q(LarsArgs& a, int m, hipStream_t s, int nwg = 1) {
    const size_t shm = lars_q_lds_doubles(m, T, LDSQ) * sizeof(double);
    if (shm > 48 * 1024)
        DLSA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(lars_q_kernel<T, LDSQ>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
    a.nwg = LDSQ ? 1 : nwg;
    if (a.nwg > 1) DLSA_HIP_CHECK(hipMemsetAsync(a.bar, 0, 256, s));
    // several workgroups meet at a grid barrier per append: the plain launch (bounded barrier, single-workgroup rerun), or -- asked for
    // through dlsa_kernel_options.cooperative -- a cooperative one
    void* kargs[] = {(void*)&a};
    if (a.nwg == 1 || launch_cooperative(reinterpret_cast<const void*>(lars_q_kernel<T, LDSQ>), dim3(a.nwg), dim3(T), kargs, shm, s) != hipSuccess)
        hipLaunchKernelGGL((lars_q_kernel<T, LDSQ>), dim3(a.nwg), dim3(T), shm, s, a);
    DLSA_HIP_CHECK(hipGetLastError());
    return DLSA_OK;
}

constexpr size_t LARS_Q_STATIC_LDS = 1024;      // red, sh_i and alignment

}  // namespace

// DLSA_LARS_Q=0 keeps lars.hip's kernels for every width (A/B runs, tests of both forms)
bool lars_q_eligible(int p, int intercept) {
    const int m = p - (intercept ? 1 : 0);
    if (m < 1 || m > LARS_Q_MAX_M) return false;
    if (const char* e = kernel_knob("DLSA_LARS_Q")) return atoi(e) != 0;
    return true;
}

// Workgroups that share the fused pass when Q and RT are in global memory: the pass is bound by what ONE CU streams from L2, so the
// row groups are dealt over nwg CUs at the price of one grid barrier per append (~1.4 us) and the exchange of the partial sums
// (bench/lars_ab.py).  DLSA_LARS_Q_WGS overrides (1 .. 8).
static int lars_q_workgroups(int m) {
    int nwg = m <= 200 ? 1 : (m <= 420 ? 4 : 8);
    if (const char* e = kernel_knob("DLSA_LARS_Q_WGS")) nwg = atoi(e);
    return std::max(1, std::min(nwg, 8));
}

// DLSA_LARS_Q_THREADS: 256 | 512 | 1024 forces the workgroup size; DLSA_LARS_Q_LDS=0 keeps Q and RT in global memory
int lars_q_run(LarsArgs& a, int p, int intercept, hipStream_t s, int max_wgs, int* wgs_used) {
    const int m = p - (intercept ? 1 : 0);
    int threads = 0;
    if (const char* e = kernel_knob("DLSA_LARS_Q_THREADS")) threads = atoi(e);
    bool want_lds = true;
    if (const char* e = kernel_knob("DLSA_LARS_Q_LDS")) want_lds = atoi(e) != 0;
    if (wgs_used) *wgs_used = 1;
    auto fits = [&](int T) { return lars_q_lds_doubles(m, T, true) * 8 + LARS_Q_STATIC_LDS <= (size_t)kLdsBytes && (m + 2) <= T; };
    if (want_lds) {
        const int T = threads == 512 ? 512 : 256;
        if (fits(T)) return T == 512 ? launch_q<512, true>(a, m, s) : launch_q<256, true>(a, m, s);
        if (threads == 0 && fits(256)) return launch_q<256, true>(a, m, s);
    }
    const int nwg = std::min(lars_q_workgroups(m), std::max(1, max_wgs));
    if (wgs_used) *wgs_used = nwg;
    if (threads == 1024 || m + 2 > 512) return launch_q<1024, false>(a, m, s, nwg);
    return launch_q<512, false>(a, m, s, nwg);      // (p = 260: 2.47 ms with 512 threads, 2.56 with 1024)
}

}  // namespace dlsa
