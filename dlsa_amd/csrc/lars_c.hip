// LARS / lasso path of the least-squares approximation for WIDE problems (LARS_C_MIN_M < m = p - intercept <= LARS_C_MAX_M): the
// carried Cholesky rows of lars_q.hip with the fused pass split by COLUMNS over 16 .. 64 workgroups (round 6; reference: lars_lsa and
// updateR, dlsa/lsa.py:12-32, 90-212).
//
// The method is lars_q.hip's: row i of Q = R^{-T} Sigma[active, :] is written when position i is appended,
//     Q[na][j] = (Sigma[new][j] - sum_{i<na} Q[i][new] Q[i][j]) / r_pp,     r_pp^2 = Sigma[new][new] - |Q[:, new]|^2   (lsa.py:17-19)
// so r = R^{-T} x of the next append is a column of Q, Sigma[:, active] w = A Q't is carried as v += Q[na] t_na, and the new column
// of R^{-1} (a row of RT = (R^{-1})') comes out of the same pass over [Q | RT].  At m = 2000 that pass reads up to 64 MB per step:
// one CU streams ~75 GB/s from L2, so the pass wants tens of CUs -- and lars_q.hip's way of sharing it (row groups dealt over the
// workgroups, every workgroup adding all partial rows) grows with the square of their number, as did lars.hip's reduce of the
// partial u vectors (20 of its 75 ms at p = 2000, profiles/r05_lars_grid_phases.txt).  Here a workgroup owns COLUMNS: blocks of 16
// columns (one 128-byte line per row) of Q and of RT are dealt cyclically, a workgroup sums ITS columns over ALL rows -- its threads
// are (column pair) x (row group), the row groups' partial sums meet in LDS in a fixed order -- and writes its entries of the new
// rows of Q and RT, which are final.  What the others need of them goes through ONE bounded grid barrier per append: afterwards
// every workgroup reads the two new rows (<= 32 KB, L2) and carries v, Gi1, t for all variables itself; the column of Q that is the
// next append's r is gathered from global memory (rows published by earlier barriers).  Everything else -- selection, step length,
// lasso crossings, the path record -- is replicated: thread t of every workgroup owns variables 4t .. 4t + 3 (coefficient,
// correlation, v, state and position in registers), the same numbers in the same order in every workgroup, so the decisions agree
// without communication.  Workgroup 0 writes the path.  A lasso drop truncates to the positions before the first dropped one,
// recomputes v and Gi1 with the same pass (results through a scratch row) and appends the kept positions again, as lars_q.hip.
// A barrier wait that times out aborts the launch; the host reruns the path on lars.hip's single-workgroup kernel.
#include "common.h"
#include "lars.h"
#include "options.h"
#include <math.h>
#include <stdlib.h>
#include <algorithm>

namespace dlsa {
namespace {

constexpr int CT = 512;                  // threads per workgroup (eight waves: every phase of a step ends in a workgroup barrier or a block
constexpr int VPT = 4;                   // reduction, which cost by the number of waves) with VPT adjacent variables each, in registers
constexpr int CWAVES = CT / 64;
static_assert(CT * VPT >= LARS_C_MAX_M + 4 && VPT % 2 == 0, "every variable has a thread");

#ifdef DLSA_LARS_PROF
__shared__ long long c_prof_t[16];
__shared__ long long c_prof_last;
#define CPROF_DECL do { if (threadIdx.x == 0) { for (int q_ = 0; q_ < 16; ++q_) c_prof_t[q_] = 0; c_prof_last = wall_clock64(); } } while (0)
#define CTICK(i) do { if (threadIdx.x == 0) { const long long now_ = wall_clock64(); c_prof_t[i] += now_ - c_prof_last; c_prof_last = now_; } } while (0)
#else
#define CPROF_DECL
#define CTICK(i)
#endif

// workgroup barrier for data exchanged through LDS (no vmcnt drain: see lars_q.hip)
__device__ __forceinline__ void c_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ int c_uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ bool c_uni(bool b) { return __builtin_amdgcn_readfirstlane((int)b) != 0; }

// lars.hip's bounded grid barrier: a monotonic counter with agent-scope release / acquire, relaxed polling with one acquire after
// the match; a wait longer than the timeout sets the abort word and every workgroup leaves.
struct CGrid {
    int nwg, wg;
    unsigned* bar;
    long long timeout;
    unsigned phase;
};
__device__ __forceinline__ bool c_grid_barrier(CGrid& gr) {
    __shared__ int gb_ok;
    __syncthreads();                 // this workgroup's global stores have completed in every wave
    ++gr.phase;
    if (threadIdx.x == 0) {
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
    }
    __syncthreads();
    return gb_ok != 0;
}

// block-wide reduction of N values (ops: 0 sum, 1 min, 2 max); one barrier, two buffers of 4 * CWAVES doubles used alternately
template <int N>
__device__ __forceinline__ void c_reduce(double (&v)[N], const int (&ops)[N], double* red, int& phase) {
    static_assert(N <= 4, "a buffer holds four values per wave");
#pragma unroll
    for (int q = 0; q < N; ++q)
        v[q] = ops[q] == 0 ? wave_allreduce_sum(v[q]) : (ops[q] == 1 ? wave_allreduce_min(v[q]) : wave_allreduce_max(v[q]));
    double* r = red + (phase & 1) * 4 * CWAVES;
    ++phase;
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int q = 0; q < N; ++q) r[q * CWAVES + (threadIdx.x >> 6)] = v[q];
    }
    c_lds_barrier();
#pragma unroll
    for (int q = 0; q < N; ++q) {
        double s = r[q * CWAVES];
        for (int k = 1; k < CWAVES; ++k) {
            const double u = r[q * CWAVES + k];
            s = ops[q] == 0 ? s + u : (ops[q] == 1 ? fmin(s, u) : fmax(s, u));
        }
        v[q] = s;
    }
}

__device__ __forceinline__ double c_rsqrt_newton(double d) {      // 1 / sqrt(d) to about an ulp (lars_q.hip)
    double y = __builtin_amdgcn_rsq(d);
    double e = fma(-d * y, y, 1.0);
    y = fma(0.5 * y, e, y);
    e = fma(-d * y, y, 1.0);
    return fma(0.5 * y, e, y);
}

// A thread's place in the column-split pass, fixed for the path: the workgroup owns the 16-column blocks b = wg, wg + nwg, ... of Q
// and of RT (NBW = blocks per workgroup and matrix); its CP = 16 NBW column-pair slots (8 NBW of Q, then 8 NBW of RT) are walked by
// RG = CT / CP row groups.  A wave's 64 threads cover whole 128-byte lines of 64 / CP consecutive row groups.
struct CLayout {
    int CP, RG, rg, c2;
    bool isq, live;
};
__device__ __forceinline__ CLayout c_layout(int tid, int ld, int nwg, int wg) {
    CLayout L;
    const int nblk = (ld + 15) >> 4, NBW = (nblk + nwg - 1) / nwg;
    L.CP = 16 * NBW;
    L.RG = CT / L.CP;
    const int slot = tid % L.CP;
    L.rg = tid / L.CP;
    L.isq = slot < 8 * NBW;
    const int s = L.isq ? slot : slot - 8 * NBW;
    const int b = wg + (s >> 3) * nwg;
    L.c2 = 16 * b + 2 * (s & 7);
    L.live = L.rg < L.RG && b < nblk && L.c2 < ld;
    return L;
}

// This workgroup's columns of  M' x  over the first n rows of Qm (row stride ld) and of RT (row i holds columns 0 .. i; ncolR columns
// are produced): eq(c2, {sum_c2, sum_c2+1}) / er(c2, {..}) run on the row-group-0 thread of every owned column pair after the row
// groups' partial sums have met in LDS.  One LDS barrier.  x in LDS.
template <typename EQ, typename ER>
__device__ __forceinline__ void c_col_mv(const CLayout& L, const double* __restrict__ Qm, const double* __restrict__ RT, int ld, int ncolR, int n,
                                         const double* __restrict__ x, double2* part, EQ&& eq, ER&& er) {
    const bool mine = L.live && (L.isq ? Qm != nullptr : L.c2 < ncolR);
    if (mine) {
        double2 a0 = {0.0, 0.0}, a1 = a0, a2 = a0, a3 = a0;
#ifndef DLSA_LARS_C_NB
#define DLSA_LARS_C_NB 8
#endif
        constexpr int NB = DLSA_LARS_C_NB;       // 16-byte row loads in flight per thread (8: 64 KB per workgroup)
        const int RG = L.RG;
        int i = L.rg;
        if (!L.isq && L.c2 > i) i += (L.c2 - i + RG - 1) / RG * RG;      // column pair (c2, c2 + 1) of RT exists from row c2 on
        const double* __restrict__ rp = (L.isq ? Qm : RT) + (int64_t)i * ld + L.c2;
        const double* __restrict__ xp = x + i;
        const int64_t step = (int64_t)RG * ld;
        while (i + (NB - 1) * RG < n) {
            double2 q[NB];
            double xv[NB];
#pragma unroll
            for (int k = 0; k < NB; ++k) {
                q[k] = *reinterpret_cast<const double2*>(rp);
                xv[k] = *xp;
                rp += step; xp += RG; i += RG;
            }
#pragma unroll
            for (int k = 0; k < NB; k += 4) {
                a0.x = fma(xv[k], q[k].x, a0.x); a0.y = fma(xv[k], q[k].y, a0.y);
                a1.x = fma(xv[k + 1], q[k + 1].x, a1.x); a1.y = fma(xv[k + 1], q[k + 1].y, a1.y);
                a2.x = fma(xv[k + 2], q[k + 2].x, a2.x); a2.y = fma(xv[k + 2], q[k + 2].y, a2.y);
                a3.x = fma(xv[k + 3], q[k + 3].x, a3.x); a3.y = fma(xv[k + 3], q[k + 3].y, a3.y);
            }
        }
        if (i < n) {      // the last, partial batch: the same independent loads under a row mask
            double2 q[NB];
            double xv[NB];
#pragma unroll
            for (int k = 0; k < NB; ++k) {
                q[k] = double2{0.0, 0.0};
                xv[k] = 0.0;
                if (i < n) {
                    q[k] = *reinterpret_cast<const double2*>(rp);
                    xv[k] = *xp;
                    rp += step; xp += RG; i += RG;
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
    c_lds_barrier();
    if (mine && L.rg == 0) {
        double2 t = part[threadIdx.x];
        for (int q = 1; q < L.RG; ++q) { const double2 u = part[threadIdx.x + q * L.CP]; t.x += u.x; t.y += u.y; }
        if (L.isq) eq(L.c2, t); else er(L.c2, t);
    }
}

__host__ __device__ inline size_t lars_c_lds_bytes(int m) {
    const size_t ld = (size_t)((m + 1) & ~1);
    return (2 * (size_t)CT + 5 * ld) * 8 + 3 * ld * 4;           // part | ckey, tv, gi1, rv, sgn | pos, act, dropf
}

__global__ __launch_bounds__(CT) void lars_c_kernel(LarsArgs a) {
    __shared__ double red[2 * 4 * CWAVES];
    __shared__ int sh_i[4];
    int red_phase = 0;
    extern __shared__ __attribute__((aligned(16))) double dyn[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int p = a.p;
    const int off = a.intercept ? 1 : 0;
    const int m = p - off;
    const int ld = (m + 1) & ~1;
    const double eps = a.eps;
    double2* part = reinterpret_cast<double2*>(dyn);
    double* ckey = dyn + 2 * CT;         // Cvec[j] while variable j may enter, NaN otherwise (the select's only read)
    double* tv = ckey + ld;              // R^{-T} sgn, by active position
    double* gi1 = tv + ld;               // R^{-1} R^{-T} sgn, by active position
    double* rv = gi1 + ld;               // x of a pass: column `new` of Q
    double* sgn = rv + ld;               // sign of the correlation at entry, by active position
    int* pos = reinterpret_cast<int*>(sgn + ld);   // position in the active list by variable, m when not active
    int* act = pos + ld;                 // active list (variable ids)
    int* dropf = act + ld;               // by active position
    double* __restrict__ S = a.S;
    double* __restrict__ Q = a.RinvT;    // m x ld
    double* __restrict__ RT = a.Rinv;    // m x ld, row i holds columns 0 .. i (+ a zero pad)
    CGrid grid;
    grid.nwg = max(1, a.nwg);
    grid.wg = (int)blockIdx.x;
    grid.bar = a.bar;
    grid.timeout = a.bar_timeout;
    grid.phase = 0;
    const int G = grid.nwg, g = grid.wg;
    const bool writer = g == 0;
    const CLayout L = c_layout(tid, ld, G, g);
    int scratch_passes = 0;              // results that are not rows of Q / RT go through a.upart: two buffers of 2 ld doubles, alternately
#define LARS_C_ABORT() do { if (tid == 0 && writer) *a.n_steps = -1; return; } while (0)
#define LARS_C_SYNC() do { if (!c_grid_barrier(grid)) LARS_C_ABORT(); } while (0)

    CPROF_DECL;
    // thread t owns variables VPT t .. VPT t + VPT - 1 (registers for the whole path)
    const int j0 = VPT * tid;
    bool own[VPT];
    double r_absb[VPT], r_bsgn[VPT], r_a12[VPT], r_beta[VPT], r_cvec[VPT], r_v[VPT];
    int r_state[VPT], r_pos[VPT];
#pragma unroll
    for (int e = 0; e < VPT; ++e) { own[e] = j0 + e < m; r_a12[e] = 0.0; r_beta[e] = 0.0; r_cvec[e] = 0.0; r_v[e] = 0.0; r_pos[e] = m; }
    // ---- prologue: intercept Schur complement (lsa.py:98-104) and rescaling (lsa.py:108-109).  |b0| and a12 are staged in LDS (tv,
    // gi1) for the build of S
    double a11 = 1.0, beta0c = 0.0;
    if (a.intercept) a11 = a.Sigma0[0];
#pragma unroll
    for (int e = 0; e < VPT; ++e) {
        const int j = j0 + e;
        const double b0 = own[e] ? a.b0[j + off] : 0.0;
        r_absb[e] = fabs(b0);
        r_bsgn[e] = (b0 > 0.0) ? 1.0 : ((b0 < 0.0) ? -1.0 : 0.0);
        if (a.intercept && own[e]) r_a12[e] = a.Sigma0[(int64_t)(j + 1) * a.lds0];
        r_state[e] = own[e] ? 0 : 2;
        if (j < ld) { tv[j] = r_absb[e]; gi1[j] = r_a12[e]; rv[j] = r_bsgn[e]; pos[j] = m; }
    }
    __syncthreads();
    if (a.intercept) {
        double s[1] = {0.0};
        const int ops[1] = {0};
#pragma unroll
        for (int e = 0; e < VPT; ++e) if (own[e]) s[0] += r_a12[e] * a.b0[j0 + e + 1];
        c_reduce(s, ops, red, red_phase);
        beta0c = s[0] / a11;
    }
    for (int i = g; i < m; i += G) {
        const double ai = tv[i], a12i = gi1[i];
        for (int j = tid; j < ld; j += CT) {
            double val = 0.0;
            if (j < m) {
                val = a.Sigma0[(int64_t)(i + off) * a.lds0 + (j + off)];
                if (a.intercept) val -= a12i * gi1[j] / a11;
                val = ai * val * tv[j];
            }
            S[(int64_t)i * ld + j] = val;
        }
    }
    LARS_C_SYNC();
    // Cvec = b' Sigma (lsa.py:114): this workgroup's columns of S' sign(b0), the others' through the scratch row
    {
        double* scr = a.upart + (size_t)(scratch_passes++ & 1) * 2 * ld;
        c_col_mv(L, S, nullptr, ld, 0, m, rv, part, [&](int j, double2 t) { *reinterpret_cast<double2*>(scr + j) = t; }, [&](int, double2) {});
        LARS_C_SYNC();
#pragma unroll
        for (int h = 0; h < VPT; h += 2)
            if (j0 + h < ld) {
                const double2 t = *reinterpret_cast<const double2*>(scr + j0 + h);
                r_cvec[h] = own[h] ? t.x : 0.0; r_cvec[h + 1] = own[h + 1] ? t.y : 0.0;
            }
    }
    const int max_steps = a.max_steps > 0 ? a.max_steps : 8 * m;
    const double logn = log(a.n);
    double Cmax;
    {
        double v2[2] = {0.0, 0.0};
        const int ops[2] = {0, 2};
#pragma unroll
        for (int e = 0; e < VPT; ++e)
            if (own[e]) {
                if (writer) a.beta_path[j0 + e] = 0.0;
                v2[0] += r_bsgn[e] * r_cvec[e];
                v2[1] = fmax(v2[1], fabs(r_cvec[e]));
            }
        c_reduce(v2, ops, red, red_phase);
        Cmax = v2[1];
        if (tid == 0 && writer) {
            a.aic[0] = v2[0]; a.bic[0] = v2[0];
            a.beta0[0] = a.intercept ? beta0c : 0.0;
        }
    }
    c_lds_barrier();                      // (the staged |b0|, a12, sign(b0) have been read by everyone)
#pragma unroll
    for (int e = 0; e < VPT; ++e) if (j0 + e < ld) ckey[j0 + e] = own[e] ? r_cvec[e] : __builtin_nan("");
    c_lds_barrier();
    CTICK(0);

    int na = 0, k = 0;
    bool had_drops = false;
    double tsq = 0.0;         // |R^{-T} sgn|^2 = 1 / A^2, carried with the factor

    // Append variable `inew` with sign `sg` at position n_at (lsa.py:12-32 on the carried rows); 1 if the rank grew, 0 if the column
    // is machine-singular (nothing is modified then), -1 when the grid barrier gave up.
    auto append = [&](int n_at, int inew, double sg, double eps_rank) -> int {
        // the owner's two entries of row `inew` of S (consumed after the pass) and the diagonal
        const bool emit_q = L.live && L.isq && L.rg == 0;
        const double2 s2 = emit_q ? *reinterpret_cast<const double2*>(S + (int64_t)inew * ld + L.c2) : double2{0.0, 0.0};
        const double sdiag = S[(int64_t)inew * ld + inew];
        // r = column `inew` of Q, gathered from global memory (every row was published by the barrier of its own append);
        // |r|^2 and r.t by a block reduction whose barrier also publishes r
        double sums[2] = {0.0, 0.0};
        for (int i = tid; i < n_at; i += CT) {
            const double ri = Q[(int64_t)i * ld + inew];
            rv[i] = ri;
            sums[0] = fma(ri, ri, sums[0]);
            sums[1] = fma(ri, tv[i], sums[1]);
        }
        CTICK(1);
        {
            const int ops[2] = {0, 0};
            c_reduce(sums, ops, red, red_phase);
        }
        const double d = sdiag - sums[0];            // r_pp^2 (lsa.py:18)
        if (c_uni(n_at > 0 && d <= eps_rank)) return 0;
        const double rinv = c_rsqrt_newton(d);       // 1 / r_pp
        const double tn = (sg - sums[1]) * rinv;     // new entry of R^{-T} sgn
        tsq = fma(tn, tn, tsq);
        CTICK(2);
        double* __restrict__ qrow = Q + (int64_t)n_at * ld;
        double* __restrict__ rrow = RT + (int64_t)n_at * ld;
        c_col_mv(L, Q, RT, ld, n_at + 1, n_at, rv, part,
            [&](int j, double2 t) {
                double q0 = (s2.x - t.x) * rinv, q1 = (s2.y - t.y) * rinv;
                const int2 pj = *reinterpret_cast<const int2*>(pos + j);
                if (pj.x < n_at) q0 = 0.0;            // a variable appended earlier: its entry of the Schur complement is zero
                if (pj.y < n_at) q1 = 0.0;
                *reinterpret_cast<double2*>(qrow + j) = double2{q0, q1};
            },
            [&](int i, double2 t) {
                // new column of R^{-1} (a row of RT): c = [-R^{-1} r / rpp ; 1 / rpp], zero beyond the diagonal
                const double c0 = i < n_at ? -t.x * rinv : (i == n_at ? rinv : 0.0);
                const double c1 = i + 1 < n_at ? -t.y * rinv : (i + 1 == n_at ? rinv : 0.0);
                *reinterpret_cast<double2*>(rrow + i) = double2{c0, c1};
            });
        CTICK(3);
        if (!c_grid_barrier(grid)) return -1;
        CTICK(4);
        // every workgroup carries v, Gi1 and t for ALL variables: the two new rows, read back whole
#pragma unroll
        for (int h = 0; h < VPT; h += 2) {
            const int jh = j0 + h;
            if (jh < ld) {
                const double2 q = *reinterpret_cast<const double2*>(qrow + jh);
                r_v[h] = fma(q.x, tn, r_v[h]); r_v[h + 1] = fma(q.y, tn, r_v[h + 1]);
            }
            if (jh <= n_at) {
                const double2 c = *reinterpret_cast<const double2*>(rrow + jh);
                gi1[jh] = jh < n_at ? fma(c.x, tn, gi1[jh]) : c.x * tn;
                if (jh + 1 <= n_at) gi1[jh + 1] = jh + 1 < n_at ? fma(c.y, tn, gi1[jh + 1]) : c.y * tn;
            }
        }
        if (tid == 0) { tv[n_at] = tn; sgn[n_at] = sg; act[n_at] = inew; pos[inew] = n_at; }
        c_lds_barrier();
        CTICK(5);
        return 1;
    };

    int re_i = 0, re_end = 0;      // positions [re_i, re_end) of the active list wait to be appended again (after a lasso drop)
    while (k < max_steps && na < m) {
        ++k;
        // ---- appends: first the positions a lasso drop of the previous step left to be rebuilt (that step admits no new variable,
        // lsa.py:130), else the new variables in increasing index order (lsa.py:130-149): every wave scans ckey for the first
        // candidate and the number of candidates (ballots); further scans only when there are ties
        int start = 0;
        while (true) {
            int inew, n_at, left = 0;
            double sg, eps_rank;
            const bool rebuild = re_i < re_end;
            if (rebuild) {
                n_at = re_i; inew = c_uni(act[re_i]); sg = sgn[re_i]; eps_rank = 0.0;
            } else if (!had_drops) {
                inew = m;
                int ncand = 0;
                const double thr = Cmax - eps;
                double cnew = 0.0;
                for (int jb0 = start & ~127; jb0 < m; jb0 += 128) {
                    const int ja = jb0 + lane, jb = ja + 64;
                    const double ca = ckey[min(ja, ld - 1)], cb = ckey[min(jb, ld - 1)];      // (NaN: not a candidate)
                    const unsigned long long ma = __ballot(ja >= start && ja < m && fabs(ca) >= thr);
                    const unsigned long long mb = __ballot(jb >= start && jb < m && fabs(cb) >= thr);
                    if (inew == m && (ma | mb)) {
                        const int l = ma ? __ffsll((long long)ma) - 1 : __ffsll((long long)mb) - 1;
                        inew = (ma ? jb0 : jb0 + 64) + l;
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
            if (grew < 0) LARS_C_ABORT();            // a grid barrier gave up: the host reruns on one workgroup
            const bool owner_new = tid == inew / VPT;
#pragma unroll
            for (int e = 0; e < VPT; ++e) if (owner_new && e == inew % VPT) r_pos[e] = grew ? n_at : m;
            if (rebuild) { ++re_i; continue; }
#pragma unroll
            for (int e = 0; e < VPT; ++e) if (owner_new && e == inew % VPT) r_state[e] = grew ? 1 : 2;      // 2: machine-singular, ignored (lsa.py:139-144)
            if (owner_new) ckey[inew] = __builtin_nan("");
            if (grew) ++na;
            start = inew + 1;
            if (left <= 0) break;
            c_lds_barrier();
        }
        if (na == 0) break;   // nothing could enter (degenerate input)
        CTICK(6);
        // ---- equiangular direction w = A Gi1 (lsa.py:151-153), u = Sigma[:, active] w = A v; step length (lsa.py:154-162) and lasso
        // modification (lsa.py:164-173)
        const double A = c_rsqrt_newton(tsq);
        double gamhat = Cmax * (tsq * A);              // Cmax / A
        double mins[2] = {INFINITY, INFINITY};
        double uj[VPT], wj[VPT];
#pragma unroll
        for (int e = 0; e < VPT; ++e) {
            uj[e] = A * r_v[e];
            wj[e] = A * gi1[min(r_pos[e], ld - 1)];
            if (r_state[e] == 0) {
                const double g1 = (Cmax - r_cvec[e]) / (A - uj[e]);
                const double g2 = (Cmax + r_cvec[e]) / (A + uj[e]);
                if (g1 > eps) mins[0] = fmin(mins[0], g1);
                if (g2 > eps) mins[0] = fmin(mins[0], g2);
            } else if (r_state[e] == 1 && a.type == 1) {
                const double z = -r_beta[e] / wj[e];
                if (z > eps) mins[1] = fmin(mins[1], z);
            }
        }
        {
            const int ops[2] = {1, 1};
            c_reduce(mins, ops, red, red_phase);
        }
        gamhat = fmin(mins[0], gamhat);
        had_drops = c_uni(a.type == 1 && mins[1] < gamhat);
        if (had_drops) gamhat = mins[1];
        CTICK(7);
        // ---- move (lsa.py:175-177), drops (lsa.py:179-186), the path point: un-scaled beta (lsa.py:194-201), RSS, dof, AIC/BIC
        // (:190-210) and Cmax of the next step (lsa.py:128-129): every thread on its own two variables
        double rec[4] = {0.0, 0.0, 0.0, 0.0};      // RSS, dof, a12 . beta, max |Cvec| over the variables that are not active
        int nnz = 0;
#pragma unroll
        for (int e = 0; e < VPT; ++e) {
            if (!own[e]) continue;
            const int j = j0 + e;
            if (r_state[e] == 1) {
                bool dropped = false;
                if (had_drops) dropped = (-r_beta[e] / wj[e]) == gamhat;
                r_beta[e] = dropped ? 0.0 : r_beta[e] + gamhat * wj[e];
                dropf[r_pos[e]] = dropped ? 1 : 0;
                if (dropped) { r_state[e] = 0; pos[j] = m; r_pos[e] = m; }
            }
            r_cvec[e] -= gamhat * uj[e];
            ckey[j] = r_state[e] == 0 ? r_cvec[e] : __builtin_nan("");
            const double ub = r_absb[e] * r_beta[e];
            if (writer) a.beta_path[(int64_t)k * m + j] = ub;
            rec[0] += (r_bsgn[e] - r_beta[e]) * r_cvec[e];
            nnz += fabs(ub) > eps ? 1 : 0;
            if (a.intercept) rec[2] += r_a12[e] * ub;
            if (r_state[e] != 1) rec[3] = fmax(rec[3], fabs(r_cvec[e]));
        }
        rec[1] = (double)nnz;      // (small integers: the sum is exact in any order)
        {
            const int ops[4] = {0, 0, 0, 2};
            c_reduce(rec, ops, red, red_phase);
        }
        Cmax = rec[3];
        if (tid == CT - 1 && writer) {
            a.aic[k] = rec[0] + 2.0 * rec[1];
            a.bic[k] = rec[0] + logn * rec[1];
            a.beta0[k] = a.intercept ? beta0c - rec[2] / a11 : 0.0;
        }
        CTICK(8);
        if (had_drops) {
            if (tid == 0) {
                int q = 0, first = na;
                for (int i = 0; i < na; ++i) {
                    if (!dropf[i]) { act[q] = act[i]; sgn[q] = sgn[i]; ++q; }
                    else if (first == na) first = i;
                }
                sh_i[1] = q; sh_i[3] = first;
            }
            c_lds_barrier();
            const int keep = c_uni(sh_i[1]), first = c_uni(sh_i[3]);
            for (int i = first + tid; i < keep; i += CT) pos[act[i]] = m;      // appended again at the top of the next step
            // The rows before the first dropped position are unchanged; t there too.  v and Gi1 lose the later rows' terms:
            // v = Q[:first]' t, Gi1 = R^{-1}[:first, :first] t -- this workgroup's columns of them, the others' through the scratch row
            double ts[1] = {0.0};
            for (int i = tid; i < first; i += CT) ts[0] = fma(tv[i], tv[i], ts[0]);
            {
                const int ops[1] = {0};
                c_reduce(ts, ops, red, red_phase);          // (its barrier also publishes pos)
            }
            tsq = ts[0];
            double* scr = a.upart + (size_t)(scratch_passes++ & 1) * 2 * ld;
            c_col_mv(L, Q, RT, ld, first, first, tv, part,
                [&](int j, double2 t) { *reinterpret_cast<double2*>(scr + j) = t; },
                [&](int i, double2 t) { *reinterpret_cast<double2*>(scr + ld + i) = t; });
            LARS_C_SYNC();
#pragma unroll
            for (int h = 0; h < VPT; h += 2) {
                const int jh = j0 + h;
                if (jh < ld) {
                    const double2 t = *reinterpret_cast<const double2*>(scr + jh);
                    r_v[h] = t.x; r_v[h + 1] = t.y;
                }
                if (jh < first) {
                    const double2 t = *reinterpret_cast<const double2*>(scr + ld + jh);
                    gi1[jh] = t.x;
                    if (jh + 1 < first) gi1[jh + 1] = t.y;
                }
            }
            c_lds_barrier();
            re_i = first; re_end = keep;
            na = keep;
        }
        CTICK(9);
    }
#ifdef DLSA_LARS_PROF
    if (tid == 0 && writer)
        printf("LARS_C_PROF p=%d G=%d CP=%d RG=%d steps=%d us: prologue %.0f | gather %.0f sums %.0f pass %.0f barrier %.0f readback %.0f | "
               "select+append %.0f mins %.0f move+record %.0f drops %.0f\n", p, G, L.CP, L.RG, k, c_prof_t[0] * 0.01, c_prof_t[1] * 0.01, c_prof_t[2] * 0.01,
               c_prof_t[3] * 0.01, c_prof_t[4] * 0.01, c_prof_t[5] * 0.01, c_prof_t[6] * 0.01, c_prof_t[7] * 0.01, c_prof_t[8] * 0.01, c_prof_t[9] * 0.01);
#endif
    if (tid == 0 && writer) *a.n_steps = k;
#undef LARS_C_ABORT
#undef LARS_C_SYNC
}

}  // namespace

// Wide paths: LARS_C_MIN_M < m <= LARS_C_MAX_M.  dlsa_kernel_options.lars_q = 0 keeps lars.hip's kernels for every width (A/B runs,
// tests of both forms), as for lars_q.hip; 2 takes this kernel for every width it can (the crossover runs of bench/lars_crossover.py).
// Tried and dropped: the transpose of Q beside Q, so that the next append's r is one contiguous read instead of a 16 KB-strided
// gather -- 3.4 -> 3.0 us of a step at m = 2000, the path's time unchanged (35.0 ms): the gather waits for a fabric round trip to rows
// other XCDs wrote (every grid barrier's agent-scope acquire empties this XCD's L2), not for its own access pattern.
bool lars_c_eligible(int p, int intercept) {
    const int m = p - (intercept ? 1 : 0);
    if (m > LARS_C_MAX_M || lars_c_lds_bytes(m) + 1024 > (size_t)kLdsBytes) return false;
    // dlsa_kernel_options.lars_q: automatic = the measured hand-over (LARS_C_MIN_M); 0: lars.hip for every width; 1: lars_q.hip for every
    // width it takes (<= LARS_Q_MAX_M), this kernel beyond; 2: this kernel for every width it can take (A/B runs, tests)
    const char* e = kernel_knob("DLSA_LARS_Q");
    if (!e) return m > LARS_C_MIN_M;
    const int mode = atoi(e);
    if (mode == 0) return false;
    return mode == 2 ? m >= 64 : m > LARS_Q_MAX_M;
}

// Workgroups: two 16-column blocks of Q (and of RT) per workgroup -- m = 2000: 63 workgroups of 32 column-pair slots x 16 row groups;
// dlsa_kernel_options.lars_wgs overrides (2 .. LARS_C_MAX_WGS; every count gives the same path: the sums over the rows are taken in
// an order that depends on the count only to rounding).
int lars_c_run(LarsArgs& a, int p, int intercept, hipStream_t s, int* wgs_used) {
    const int m = p - (intercept ? 1 : 0);
    const int ld = (m + 1) & ~1, nblk = (ld + 15) / 16;
    int nwg = (nblk + 1) / 2;
    if (const char* e = kernel_knob("DLSA_LARS_WGS")) nwg = atoi(e);
    nwg = std::max(2, std::min(nwg, std::min(LARS_C_MAX_WGS, nblk)));
    // the blocks per workgroup the count implies must leave at least one row group (16 NBW column-pair slots <= CT threads)
    while (16 * ((nblk + nwg - 1) / nwg) > CT) ++nwg;
    const size_t shm = lars_c_lds_bytes(m);
    DLSA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(lars_c_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
    a.nwg = nwg;
    DLSA_HIP_CHECK(hipMemsetAsync(a.bar, 0, 256, s));
    void* kargs[] = {(void*)&a};
    if (launch_cooperative(reinterpret_cast<const void*>(lars_c_kernel), dim3(nwg), dim3(CT), kargs, shm, s) != hipSuccess)
        hipLaunchKernelGGL(lars_c_kernel, dim3(nwg), dim3(CT), shm, s, a);
    DLSA_HIP_CHECK(hipGetLastError());
    if (wgs_used) *wgs_used = nwg;
    return DLSA_OK;
}

}  // namespace dlsa
