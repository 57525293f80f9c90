// Structured passes for one-hot designs (reference: the dummy path of logistic_model,
// dlsa/models.py:56-131, where the design matrix is [intercept | standardised numerics | one-hot factor levels]).
//
// With f factors a row of the p-column design has only D + f non-zeros (D = intercept + numeric columns), so
//   eta_i = d_i . beta_D + sum_t beta[col(t, code_it)]                       is a GATHER,
//   g     = X'r:  g_D = sum r_i d_i,  g[col(t, l)] = sum_{code_it = l} r_i   is a HISTOGRAM,
//   H     = X'WX: H_DD (D x D, dense), H[a][col(t,l)] = sum_{code_it=l} w_i d_ia  (vector histogram per level),
//                 H[col(t,l)][col(t',l')] = sum_{code_it=l, code_it'=l'} w_i      (weighted co-occurrence counts)
// and both passes read the 8q + 4f raw bytes of a row (76 B for the airline-shaped config 4) instead of the 8p
// bytes of the dense row (2080 B): HBM-bound on a 27x smaller stream, no MFMA work at all.  The p x p result is
// the same matrix the dense Gram kernel produces (to rounding), so Cholesky / WLS / LARS are unchanged.
//
// One thread owns one row (64 distinct rows per wave: the transcendentals are not replicated).  Histograms are
// accumulated with LDS atomics in per-workgroup tables and flushed to per-workgroup partials that a second kernel
// sums in a fixed order.  The waves of a workgroup add to its tables in a fixed order as well (OhDesc::ordered, the
// default: turn-taking in the logit pass, a systolic wave-by-unit schedule in the Gram), so results are bit-identical
// from run to run like those of the dense kernels; DLSA_OH_ORDERED=0 lets all waves add at once (faster Gram, last bits vary).  The factor-pair tables of the Gram are dealt to
// workgroup ROLES so that each role's tables fit in LDS (every role streams all rows; they are cheap).
#include "common.h"
#include <vector>
#include <algorithm>
#include <string.h>

namespace dlsa {

constexpr int OH_MAXD = 8;            // dense columns (intercept + numerics) handled in registers
constexpr int OH_MAXF = 8;            // factors
constexpr int OH_THREADS = 256;
#ifndef DLSA_OH_GRAM_THREADS
#define DLSA_OH_GRAM_THREADS 1024
#endif
constexpr int OH_GRAM_THREADS = DLSA_OH_GRAM_THREADS;      // the Gram pass: one workgroup per CU (its tables fill the LDS), so all its latency hiding is waves
constexpr int OH_LDS_BUDGET = 152 * 1024;     // bytes of histogram tables per workgroup role
constexpr int OH_LOGIT_REP = 8;             // LDS copies (at most) of the logit pass's residual histogram
// copies actually used: as many as keep the workgroup's LDS near 32 KB (several workgroups per CU), at least one
static int oh_logit_rep(int p) {
    int r = OH_LOGIT_REP;
    while (r > 1 && (size_t)(1 + r) * p * sizeof(double) > 32 * 1024) r /= 2;
    return r;
}
constexpr int OH_MAX_BLOCKS = 512;          // two workgroups per CU; every workgroup flushes its tables once

struct OhTable {                      // one factor-pair table of a Gram role (t <= u; t == u: the diagonal counts), or a BAND of its rows
    int t, u;                         // factor indices
    int lds_off;                      // offset (doubles) of its ltn x L_u (or ltn) cells in the role's LDS image
    int lt0, ltn;                     // the levels lt0 .. lt0 + ltn - 1 of factor t: a table larger than the LDS budget is cut into
                                      // row bands that go to different roles (300 x 300 levels: five bands of 64 rows)
};

struct OhRole {
    int ntab;
    OhTable tab[OH_MAXF * (OH_MAXF + 1) / 2];
    int with_dense;                   // this role also accumulates H_DD and H_D,dummy
    int dense_off;                    // offset of the D x nlev_total block (H_D,dummy), if with_dense
    int cells;                        // doubles in the LDS image (and in the role's partial)
    int dense_rep;                    // copies of the H_D,dummy block in LDS (copy r >= 1 sits after the image, at
                                      // cells + (r-1) * nlev_total * OH_MAXD): lanes spread over them, so the lanes of a
                                      // wave that share a hot level do not all serialise on the same eight addresses
};

struct OhDesc {                       // device-visible description of the design
    int p, D, f;
    int dense_kind[OH_MAXD];          // 0: constant 1, 1: numeric column dense_src
    int dense_src[OH_MAXD];
    double dense_shift[OH_MAXD], dense_scale[OH_MAXD];
    int dense_col[OH_MAXD];           // output column of dense column a
    int lvl_off[OH_MAXF + 1];         // factor t's levels occupy [lvl_off[t], lvl_off[t+1]) of level_col
    int nlev_total;
    int dbg;                          // DLSA_OH_DBG (timing experiments only, wrong results): 1 = no dense x level atomics, 2 = no pair-table atomics
    int ordered;                      // LDS accumulation of the passes.  2 (default, Gram): EXACT -- every addend goes in as a 64-bit
                                      // fixed-point integer (ds_add_u64), integer addition is associative, so all waves add at once
                                      // and the result is bit-identical from run to run whatever the order; 1: floating-point adds in
                                      // a fixed wave order (turn-taking in the logit pass, the systolic schedule in the Gram;
                                      // DLSA_OH_ORDERED=1, and the Gram's fall-back when an addend leaves the fixed-point range);
                                      // 0 (DLSA_OH_ORDERED=0): floating-point adds from all waves at once, last bits vary
    int* overflow;                    // exact mode: set to 1 by a thread whose addend exceeds OH_FIX_VMAX (or is not finite)
};

}  // namespace dlsa

struct dlsa_onehot_plan {
    dlsa::OhDesc desc;
    int32_t* d_level_col;             // device: column of every (factor, level), -1 = no column (baseline / dropped)
    std::vector<int32_t> h_level_col;
    std::vector<dlsa::OhRole> roles;
    dlsa::OhRole* d_roles;
    bool needs_num;                   // some dense column is numeric
};

namespace dlsa {

// standardised dense vector of row i (d[a], a < D)
__device__ __forceinline__ void oh_dense_row(const OhDesc& ds, const double* __restrict__ num, int64_t ldn, int64_t i,
                                             double (&d)[OH_MAXD]) {
#pragma unroll
    for (int a = 0; a < OH_MAXD; ++a) {
        d[a] = 0.0;
        if (a < ds.D) d[a] = ds.dense_kind[a] == 0 ? 1.0 : (num[i * ldn + ds.dense_src[a]] - ds.dense_shift[a]) / ds.dense_scale[a];
    }
}

__device__ __forceinline__ double oh_exp_neg(double a) {      // exp(-a), a >= 0 (as logit.hip)
    a = fmin(a, 745.2);
    const double kf = rint(a * 1.4426950408889634);
    double r = fma(kf, 6.93147180369123816490e-01, -a);
    r = fma(kf, 1.90821492927058770002e-10, r);
    double q = 1.6059043836821613e-10;
    q = fma(q, r, 2.08767569878681e-09);
    q = fma(q, r, 2.505210838544172e-08);
    q = fma(q, r, 2.755731922398589e-07);
    q = fma(q, r, 2.7557319223985893e-06);
    q = fma(q, r, 2.48015873015873e-05);
    q = fma(q, r, 1.984126984126984e-04);
    q = fma(q, r, 1.388888888888889e-03);
    q = fma(q, r, 8.333333333333333e-03);
    q = fma(q, r, 4.1666666666666664e-02);
    q = fma(q, r, 1.6666666666666666e-01);
    q = fma(q, r, 0.5);
    q = fma(q, r, 1.0);
    q = fma(q, r, 1.0);
    return ldexp(q, -(int)kf);
}

__device__ __forceinline__ double oh_block_sum(double v, double* red) {
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double s = 0.0;
    for (int k = 0; k < (int)(blockDim.x >> 6); ++k) s += red[k];
    return s;
}

// ---------------------------------------------------------------------------------------------------------------
// logit pass: w, per-workgroup partial of g (p doubles) and loglik
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(OH_THREADS) void oh_logit_kernel(OhDesc ds, const int32_t* __restrict__ level_col,
                                                              const double* __restrict__ num, int64_t ldn,
                                                              const int32_t* __restrict__ codes, int64_t ldc,
                                                              const double* __restrict__ y, const double* __restrict__ beta,
                                                              int64_t n, double* __restrict__ w_out,
                                                              double* __restrict__ gpart, double* __restrict__ llpart, int nrep) {
    extern __shared__ double sm[];
    double* sbeta = sm;                           // p
    double* sg = sm + ds.p;                       // nrep x p (histograms of residuals; lanes spread over the copies,
                                                  // so the lanes of a wave that share a hot level do not serialise on one address)
    int* scol = reinterpret_cast<int*>(sm + (1 + nrep) * ds.p);     // nlev_total
    double* red = reinterpret_cast<double*>(scol + ((ds.nlev_total + 1) & ~1));
    for (int j = threadIdx.x; j < ds.p; j += blockDim.x) sbeta[j] = beta[j];
    for (int j = threadIdx.x; j < nrep * ds.p; j += blockDim.x) sg[j] = 0.0;
    double* sg_mine = sg + (threadIdx.x % nrep) * ds.p;
    for (int j = threadIdx.x; j < ds.nlev_total; j += blockDim.x) scol[j] = level_col[j];
    __syncthreads();
    double gd[OH_MAXD];
#pragma unroll
    for (int a = 0; a < OH_MAXD; ++a) gd[a] = 0.0;
    double ll = 0.0;
    // every thread runs the same number of rounds (the ordered mode has barriers inside): rows past n are clamped and masked
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t rounds = (n - (int64_t)blockIdx.x * blockDim.x + stride - 1) / stride;
    const int mywave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    for (int64_t rd = 0; rd < rounds; ++rd) {
        const int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x + rd * stride;
        const bool valid = i0 < n;
        const int64_t i = valid ? i0 : n - 1;
        double d[OH_MAXD];
        oh_dense_row(ds, num, ldn, i, d);
        double eta = 0.0;
#pragma unroll
        for (int a = 0; a < OH_MAXD; ++a)
            if (a < ds.D) eta = fma(d[a], sbeta[ds.dense_col[a]], eta);
        int cols[OH_MAXF];
#pragma unroll
        for (int t = 0; t < OH_MAXF; ++t) {
            cols[t] = -1;
            if (t < ds.f) {
                const int code = codes[i * ldc + t];
                const int nl = ds.lvl_off[t + 1] - ds.lvl_off[t];
                if (code >= 0 && code < nl) cols[t] = scol[ds.lvl_off[t] + code];
                if (cols[t] >= 0) eta += sbeta[cols[t]];
            }
        }
        const double yv = y[i];
        const double e = oh_exp_neg(fabs(eta));
        double inv = __builtin_amdgcn_rcp(1.0 + e);
        inv = fma(fma(-(1.0 + e), inv, 1.0), inv, inv);
        inv = fma(fma(-(1.0 + e), inv, 1.0), inv, inv);
        const double mu = eta >= 0.0 ? inv : e * inv;
        if (w_out && valid) w_out[i] = e * inv * inv;
        const double r = valid ? yv - mu : 0.0;
        if (valid) ll += yv * eta - (fmax(eta, 0.0) + log1p(e));
#pragma unroll
        for (int a = 0; a < OH_MAXD; ++a) gd[a] = fma(r, d[a], gd[a]);
        if (ds.ordered) {                           // one wave at a time, in wave order: a fixed order of the LDS adds (modes 1 and 2)
            for (int turn = 0; turn < nwaves; ++turn) {
                if (turn == mywave && valid) {
#pragma unroll
                    for (int t = 0; t < OH_MAXF; ++t)
                        if (t < ds.f && cols[t] >= 0) unsafeAtomicAdd(&sg_mine[cols[t]], r);
                }
                __syncthreads();
            }
        } else if (valid) {
#pragma unroll
            for (int t = 0; t < OH_MAXF; ++t)
                if (t < ds.f && cols[t] >= 0) unsafeAtomicAdd(&sg_mine[cols[t]], r);
        }
    }
    __syncthreads();
#pragma unroll
    for (int a = 0; a < OH_MAXD; ++a) {
        const double sgd = oh_block_sum(gd[a], red);
        if (threadIdx.x == 0 && a < ds.D) sg[ds.dense_col[a]] += sgd;
    }
    const double sll = oh_block_sum(ll, red);
    __syncthreads();
    double* gp = gpart + (int64_t)blockIdx.x * ds.p;
    for (int j = threadIdx.x; j < ds.p; j += blockDim.x) {
        double t = sg[j];
        for (int r = 1; r < nrep; ++r) t += sg[r * ds.p + j];      // fixed order
        gp[j] = t;
    }
    if (threadIdx.x == 0) llpart[blockIdx.x] = sll;
}

// g[j] = sum_b gpart[b][j], loglik = sum_b llpart[b] in a fixed order: the dense pass's finish kernel (logit.hip)
void logit_finish_launch(const double* gpart, const double* llpart, int nblocks, int pitch, int p, double* g,
                         double* loglik, hipStream_t stream, const double* s0part, double* s0);

// ---------------------------------------------------------------------------------------------------------------
// Exact LDS accumulation.  An addend v (|v| <= OH_FIX_VMAX = 16: w d with w <= 1/4 and a standardised numeric below 64 sigma,
// or a weight itself) is rounded ONCE to a multiple of 2^-OH_FIX_S and added as a two's-complement 64-bit integer.  Integer
// addition is associative and commutative, so the LDS atomics of sixteen waves may land in any order: the table a workgroup
// flushes is the same bit pattern every run -- determinism without the wave turn-taking that cost 60 % (2.67 vs 1.66 ms on
// config 4's shard).  Rounding error: 2^-41 per addend (4.5e-13 absolute; a workgroup's ~3e4 addends to a cell: <= 1.4e-8 worst
// case, ~8e-11 typical, on sums of 1e2..1e4 -- below the fp64 rounding of the ordered sum it replaces).  Range: a workgroup
// adds at most rows_per_workgroup * 16 * 2^40 < 2^62 (checked on the host: rows_per_workgroup < 2^18).  The conversion is the
// magic-number rounding (v 2^S + 1.5 2^52, valid below 2^51) -- two VALU ops, no 64-bit float->int instruction needed.
// ---------------------------------------------------------------------------------------------------------------
constexpr int OH_FIX_S = 40;
constexpr double OH_FIX_VMAX = 16.0;
__device__ __forceinline__ long long oh_to_fixed(double v) {
    const double magic = 6755399441055744.0;                        // 1.5 * 2^52
    const double t = fma(v, (double)(1ull << OH_FIX_S), magic);
    return __double_as_longlong(t) - __double_as_longlong(magic);
}
__device__ __forceinline__ double oh_from_fixed(long long q) { return (double)q * (1.0 / (double)(1ull << OH_FIX_S)); }

// ---------------------------------------------------------------------------------------------------------------
// Gram: a workgroup of role r accumulates r's tables in LDS and writes them to its slot of the partial buffer
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(OH_GRAM_THREADS) void oh_gram_kernel(OhDesc ds, const OhRole* __restrict__ roles, int nroles,
                                                             int blocks_per_role, const double* __restrict__ num, int64_t ldn,
                                                             const int32_t* __restrict__ codes, int64_t ldc,
                                                             const double* __restrict__ w, int64_t n,
                                                             double* __restrict__ partial, int64_t role_stride) {
    extern __shared__ double sm[];
    const int role_id = blockIdx.x / blocks_per_role;
    const int bl = blockIdx.x % blocks_per_role;
    const OhRole& role = roles[role_id];
    const int dblock = ds.nlev_total * OH_MAXD;
    const int lds_cells = role.cells + (role.with_dense ? (role.dense_rep - 1) * dblock : 0);
    double* tab = sm;                              // lds_cells
    double* red = sm + lds_cells;                  // 16
    for (int j = threadIdx.x; j < lds_cells; j += blockDim.x) tab[j] = 0.0;
    const int my_rep = role.with_dense ? (int)(threadIdx.x % role.dense_rep) : 0;
    double* dense_tab = my_rep == 0 ? tab + role.dense_off : tab + role.cells + (my_rep - 1) * dblock;
    __syncthreads();
    double hdd[OH_MAXD * (OH_MAXD + 1) / 2];
#pragma unroll
    for (int k = 0; k < OH_MAXD * (OH_MAXD + 1) / 2; ++k) hdd[k] = 0.0;
    const bool dense = role.with_dense != 0;
    const int64_t stride = (int64_t)blocks_per_role * blockDim.x;
    const int64_t rounds = (n - (int64_t)bl * blockDim.x + stride - 1) / stride;
    const int mywave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    for (int64_t rd = 0; rd < rounds; ++rd) {
        const int64_t i0 = (int64_t)bl * blockDim.x + threadIdx.x + rd * stride;
        const bool valid = i0 < n;
        const int64_t i = valid ? i0 : n - 1;
        const double wi = valid ? (w ? w[i] : 1.0) : 0.0;
        int lv[OH_MAXF];
#pragma unroll
        for (int t = 0; t < OH_MAXF; ++t) {
            lv[t] = -1;
            if (t < ds.f) {
                const int code = codes[i * ldc + t];
                if (code >= 0 && code < ds.lvl_off[t + 1] - ds.lvl_off[t]) lv[t] = code;
            }
        }
        double d[OH_MAXD];
        if (dense) {
            oh_dense_row(ds, num, ldn, i, d);
            int k = 0;
#pragma unroll
            for (int a = 0; a < OH_MAXD; ++a)
#pragma unroll
                for (int b = a; b < OH_MAXD; ++b, ++k) hdd[k] = fma(wi * d[a], d[b], hdd[k]);
        }
        // the row's LDS adds, one UNIT at a time: unit t < nd = the dense x level block of factor t (D adds), unit nd + q = pair table q.
        // Units own disjoint cells.  (Measured on config 4's shard, ordered Gram: these 20 units 2.65 ms; the dense blocks in
        // halves 2.71; one fat unit per factor 3.62; single-add units with a run-time column select 3.30.)
        const int nd = dense ? ds.f : 0, nunit = nd + role.ntab;
        const bool exact = ds.ordered == 2;
        auto add_cell = [&](double* cell, double v) {
            if (exact) {
                if (!(fabs(v) <= OH_FIX_VMAX)) { *ds.overflow = 1; return; }          // (NaN fails the test too)
                atomicAdd(reinterpret_cast<unsigned long long*>(cell), (unsigned long long)oh_to_fixed(v));
            } else {
                unsafeAtomicAdd(cell, v);
            }
        };
        auto unit = [&](int u) {
            if (u < nd) {
                int l = -1;
#pragma unroll
                for (int t = 0; t < OH_MAXF; ++t) l = (t == u) ? lv[t] : l;
                if (l < 0) return;
                double* dst = dense_tab + (ds.lvl_off[u] + l) * OH_MAXD;
#pragma unroll
                for (int a = 0; a < OH_MAXD; ++a)
                    if (a < ds.D && !DLSA_DBG_WRONG(ds.dbg, 1)) add_cell(dst + a, wi * d[a]);
            } else {
                const OhTable tb = role.tab[u - nd];
                int lt = -1, lu = -1;
#pragma unroll
                for (int t = 0; t < OH_MAXF; ++t) { lt = (t == tb.t) ? lv[t] : lt; lu = (t == tb.u) ? lv[t] : lu; }
                if (lt < 0 || lu < 0 || DLSA_DBG_WRONG(ds.dbg, 2)) return;
                lt -= tb.lt0;
                if (lt < 0 || lt >= tb.ltn) return;                         // another band's row
                if (tb.t == tb.u) add_cell(tab + tb.lds_off + lt, wi);
                else add_cell(tab + tb.lds_off + lt * (ds.lvl_off[tb.u + 1] - ds.lvl_off[tb.u]) + lu, wi);
            }
        };
        if (ds.ordered == 1) {
            // Ordered mode: wave k works on unit (step - k) -- a systolic schedule with a barrier between steps.  Every unit has
            // its own cells, and the waves reach a unit one after another in wave order, so the adds to any cell happen in a
            // fixed order (bit-reproducible) while all waves keep the LDS atomic pipeline busy on different units.
            for (int step = 0; step < nunit + nwaves - 1; ++step) {
                const int u = step - mywave;
                if (valid && u >= 0 && u < nunit) unit(u);
                __syncthreads();
            }
        } else if (valid) {
            for (int u = 0; u < nunit; ++u) unit(u);
        }
    }
    __syncthreads();
    const bool exact_out = ds.ordered == 2;
    if (role.with_dense && role.dense_rep > 1) {   // fold the copies (fixed order; exact mode: integer sums)
        for (int j = threadIdx.x; j < dblock; j += blockDim.x) {
            if (exact_out) {
                long long t = __double_as_longlong(tab[role.dense_off + j]);
                for (int r = 1; r < role.dense_rep; ++r) t += __double_as_longlong(tab[role.cells + (r - 1) * dblock + j]);
                tab[role.dense_off + j] = __longlong_as_double(t);
            } else {
                double t = tab[role.dense_off + j];
                for (int r = 1; r < role.dense_rep; ++r) t += tab[role.cells + (r - 1) * dblock + j];
                tab[role.dense_off + j] = t;
            }
        }
        __syncthreads();
    }
    double* out = partial + (int64_t)role_id * role_stride + (int64_t)bl * (role.cells + OH_MAXD * (OH_MAXD + 1) / 2);
    for (int j = threadIdx.x; j < role.cells; j += blockDim.x) out[j] = exact_out ? oh_from_fixed(__double_as_longlong(tab[j])) : tab[j];
    if (dense) {
#pragma unroll
        for (int k = 0; k < OH_MAXD * (OH_MAXD + 1) / 2; ++k) {
            const double s = oh_block_sum(hdd[k], red);
            if (threadIdx.x == 0) out[role.cells + k] = s;
        }
    }
}

// 32 cells x 8 block groups per workgroup: group y sums the partials of blocks y, y+8, ... of its cell, the groups are
// combined in a fixed order, and the value is scattered to H (both triangles); cells whose (factor, level) has no
// column are dropped
__global__ __launch_bounds__(256) void oh_gram_finish_kernel(OhDesc ds, const OhRole* __restrict__ roles, int role_id,
                                                             int blocks_per_role, const int32_t* __restrict__ level_col,
                                                             const double* __restrict__ partial, int64_t role_stride,
                                                             double* __restrict__ H, int64_t ldh) {
    __shared__ double red[8][33];
    const OhRole& role = roles[role_id];
    constexpr int NDD = OH_MAXD * (OH_MAXD + 1) / 2;
    const int per = role.cells + NDD;
    const int cx = threadIdx.x & 31, gy = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cx;
    double s = 0.0;
    if (c < per) {
        const double* src = partial + (int64_t)role_id * role_stride + c;
        for (int b = gy; b < blocks_per_role; b += 8) s += src[(int64_t)b * per];
    }
    red[gy][cx] = s;
    __syncthreads();
    if (gy != 0 || c >= per) return;
#pragma unroll
    for (int k = 1; k < 8; ++k) s += red[k][cx];
    int r0 = -1, c0 = -1;
    if (c >= role.cells) {                                   // H_DD, upper triangle order
        if (!role.with_dense) return;
        int k = c - role.cells, a = 0;
        while (k >= OH_MAXD - a) { k -= OH_MAXD - a; ++a; }
        const int b = a + k;
        if (a < ds.D && b < ds.D) { r0 = ds.dense_col[a]; c0 = ds.dense_col[b]; }
    } else if (role.with_dense && c >= role.dense_off && c < role.dense_off + ds.nlev_total * OH_MAXD) {     // H_D,dummy: [level][a]
        const int k = c - role.dense_off, lvl = k / OH_MAXD, a = k % OH_MAXD;
        if (a < ds.D && lvl < ds.nlev_total) { r0 = ds.dense_col[a]; c0 = level_col[lvl]; }
    } else {
        for (int q = 0; q < role.ntab; ++q) {
            const OhTable tb = role.tab[q];
            const int Lu = ds.lvl_off[tb.u + 1] - ds.lvl_off[tb.u];
            const int sz = tb.t == tb.u ? tb.ltn : tb.ltn * Lu;
            if (c >= tb.lds_off && c < tb.lds_off + sz) {
                const int k = c - tb.lds_off;
                const int lt = tb.lt0 + (tb.t == tb.u ? k : k / Lu), lu = tb.t == tb.u ? lt : k % Lu;
                r0 = level_col[ds.lvl_off[tb.t] + lt];
                c0 = level_col[ds.lvl_off[tb.u] + lu];
                break;
            }
        }
    }
    if (r0 < 0 || c0 < 0) return;
    H[(int64_t)r0 * ldh + c0] = s;
    H[(int64_t)c0 * ldh + r0] = s;
}

// ---------------------------------------------------------------------------------------------------------------
// host
// ---------------------------------------------------------------------------------------------------------------
static int oh_blocks(int64_t n) {
    const int64_t want = (n + OH_THREADS * 16 - 1) / (OH_THREADS * 16);
    return (int)std::max<int64_t>(1, std::min<int64_t>(want, OH_MAX_BLOCKS));
}
// The logit pass keeps little in LDS, so many small workgroups share a CU: four rows per thread, up to eight
// workgroups per CU (a 1e6-row partition: 977 workgroups instead of 244 -- one per CU, four waves, nothing to hide the
// row loads behind: 72 us for 76 MB)
constexpr int OH_LOGIT_MAX_BLOCKS = 2048;
static int oh_logit_blocks(int64_t n) {
    const int64_t want = (n + OH_THREADS * 4 - 1) / (OH_THREADS * 4);
    return (int)std::max<int64_t>(1, std::min<int64_t>(want, OH_LOGIT_MAX_BLOCKS));
}

int onehot_plan_p(const dlsa_onehot_plan* pl) { return pl->desc.p; }

size_t onehot_workspace_bytes_impl(const dlsa_onehot_plan* pl, int64_t n) {
    const int nb = oh_blocks(n), nbl = oh_logit_blocks(n);
    size_t logit = align_up((size_t)nbl * pl->desc.p * sizeof(double), 256) + align_up((size_t)nbl * sizeof(double), 256);
    size_t gram = 0;
    constexpr int NDD = OH_MAXD * (OH_MAXD + 1) / 2;
    for (auto& r : pl->roles) gram = std::max(gram, (size_t)(r.cells + NDD));
    gram = align_up(gram * nb * sizeof(double), 256) * pl->roles.size();
    return std::max(logit, gram) + 256;
}

int onehot_logit_pass_impl(const dlsa_onehot_plan* pl, const double* num, int64_t ldn, const int32_t* codes, int64_t ldc,
                           const double* y, const double* beta, int64_t n, double* w_out, double* g, double* loglik,
                           void* ws, size_t ws_bytes, hipStream_t s) {
    DLSA_REQUIRE(pl && y && beta && (num || !pl->needs_num || n == 0) && (codes || pl->desc.f == 0 || n == 0),
                 "onehot logit pass: null argument");
    OhDesc ds = pl->desc;
    { const char* e = kernel_knob("DLSA_OH_ORDERED"); ds.ordered = e ? (atoi(e) != 0) : 1; }      // the logit pass: wave turn-taking unless 0
    ds.overflow = nullptr;
    if (!ws || ws_bytes < onehot_workspace_bytes_impl(pl, n) || ((uintptr_t)ws & 255)) {
        set_error("onehot logit pass: workspace %zu bytes needed (256-aligned), got %zu", onehot_workspace_bytes_impl(pl, n), ws_bytes);
        return DLSA_ERR_WORKSPACE;
    }
    const int nb = oh_logit_blocks(n);
    Arena ar(ws, ws_bytes);
    double* gpart = (double*)ar.take((size_t)nb * ds.p * sizeof(double));
    double* llpart = (double*)ar.take((size_t)nb * sizeof(double));
    const int nrep = oh_logit_rep(ds.p);
    const size_t shm = (size_t)((1 + nrep) * ds.p + 16) * sizeof(double) + (size_t)((ds.nlev_total + 1) & ~1) * sizeof(int);
    hipLaunchKernelGGL(oh_logit_kernel, dim3(nb), dim3(OH_THREADS), shm, s, ds, (const int32_t*)pl->d_level_col, num, ldn, codes,
                       ldc, y, beta, n, w_out, gpart, llpart, nrep);
    DLSA_HIP_CHECK(hipGetLastError());
    if (g || loglik) {
        logit_finish_launch((const double*)gpart, (const double*)llpart, nb, ds.p, ds.p, g, loglik, s, nullptr, nullptr);
        DLSA_HIP_CHECK(hipGetLastError());
    }
    return DLSA_OK;
}

// irls_weights: w are the logistic weights mu (1 - mu) in (0, 1/4] of this library's own logit pass (the IRLS driver) -- the addends'
// scale is known and the exact fixed-point sums (absolute resolution 2^-40) are the default.  A CALLER's weights (the public
// dlsa_onehot_gram) may have any scale -- w ~ 1e-8 would keep five digits, w < 4.5e-13 none -- so they are summed in ordered
// floating point (full fp64 relative accuracy at any scale, also bit-reproducible, slower).
int onehot_gram_impl(const dlsa_onehot_plan* pl, const double* num, int64_t ldn, const int32_t* codes, int64_t ldc,
                     const double* w, int64_t n, double* H, int64_t ldh, void* ws, size_t ws_bytes, hipStream_t s, bool irls_weights) {
    DLSA_REQUIRE(pl && H && ldh >= pl->desc.p && (num || !pl->needs_num || n == 0) && (codes || pl->desc.f == 0 || n == 0),
                 "onehot gram: null argument or ldh < p");
    OhDesc ds = pl->desc;
    ds.dbg = 0;
#ifdef DLSA_DEBUG_KNOBS
    { const char* e = getenv("DLSA_OH_DBG"); ds.dbg = e ? atoi(e) : 0; }
#endif
    // accumulation mode of the LDS tables: exact fixed-point (2, the default), ordered floating point (DLSA_OH_ORDERED=1),
    // unordered floating point (DLSA_OH_ORDERED=0)
    { const char* e = kernel_knob("DLSA_OH_ORDERED"); ds.ordered = e ? (atoi(e) != 0 ? 1 : 0) : (irls_weights || !w ? 2 : 1); }
    const size_t ws_need = onehot_workspace_bytes_impl(pl, n);
    if (!ws || ws_bytes < ws_need || ((uintptr_t)ws & 255)) {
        set_error("onehot gram: workspace %zu bytes needed (256-aligned), got %zu", ws_need, ws_bytes);
        return DLSA_ERR_WORKSPACE;
    }
    const int nb = oh_blocks(n);
    constexpr int NDD = OH_MAXD * (OH_MAXD + 1) / 2;
    size_t per_role = 0, max_cells = 0;
    for (auto& r : pl->roles) {
        per_role = std::max(per_role, (size_t)(r.cells + NDD));
        max_cells = std::max(max_cells, (size_t)r.cells + (r.with_dense ? (size_t)(r.dense_rep - 1) * ds.nlev_total * OH_MAXD : 0));
    }
    const int64_t role_stride = (int64_t)(align_up(per_role * nb * sizeof(double), 256) / sizeof(double));
    const int nroles = (int)pl->roles.size();
    int* flag = (int*)((char*)ws + ws_need - 256);            // the spare tail of the workspace
    ds.overflow = flag;
    // exact mode's range: a workgroup's rows x OH_FIX_VMAX x 2^OH_FIX_S must stay below 2^62
    const int64_t rows_per_wg = (n + nb - 1) / nb + OH_GRAM_THREADS;
    if (ds.ordered == 2 && (double)rows_per_wg * OH_FIX_VMAX * (double)(1ull << OH_FIX_S) >= 4.0e18) ds.ordered = 1;
    const size_t shm = (max_cells + 16) * sizeof(double);
    if (shm > 64 * 1024)
        DLSA_HIP_CHECK(hipFuncSetAttribute((const void*)oh_gram_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
    for (int attempt = 0; attempt < 2; ++attempt) {
        if (ds.ordered == 2) DLSA_HIP_CHECK(hipMemsetAsync(flag, 0, sizeof(int), s));
        DLSA_HIP_CHECK(hipMemset2DAsync(H, (size_t)ldh * sizeof(double), 0, (size_t)ds.p * sizeof(double), (size_t)ds.p, s));
        hipLaunchKernelGGL(oh_gram_kernel, dim3(nb * nroles), dim3(OH_GRAM_THREADS), shm, s, ds, (const OhRole*)pl->d_roles, nroles, nb,
                           num, ldn, codes, ldc, w, n, (double*)ws, role_stride);
        DLSA_HIP_CHECK(hipGetLastError());
        for (int r = 0; r < nroles; ++r) {
            const int per = pl->roles[r].cells + NDD;
            hipLaunchKernelGGL(oh_gram_finish_kernel, dim3((per + 31) / 32), dim3(256), 0, s, ds, (const OhRole*)pl->d_roles, r, nb,
                               (const int32_t*)pl->d_level_col, (const double*)ws, role_stride, H, ldh);
        }
        DLSA_HIP_CHECK(hipGetLastError());
        if (ds.ordered != 2) break;
        // an addend outside the fixed-point range (|w d| > 16: a numeric beyond 64 sigma, or caller weights above 1/4 times that)
        // or a NaN: the launch is repeated with ordered floating-point adds, which take anything
        int over = 0;
        DLSA_HIP_CHECK(hipMemcpyAsync(&over, flag, sizeof(int), hipMemcpyDeviceToHost, s));
        DLSA_HIP_CHECK(hipStreamSynchronize(s));
        if (!over) break;
        ds.ordered = 1;
    }
    return DLSA_OK;
}

}  // namespace dlsa

extern "C" {

int dlsa_onehot_plan_create(int p, int ndense, const int32_t* dense_kind, const int32_t* dense_src,
                            const double* dense_shift, const double* dense_scale, const int32_t* dense_col,
                            int nfactor, const int32_t* nlevels, const int32_t* level_col, dlsa_onehot_plan** out) {
    using namespace dlsa;
    DLSA_REQUIRE(out && p > 0 && p <= 2048, "onehot plan: bad p=%d", p);
    DLSA_REQUIRE(ndense >= 0 && ndense <= OH_MAXD, "onehot plan: %d dense columns (intercept + numerics), at most %d", ndense, OH_MAXD);
    DLSA_REQUIRE(nfactor >= 0 && nfactor <= OH_MAXF, "onehot plan: %d factors, at most %d", nfactor, OH_MAXF);
    DLSA_REQUIRE(ndense == 0 || (dense_kind && dense_src && dense_shift && dense_scale && dense_col), "onehot plan: null dense descriptor");
    DLSA_REQUIRE(nfactor == 0 || (nlevels && level_col), "onehot plan: null factor descriptor");
    dlsa_onehot_plan* pl = new dlsa_onehot_plan();
    OhDesc& ds = pl->desc;
    memset(&ds, 0, sizeof(ds));
    ds.p = p; ds.D = ndense; ds.f = nfactor;
    pl->needs_num = false; pl->d_level_col = nullptr; pl->d_roles = nullptr;
    for (int a = 0; a < ndense; ++a) pl->needs_num |= (dense_kind[a] == 1);
    std::vector<char> used((size_t)p, 0);
    auto fail = [&](const char* msg) { set_error("onehot plan: %s", msg); delete pl; return DLSA_ERR_INVALID; };
    for (int a = 0; a < ndense; ++a) {
        ds.dense_kind[a] = dense_kind[a]; ds.dense_src[a] = dense_src[a];
        ds.dense_shift[a] = dense_shift[a]; ds.dense_scale[a] = dense_scale[a]; ds.dense_col[a] = dense_col[a];
        if (dense_col[a] < 0 || dense_col[a] >= p || used[dense_col[a]]) return fail("dense column out of range or used twice");
        used[dense_col[a]] = 1;
    }
    ds.lvl_off[0] = 0;
    for (int t = 0; t < nfactor; ++t) {
        if (nlevels[t] <= 0) return fail("a factor without levels");
        ds.lvl_off[t + 1] = ds.lvl_off[t] + nlevels[t];
    }
    ds.nlev_total = ds.lvl_off[nfactor];
    pl->h_level_col.assign(level_col, level_col + ds.nlev_total);
    for (int c : pl->h_level_col) {
        if (c < -1 || c >= p) return fail("level column out of range");
        if (c >= 0) { if (used[c]) return fail("a column is produced twice"); used[c] = 1; }
    }
    for (int j = 0; j < p; ++j) if (!used[j]) return fail("a design column has no source");
    // roles: the dense role (H_DD, H_D,dummy and as many pair tables as fit), then first-fit roles for the rest
    const int budget = OH_LDS_BUDGET / (int)sizeof(double);
    struct Pend { int t, u, cells, lt0, ltn; };
    std::vector<Pend> pend;
    for (int t = 0; t < nfactor; ++t)
        for (int u = t; u < nfactor; ++u) {
            const int row = t == u ? 1 : nlevels[u];                              // cells per level of factor t
            if (row > budget) return fail("a factor has too many levels for the structured path: use the dense path");
            // a table beyond the LDS budget is cut into bands of whole rows, as even as possible; every band is a table of its own
            int nband = (int)(((int64_t)nlevels[t] * row + budget - 1) / budget);
            while ((int64_t)((nlevels[t] + nband - 1) / nband) * row > budget) ++nband;       // (the largest band holds ceil(L_t / nband) rows)
            for (int b = 0; b < nband; ++b) {
                const int lo = (int)((int64_t)nlevels[t] * b / nband), hi = (int)((int64_t)nlevels[t] * (b + 1) / nband);
                if (hi > lo) pend.push_back(Pend{t, u, (hi - lo) * row, lo, hi - lo});
            }
        }
    std::sort(pend.begin(), pend.end(), [](const Pend& a, const Pend& b) { return a.cells > b.cells; });
    OhRole first; memset(&first, 0, sizeof(first));
    first.with_dense = 1; first.dense_off = 0; first.cells = ds.nlev_total * OH_MAXD;
    if (first.cells > budget) return fail("too many factor levels for the structured path");
    // up to four LDS copies of the H_D,dummy block, as long as they leave half of the budget to the pair tables
    first.dense_rep = 1;
    while (first.dense_rep < 4 && 2 * first.dense_rep * first.cells <= budget / 2) first.dense_rep *= 2;
    const int first_extra = (first.dense_rep - 1) * first.cells;
    pl->roles.push_back(first);
    for (auto& pd : pend) {
        constexpr int max_tab = (int)(sizeof(((OhRole*)nullptr)->tab) / sizeof(OhTable));
        OhRole* dst = nullptr;
        for (auto& r : pl->roles)
            if (r.ntab < max_tab && r.cells + (r.with_dense ? first_extra : 0) + pd.cells <= budget) { dst = &r; break; }
        if (!dst) { OhRole nr; memset(&nr, 0, sizeof(nr)); pl->roles.push_back(nr); dst = &pl->roles.back(); }
        dst->tab[dst->ntab++] = OhTable{pd.t, pd.u, dst->cells, pd.lt0, pd.ltn};
        dst->cells += pd.cells;
    }
    if (hipMalloc((void**)&pl->d_level_col, std::max<size_t>(1, pl->h_level_col.size()) * sizeof(int32_t)) != hipSuccess ||
        hipMalloc((void**)&pl->d_roles, pl->roles.size() * sizeof(OhRole)) != hipSuccess) {
        set_error("onehot plan: hipMalloc failed"); delete pl; return DLSA_ERR_HIP;
    }
    if (!pl->h_level_col.empty())
        DLSA_HIP_CHECK(hipMemcpy(pl->d_level_col, pl->h_level_col.data(), pl->h_level_col.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    DLSA_HIP_CHECK(hipMemcpy(pl->d_roles, pl->roles.data(), pl->roles.size() * sizeof(OhRole), hipMemcpyHostToDevice));
    *out = pl;
    return DLSA_OK;
}

void dlsa_onehot_plan_destroy(dlsa_onehot_plan* pl) {
    if (!pl) return;
    if (pl->d_level_col) (void)hipFree(pl->d_level_col);
    if (pl->d_roles) (void)hipFree(pl->d_roles);
    delete pl;
}

int dlsa_onehot_plan_roles(const dlsa_onehot_plan* pl) { return pl ? (int)pl->roles.size() : 0; }

size_t dlsa_onehot_workspace_bytes(const dlsa_onehot_plan* pl, int64_t n) {
    if (!pl || n < 0) return 0;
    return dlsa::onehot_workspace_bytes_impl(pl, n);
}

int dlsa_onehot_logit_pass_f64(const dlsa_onehot_plan* pl, const double* num, int64_t ldn, const int32_t* codes, int64_t ldc,
                               const double* y, const double* beta, int64_t n, double* w_out, double* g, double* loglik,
                               void* ws, size_t ws_bytes, void* stream) {
    return dlsa::onehot_logit_pass_impl(pl, num, ldn, codes, ldc, y, beta, n, w_out, g, loglik, ws, ws_bytes, (hipStream_t)stream);
}

int dlsa_onehot_gram_f64(const dlsa_onehot_plan* pl, const double* num, int64_t ldn, const int32_t* codes, int64_t ldc,
                         const double* w, int64_t n, double* H, int64_t ldh, void* ws, size_t ws_bytes, void* stream) {
    return dlsa::onehot_gram_impl(pl, num, ldn, codes, ldc, w, n, H, ldh, ws, ws_bytes, (hipStream_t)stream, false);
}

}  // extern "C"
