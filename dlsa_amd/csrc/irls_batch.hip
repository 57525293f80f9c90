// LOCK-STEP map step for MANY partitions of a narrow fp64 design (dlsa/models.py:110-131 per partition; `partition_num` is a user
// argument, models.py:32-33): the partitions of one call are fitted TOGETHER, one launch per stage of a Newton iteration over all
// of them, instead of one after the other (irls.hip: a dozen launches + a host round trip per iteration and partition -- 0.28 ms per
// partition on four chains whatever its size; 1000 partitions of 2e4 x 100 rows: 0.28 s for 20 ms worth of passes).
//
//   per iteration, four launches for ALL partitions:
//     1. the fused Newton pass in its batched form (irls_pass.hip): a table of slabs, each the rows of ONE partition with that
//        partition's own beta -> per-slab H partial, g, loglik;
//     2. batch_unpack_kernel: per partition, the slabs' partials in a fixed order -> H (dense, both triangles, straight into the
//        caller's Sig_inv block), g, loglik;
//     3. spd_inverse_small_kernel, batched (chol.hip): delta = H^-1 g, |delta|, |beta|, the pivot flag, for every live partition;
//     4. batch_update_kernel: the per-partition state machine of irls_small.hip / the oracle (log-likelihood safeguard with step
//        halving, stopping rule |delta|_inf <= tol max(1, |beta|_inf), max_iter) -- a partition that ends writes coef, Sig_inv . coef
//        and its status and leaves the active set; its slabs return at once in later passes.
//   One 4-byte read-back per iteration (the number of live partitions).
// Start (round 5): ONE fit from beta = 0 on a few leading rows of all partitions together, then every partition's full-row iterations
// from that pooled estimate (see `pooled` below; before: every partition's own fit on its leading quarter / eighth, still the fallback);
// the MLE and the Hessian at it are those of the host-driven path to the solver tolerance.  Shapes: what the fused pass and the one-launch inverse take (49 <= p
// + intercept <= 112; aligned rows of even width, or packed rows of odd width without the intercept; contiguous or i % K strided
// partitions); chosen by a cost model against the chained path.
// Scratch beyond the caller's workspace comes from the stream-ordered pool and is freed before the call returns.
#include "common.h"
#include "options.h"
#include "irls_batch.h"
#include <algorithm>
#include <cmath>
#include <vector>
#ifndef FP_TIMELINE
#define FP_TIMELINE 0
#endif
#include <stdio.h>
#include <stdlib.h>

namespace dlsa {

// irls_pass.hip
int irls_pass_batched_pp(int p);
int irls_pass_batched_gp(int p);
int irls_pass_batched_ll_at(int p);
bool irls_pass_batched_shape_ok(const double* X, int64_t ldx, const double* y, int p, int intercept, int64_t base_ldx);
int irls_pass_batched_launch(const double* X, int64_t ldx, const double* y, const double* beta, int64_t beta_stride, int p, int intercept,
                             const FusedSlab* d_slabs, int nslab, const int* d_active, double* partial, double* gpart,
                             unsigned long long* clk, hipStream_t stream, int want_h);
// chol.hip
bool chol_small_ok(int p);
int launch_chol_small_batched(int count, const double* A, int64_t lda, int64_t sA, int p, const double* rhs, const double* ref, int64_t sV,
                              double* Hinv, int64_t sH, double* xout, double* stats, int64_t sS, const int* active, hipStream_t s);

constexpr int BATCH_SLAB_ROWS = 24576;          // a partition longer than this is cut into equal slabs (whole 32-row chunks)

struct BatchState {            // per partition, device
    double ll_prev, dprev;     // dprev: the previous gradient-only step (its ratio to the next one is the contraction rate)
    int have_prev, halvings, iters, evals, last_pass, status, restarted;
};

// H, g, loglik of every live partition from its slabs' partials (fixed order over the slabs: bit-reproducible)
__global__ __launch_bounds__(256) void batch_unpack_kernel(const double* __restrict__ partial, const double* __restrict__ gpart,
                                                           const int* __restrict__ slab_begin, const int* __restrict__ active, int PP, int GP,
                                                           int ll_at, int p, int rot, double* __restrict__ H, double* __restrict__ g,
                                                           double* __restrict__ ll) {
    // p = columns of the kernel's design; rot: its LAST column is the implicit intercept, which the outputs carry FIRST (models.py:136-142)
    const int k = blockIdx.x;
    if (!active[k]) return;
    const int s0 = slab_begin[k], s1 = slab_begin[k + 1];
    double* __restrict__ Hk = H + (int64_t)k * p * p;
    auto om = [&](int c) { return rot ? (c == p - 1 ? 0 : c + 1) : c; };
    for (int e = threadIdx.x; e < (H ? p * p : 0); e += blockDim.x) {           // (H == nullptr: a gradient-only pass left no H partials)
        const int i = e / p, j = e - i * p;
        if (i > j) continue;
        double s = 0.0;
        for (int sl = s0; sl < s1; sl += 4) {            // (four slabs' loads in flight, added in their order)
            double v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = sl + u < s1 ? partial[(int64_t)(sl + u) * PP * PP + (int64_t)i * PP + j] : 0.0;
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (sl + u < s1) s += v[u];
        }
        Hk[om(i) * p + om(j)] = s;
        Hk[om(j) * p + om(i)] = s;
    }
    for (int j = threadIdx.x; j <= p; j += blockDim.x) {
        double s = 0.0;
        for (int sl = s0; sl < s1; ++sl) s += gpart[(int64_t)sl * GP + (j < p ? j : ll_at)];
        if (j < p) g[(int64_t)k * p + om(j)] = s;
        else ll[k] = s;
    }
}

// the Newton state machine of one partition after a pass at its current beta (irls_small.hip's, lsa-free part of models.py:110-131)
__global__ __launch_bounds__(128) void batch_update_kernel(int p, double tol, int max_iter, const double* __restrict__ H,
                                                           const double* __restrict__ ll_in, const double* __restrict__ delta,
                                                           const double* __restrict__ stats, double* __restrict__ beta,
                                                           double* __restrict__ prev, double* __restrict__ stepv, BatchState* __restrict__ st,
                                                           int* __restrict__ active, int* __restrict__ n_live, double* __restrict__ coef,
                                                           double* __restrict__ smc, double* __restrict__ loglik, int* __restrict__ n_iter,
                                                           int* __restrict__ status, int phase_a, double etarget, int may_restart) {
    // phase_a: the cold start on the partitions' leading rows -- a partition that ends there writes no outputs (a failed one goes back
    // to beta = 0): batch_restart_kernel then opens the full-row phase from the subsample MLEs
    const int k = blockIdx.x, j = threadIdx.x;
    if (!active[k]) return;
    BatchState s = st[k];
    const double ll = ll_in[k];
    double* bk = beta + (int64_t)k * p;
    double* pk = prev + (int64_t)k * p;
    double* sk = stepv + (int64_t)k * p;
    const double* dk = delta + (int64_t)k * p;
    int end = -1;                                  // >= 0: the partition's fit ends with this status
    bool stepped = false;
    ++s.evals;
    if (s.last_pass) end = DLSA_PART_NOT_CONVERGED;            // max_iter reached: H, loglik are those of the last iterate
    else if (!isfinite(ll)) end = DLSA_PART_NAN;
    else if (s.have_prev && ll < s.ll_prev - 1e-12 * fabs(s.ll_prev) && s.halvings < 30) {
        // the previous step overshot: halve it and evaluate again (not a new iteration)
        if (j < p) { const double h = 0.5 * sk[j]; sk[j] = h; bk[j] = pk[j] + h; }
        ++s.halvings;
        if (s.evals > 2 * max_iter + 64) end = DLSA_PART_NOT_CONVERGED;
    } else {
        const double dmax = stats[3 * k], bmax = stats[3 * k + 1], flag = stats[3 * k + 2];
        ++s.iters;
        // (gradient-only phase: a step beyond the iterate's own size means the pooled Hessian is no stand-in for this partition's --
        // not taken; the Newton phase goes on from here)
        if (phase_a == 2 && !(dmax <= fmax(1.0, bmax))) { end = DLSA_PART_OK; stepped = true; }
        else if (flag == 1.0) end = DLSA_PART_NOT_SPD;
        else if (flag == 2.0 || !isfinite(dmax)) end = DLSA_PART_NAN;
        else if (dmax <= tol * fmax(1.0, bmax)) end = DLSA_PART_OK;
        else {
            if (j < p) { const double b = bk[j], d = dk[j]; pk[j] = b; sk[j] = d; bk[j] = b + d; }
            s.ll_prev = ll; s.have_prev = 1; s.halvings = 0;
            if (s.iters >= max_iter) s.last_pass = 1;
            if (phase_a == 2) {
                // gradient-only phase: steps shrink by rho per pass, so ~rho dmax / (1 - rho) is left after this one -- near enough
                // for the Newton phase to end in three passes?  Then this partition is done here.
                const double rho = s.dprev > 0.0 ? dmax / s.dprev : 1.0;
                if (rho < 0.5 && rho * dmax / (1.0 - rho) <= etarget * fmax(1.0, bmax)) { end = DLSA_PART_OK; stepped = true; }
                s.dprev = dmax;
            }
        }
    }
    if (!phase_a && may_restart && !s.restarted && (end == DLSA_PART_NOT_SPD || end == DLSA_PART_NAN)) {
        // the start phases left this partition somewhere Newton's method cannot go on from: once more from beta = 0, as a call
        // without them would have started (a partition without a finite MLE fails again, and is reported)
        if (j < p) bk[j] = 0.0;
        s.have_prev = 0; s.halvings = 0; s.iters = 0; s.evals = 0; s.last_pass = 0; s.ll_prev = 0.0; s.restarted = 1;
        end = -1;
    }
    if (end >= 0 && phase_a) {
        if (j < p && !stepped) bk[j] = end == DLSA_PART_OK ? bk[j] + dk[j] : 0.0;       // (a start phase takes its last, small step too)
        s.status = end;
    } else if (end >= 0) {
        // outputs: coef, Sig_inv . coef (models.py:131); Sig_inv is the H the unpack kernel has just written in place
        const double* Hk = H + (int64_t)k * p * p;
        if (j < p) {
            coef[(int64_t)k * p + j] = bk[j];
            double acc = 0.0;
            for (int c = 0; c < p; ++c) acc = fma(Hk[(int64_t)j * p + c], bk[c], acc);
            smc[(int64_t)k * p + j] = acc;
        }
        if (j == 0) { loglik[k] = ll; n_iter[k] = s.iters; status[k] = end; }
        s.status = end;
    }
    __syncthreads();                               // every thread has read active[k] / the old beta before they change
    if (j == 0) {
        st[k] = s;
        if (end >= 0) active[k] = 0;
        else atomicAdd(n_live, 1);
    }
}

// between the subsample phase and the full-row phase: everybody is live again, the safeguard's memory is cleared (another objective),
// and the iteration count starts over -- max_iter and n_iter mean full-row iterations, as on the chained path (its it_sub is dropped too)
__global__ void batch_restart_kernel(int K, BatchState* __restrict__ st, int* __restrict__ active) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    BatchState s = st[k];
    s.have_prev = 0; s.halvings = 0; s.evals = 0; s.last_pass = 0; s.status = 0; s.ll_prev = 0.0; s.iters = 0; s.dprev = 0.0; s.restarted = 0;
    st[k] = s;
    active[k] = 1;
}

// pooled start: the groups' H, g, loglik (batch_unpack_kernel over the group table) summed in a fixed order into slot 0
__global__ __launch_bounds__(256) void batch_pool_sum_kernel(int G, int p, double* __restrict__ H, double* __restrict__ g, double* __restrict__ ll) {
    const int64_t pp = (int64_t)p * p, e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < pp) {
        // (eight loads in flight per trip, the additions in their order: as a plain loop every group waited for its own L2 round trip)
        double s = H[e];
        for (int q0 = 1; q0 < G; q0 += 8) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = q0 + u < G ? H[(int64_t)(q0 + u) * pp + e] : 0.0;
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (q0 + u < G) s += v[u];
        }
        H[e] = s;
    } else if (e - pp <= p) {
        const int j = (int)(e - pp);
        double s = j < p ? g[j] : ll[0];
        for (int q = 1; q < G; ++q) s += j < p ? g[(int64_t)q * p + j] : ll[q];
        if (j < p) g[j] = s;
        else ll[0] = s;
    }
}

// gradient-only iterations after the pooled start: partition k's Newton step with the POOLED Hessian in place of its own,
// delta = (rows_k / rows_pool  H_pool)^-1 g_k -- H_pool^-1 is the explicit inverse the pooled fit's last iteration left
__global__ __launch_bounds__(128) void batch_pool_step_kernel(int p, const double* __restrict__ Hinv, const double* __restrict__ scale,
                                                              const double* __restrict__ g, const double* __restrict__ beta,
                                                              const int* __restrict__ active, double* __restrict__ delta,
                                                              double* __restrict__ stats) {
    const int k = blockIdx.x, j = threadIdx.x;
    if (!active[k]) return;
    __shared__ double gs[128], red[2][2];
    if (j < p) gs[j] = g[(int64_t)k * p + j];
    __syncthreads();
    double d = 0.0, b = 0.0;
    if (j < p) {
        double acc = 0.0;
        for (int c = 0; c < p; ++c) acc = fma(Hinv[(int64_t)j * p + c], gs[c], acc);
        d = acc * scale[k];
        delta[(int64_t)k * p + j] = d;
        b = fabs(beta[(int64_t)k * p + j]);
    }
    double dm = isfinite(d) ? fabs(d) : INFINITY;
    for (int o = 32; o; o >>= 1) { dm = fmax(dm, __shfl_xor(dm, o)); b = fmax(b, __shfl_xor(b, o)); }
    if ((j & 63) == 0) { red[j >> 6][0] = dm; red[j >> 6][1] = b; }
    __syncthreads();
    if (j == 0) { stats[3 * k] = fmax(red[0][0], red[1][0]); stats[3 * k + 1] = fmax(red[0][1], red[1][1]); stats[3 * k + 2] = 0.0; }
}

// ... and its end: every partition starts the full-row phase from the pooled estimate
__global__ __launch_bounds__(256) void batch_pool_spread_kernel(int K, int p, double* __restrict__ beta) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (int64_t)K * p || e < p) return;
    beta[e] = beta[e % p];
}

// Is the lock-step path the faster one?  Measured constants: a fused pass streams ~4.5e9 rows/s at p = 100 (scaled by the width).
bool irls_batched_eligible(const double* X, int64_t ldx, const double* y, const int64_t* rows_host, int K, int p, int intercept,
                           int64_t row_step, double* est_ms) {
    if (est_ms) *est_ms = 0.0;                     // (stays 0 when forced)
    const char* e = knob("DLSA_IRLS_BATCHED");
    if (e && atoi(e) == 0) return false;
    const int pe = p + (intercept ? 1 : 0);
    // strided partitions (partition_id = i % K, models.py:33: rows first, first + step, ...): the row pitch is ldx * step, and a slab of
    // at least 2048 rows must stay inside the 32-bit DMA offsets
    if (row_step < 1 || K < 2 || !irls_pass_batched_shape_ok(X, ldx * row_step, y, p, intercept, ldx) || !chol_small_ok(pe)) return false;
    if ((double)(2048 + 8 * 32) * (double)ldx * (double)row_step * 8.0 >= 2.0e9) return false;
    int64_t total = 0, mn = INT64_MAX;
    for (int k = 0; k < K; ++k) { total += rows_host[k]; mn = std::min(mn, rows_host[k]); }
    if (mn < 1) return false;                      // empty partitions: the host-driven path writes the reference's zero block
    if (e && atoi(e) != 0) return true;            // forced (A/B runs, tests)
    // (round 5, bench/many_partitions.py: lock step = pooled start + ~3 gradient-only passes at half price + 3 Newton passes ~ 6 pass units
    // + ~10 iterations' launches; the chains are bound by ~0.23 ms per partition or by their ~4.6 pass units of row traffic)
    const double pass_s = (double)total * (pe / 100.0) * 2.3e-10;
    const double t_batch = 6.0 * pass_s + 3.0e-3, t_chain = std::max(K * 0.23e-3, 4.6 * pass_s);
    if (est_ms) *est_ms = t_batch * 1e3;
    return K >= 8 && t_batch < t_chain;
}

// labels of strided partitions, slab by slab, into one contiguous buffer (the DMA's y pieces want consecutive labels)
__global__ __launch_bounds__(256) void batch_gather_y_kernel(const double* __restrict__ y, const FusedSlab* __restrict__ slabs,
                                                             const int64_t* __restrict__ ysrc, int64_t step, double* __restrict__ ybuf) {
    const FusedSlab sd = slabs[blockIdx.x];
    const int64_t src0 = ysrc[blockIdx.x];
    for (int i = threadIdx.x; i < sd.nrows; i += blockDim.x) ybuf[sd.yoff + i] = y[src0 + (int64_t)i * step];
}

int irls_batched_fit(const double* X, int64_t ldx, const double* y, const int64_t* first_host, const int64_t* rows_host, int64_t row_step, int K,
                     int pdata, int intercept, double tol, int max_iter, double* coef, double* Sig_inv, double* Sig_invMcoef, int* n_iter_host, int* status_host,
                     double* loglik_host, hipStream_t stream) {
    const int p = pdata + (intercept ? 1 : 0);          // columns of beta / g / H (intercept first)
    const int PP = irls_pass_batched_pp(p), GP = irls_pass_batched_gp(p), ll_at = irls_pass_batched_ll_at(p);
    // ---- the slab table: partition k in nsl equal slabs of whole chunks (rows first + r step: the kernel's row pitch is ldx * step)
    const int64_t pitch = ldx * row_step;
    const int64_t fit_rows = (int64_t)(1.9e9 / ((double)pitch * 8.0)) / 32 * 32 - 8 * 32;        // what the 32-bit DMA offsets hold
    const int64_t slab_cap = std::max<int64_t>(2048, std::min<int64_t>(BATCH_SLAB_ROWS, fit_rows));
    const bool gather_y = row_step > 1;
    std::vector<FusedSlab> slabs;
    std::vector<int64_t> ysrc;                       // strided labels: index of the slab's first label in y
    std::vector<int> slab_begin((size_t)K + 1, 0);
    int64_t ytotal = 0;
    for (int k = 0; k < K; ++k) {
        const int64_t nk = rows_host[k];
        const int nsl = (int)std::max<int64_t>(1, (nk + slab_cap - 1) / slab_cap);
        const int64_t per = ((nk + nsl - 1) / nsl + 31) / 32 * 32;
        slab_begin[(size_t)k] = (int)slabs.size();
        for (int64_t r = 0; r < nk; r += per) {
            FusedSlab sd;
            sd.xoff = (first_host[k] + r * row_step) * ldx;
            sd.nrows = (int)std::min<int64_t>(per, nk - r); sd.part = k;
            sd.yoff = gather_y ? ytotal : first_host[k] + r;
            ysrc.push_back(first_host[k] + r * row_step);
            ytotal += (sd.nrows + 1) / 2 * 2;       // (even: the y pieces are 16-byte DMA loads from 8-byte aligned labels)
            DLSA_REQUIRE((double)(sd.nrows + 8 * 32) * (double)pitch * 8.0 < 2.0e9, "irls_fit (batched): a slab exceeds the 32-bit DMA offsets");
            slabs.push_back(sd);
        }
    }
    slab_begin[(size_t)K] = (int)slabs.size();
    const int nslab = (int)slabs.size();
    // ---- cold start: the first Newton iterations from beta = 0 run on the leading 1 / 8 of every partition's rows (they only have to
    // get near the MLE: 4-5 passes at an eighth of the cost), then the full-row iterations start from there (3-4 passes instead of 7).
    // Only when every partition's eighth still pins its MLE (>= 40 rows per coefficient, >= 2048 rows).
    std::vector<FusedSlab> slabsA, slabsP;
    std::vector<int> beginA((size_t)K + 1, 0);
    bool phase_a = true;
    // the leading `lead(k)` rows of every partition as a slab table (the labels of a partition's leading rows lead its gathered labels)
    auto leading_rows = [&](std::vector<FusedSlab>& tab, std::vector<int>* begin, auto lead) {
        for (int k = 0; k < K; ++k) {
            const int64_t nL = lead(k);
            const FusedSlab& f0 = slabs[(size_t)slab_begin[(size_t)k]];
            const int nsl = (int)std::max<int64_t>(1, (nL + slab_cap - 1) / slab_cap);
            const int64_t per = ((nL + nsl - 1) / nsl + 31) / 32 * 32;
            if (begin) (*begin)[(size_t)k] = (int)tab.size();
            for (int64_t r = 0; r < nL; r += per) {
                FusedSlab sd;
                sd.xoff = f0.xoff + r * pitch; sd.yoff = f0.yoff + r;
                sd.nrows = (int)std::min<int64_t>(per, nL - r); sd.part = k;
                tab.push_back(sd);
            }
        }
        if (begin) (*begin)[(size_t)K] = (int)tab.size();
    };
    int64_t mn = INT64_MAX, mx = 0;
    for (int k = 0; k < K; ++k) { mn = std::min(mn, rows_host[k]); mx = std::max(mx, rows_host[k]); }
    const int64_t need_rows = std::max<int64_t>(2048, 40 * (int64_t)p);
    {
        const char* e = knob("DLSA_IRLS_SUBSAMPLE");
        const int div = e ? atoi(e) : (mn / 8 / 32 * 32 >= need_rows ? 8 : 4);       // an eighth where that is enough rows, else a quarter
        if (div < 2 || mn / div / 32 * 32 < need_rows) phase_a = false;
        if (phase_a) leading_rows(slabsA, &beginA, [&](int k) { return rows_host[k] / div / 32 * 32; });
    }
    const int nslabA = (int)slabsA.size();
    // ---- pooled start (round 5): ONE fit on leading rows of ALL partitions together, and every partition's full-row iterations start
    // from that estimate.  Where rows are exchangeable across partitions (the reference shuffles / deals them out as i % K,
    // models.py:33) the pooled estimate is ~sqrt(p / rows_k) from partition k's own MLE -- nearer than the MLE of its own quarter --
    // once the pool holds 8 rows_max rows (its own error is then a third of that; and >= 1000 rows per coefficient), whatever K is.  With many partitions that
    // is a small fraction of the subsample phase's rows, and an iteration pays one inversion instead of K: 1000 x 2e4 x 100 four
    // pooled iterations of ~0.4 ms instead of four subsample iterations of 1.8 ms.  Partitions that differ from each other just take
    // the Newton iterations their distance asks for; a pooled fit that fails falls back to the subsample phase.
    const char* pool_env = knob("DLSA_IRLS_POOLED_START");
    bool pooled = !(pool_env && atoi(pool_env) == 0);
    const int G = std::min(K, 128);                  // the pooled table's slabs in G groups: the unpack stays parallel
    std::vector<int> beginP((size_t)G + 1, 0);
    if (pooled) {
        const int64_t want = std::max<int64_t>(8 * mx, 1000 * (int64_t)p), each = std::max<int64_t>(256, ((want + K - 1) / K + 31) / 32 * 32);
        auto lead = [&](int k) { return std::max<int64_t>(32, std::min<int64_t>(each, rows_host[k] / 4 / 32 * 32)); };
        int64_t rows_pool = 0;
        for (int k = 0; k < K; ++k) rows_pool += std::min<int64_t>(lead(k), rows_host[k]);
        // (a pool of fewer rows than this is no better a start than a partition's own rows, and its Hessian no stand-in)
        if (rows_pool < std::max<int64_t>(2 * mx, 200 * (int64_t)p)) pooled = false;      // (10 x 1e6 x 100 with 2.5 rows_max: 25.7 -> 21.6 ms)
        if (pooled) {
            leading_rows(slabsP, nullptr, [&](int k) { return std::min<int64_t>(lead(k), rows_host[k]); });
            int at = 0;
            const int nP = (int)slabsP.size();
            for (int q = 0; q < G; ++q) {
                beginP[(size_t)q] = at;
                const int k1 = (int)((int64_t)(q + 1) * K / G);
                while (at < nP && slabsP[(size_t)at].part < k1) slabsP[(size_t)at++].part = q;
            }
            beginP[(size_t)G] = at;
        }
    }
    const int nslabP = (int)slabsP.size();
    std::vector<double> pool_scale;                  // rows_pool / rows_k: the pooled Hessian stands in for rows_k / rows_pool of itself
    if (pooled) {
        int64_t rows_pool = 0;
        for (const FusedSlab& sd : slabsP) rows_pool += sd.nrows;
        pool_scale.resize((size_t)K);
        for (int k = 0; k < K; ++k) pool_scale[(size_t)k] = (double)rows_pool / (double)rows_host[k];
    }

    // ---- scratch: ONE block from the stream-ordered pool, freed before the call returns
    const size_t pb = (size_t)K * p * sizeof(double);
    size_t off = 0;
    auto carve = [&](size_t bytes) { const size_t o = off; off = align_up(off + bytes, 256); return o; };
    const size_t o_slabs = carve((size_t)nslab * sizeof(FusedSlab)), o_begin = carve(((size_t)K + 1) * sizeof(int)),
                 o_active = carve((size_t)K * sizeof(int)), o_live = carve(256), o_iter = carve((size_t)K * sizeof(int)),
                 o_status = carve((size_t)K * sizeof(int)), o_state = carve((size_t)K * sizeof(BatchState)),
                 o_partial = carve((size_t)nslab * PP * PP * sizeof(double)), o_gpart = carve((size_t)nslab * GP * sizeof(double)),
                 o_g = carve(pb), o_ll = carve((size_t)K * sizeof(double)), o_llout = carve((size_t)K * sizeof(double)), o_delta = carve(pb),
                 o_stats = carve((size_t)K * 3 * sizeof(double)), o_beta = carve(pb), o_prev = carve(pb), o_step = carve(pb),
                 o_hinv = carve((size_t)K * p * p * sizeof(double)), o_clk = carve(256),
                 o_ysrc = carve(gather_y ? (size_t)nslab * sizeof(int64_t) : 0), o_ybuf = carve(gather_y ? (size_t)(ytotal + 64) * sizeof(double) : 0),
                 o_slabsA = carve((size_t)nslabA * sizeof(FusedSlab)), o_beginA = carve(((size_t)K + 1) * sizeof(int)),
                 o_slabsP = carve((size_t)nslabP * sizeof(FusedSlab)), o_beginP = carve(((size_t)G + 1) * sizeof(int)),
                 o_scale = carve((size_t)K * sizeof(double));
    char* pool = nullptr;
    DLSA_HIP_CHECK(hipMallocAsync((void**)&pool, off, stream));
    FusedSlab* d_slabs = (FusedSlab*)(pool + o_slabs);
    int *d_begin = (int*)(pool + o_begin), *d_active = (int*)(pool + o_active), *d_live = (int*)(pool + o_live), *d_iter = (int*)(pool + o_iter),
        *d_status = (int*)(pool + o_status);
    BatchState* d_state = (BatchState*)(pool + o_state);
    double *d_partial = (double*)(pool + o_partial), *d_gpart = (double*)(pool + o_gpart), *d_g = (double*)(pool + o_g), *d_ll = (double*)(pool + o_ll),
           *d_llout = (double*)(pool + o_llout), *d_delta = (double*)(pool + o_delta), *d_stats = (double*)(pool + o_stats),
           *d_beta = (double*)(pool + o_beta), *d_prev = (double*)(pool + o_prev), *d_step = (double*)(pool + o_step), *d_hinv = (double*)(pool + o_hinv);
    unsigned long long* d_clk = (unsigned long long*)(pool + o_clk);
    int64_t* d_ysrc = (int64_t*)(pool + o_ysrc);
    double* d_ybuf = (double*)(pool + o_ybuf);
    const double* ylab = gather_y ? d_ybuf : y;
    FusedSlab* d_slabsA = (FusedSlab*)(pool + o_slabsA);
    int* d_beginA = (int*)(pool + o_beginA);
    FusedSlab* d_slabsP = (FusedSlab*)(pool + o_slabsP);
    int* d_beginP = (int*)(pool + o_beginP);
    double* d_scale = (double*)(pool + o_scale);
    auto release = [&]() { (void)hipFreeAsync(pool, stream); };
    int rc = DLSA_OK;
    auto fail = [&](int code) { release(); return code; };
#define DLSA_BATCH_CHECK(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); return fail(DLSA_ERR_HIP); } } while (0)
    DLSA_BATCH_CHECK(hipMemcpyAsync(d_slabs, slabs.data(), (size_t)nslab * sizeof(FusedSlab), hipMemcpyHostToDevice, stream));
    DLSA_BATCH_CHECK(hipMemcpyAsync(d_begin, slab_begin.data(), ((size_t)K + 1) * sizeof(int), hipMemcpyHostToDevice, stream));
    if (gather_y) {
        DLSA_BATCH_CHECK(hipMemcpyAsync(d_ysrc, ysrc.data(), (size_t)nslab * sizeof(int64_t), hipMemcpyHostToDevice, stream));
        hipLaunchKernelGGL(batch_gather_y_kernel, dim3(nslab), dim3(256), 0, stream, y, (const FusedSlab*)d_slabs, (const int64_t*)d_ysrc, row_step, d_ybuf);
    }
    if (phase_a) {
        DLSA_BATCH_CHECK(hipMemcpyAsync(d_slabsA, slabsA.data(), (size_t)nslabA * sizeof(FusedSlab), hipMemcpyHostToDevice, stream));
        DLSA_BATCH_CHECK(hipMemcpyAsync(d_beginA, beginA.data(), ((size_t)K + 1) * sizeof(int), hipMemcpyHostToDevice, stream));
    }
    if (pooled) {
        DLSA_BATCH_CHECK(hipMemcpyAsync(d_slabsP, slabsP.data(), (size_t)nslabP * sizeof(FusedSlab), hipMemcpyHostToDevice, stream));
        DLSA_BATCH_CHECK(hipMemcpyAsync(d_beginP, beginP.data(), ((size_t)G + 1) * sizeof(int), hipMemcpyHostToDevice, stream));
        DLSA_BATCH_CHECK(hipMemcpyAsync(d_scale, pool_scale.data(), (size_t)K * sizeof(double), hipMemcpyHostToDevice, stream));
    }
    {
        std::vector<int> ones((size_t)K, 1);
        DLSA_BATCH_CHECK(hipMemcpyAsync(d_active, ones.data(), (size_t)K * sizeof(int), hipMemcpyHostToDevice, stream));
        DLSA_BATCH_CHECK(hipStreamSynchronize(stream));        // (the host vectors above go out of scope / are reused)
    }
    DLSA_BATCH_CHECK(hipMemsetAsync(d_state, 0, (size_t)K * sizeof(BatchState), stream));
    DLSA_BATCH_CHECK(hipMemsetAsync(d_beta, 0, pb, stream));
    DLSA_BATCH_CHECK(hipMemsetAsync(d_prev, 0, pb, stream));
    DLSA_BATCH_CHECK(hipMemsetAsync(d_step, 0, pb, stream));

    const int cap = 2 * max_iter + 66;
    bool used_start = false;                        // the Newton phase does not begin at beta = 0
    const char* trace_env = knob("DLSA_IRLS_TRACE");
    const bool trace = trace_env && atoi(trace_env) != 0;
    int live = K;
    // one phase: passes over the given slab table until no partition is live
    // (groups > 0: the pooled fit -- the table's parts are `groups` groups which all read beta row 0, summed into ONE problem)
    // (grad_passes > 0: that many gradient-only passes -- no H, the step from the pooled Hessian's inverse in d_hinv slot 0)
    auto run_phase = [&](const FusedSlab* tab, int ntab, const int* begin, double ptol, int in_a, int groups, int grad_passes) -> int {
        const int nprob = groups ? 1 : K, nunpack = groups ? groups : K;
        // (the start phases are the driver's own: the caller's max_iter bounds the full-row Newton iterations only)
        // -- and a start phase that has not settled in 12 iterations (a healthy fit from zero takes 4-6 to a step of 3e-2) is given up: its
        // relative step test is fooled by an estimate that grows without bound (separable rows: steps of 0.5 on |beta| -> 50)
        const int iter_cap = in_a ? 12 : max_iter, pass_cap = grad_passes ? grad_passes : in_a ? 40 : cap;
        live = nprob;
        for (int it = 0; it < pass_cap && live > 0; ++it) {
            int r = irls_pass_batched_launch(X, pitch, ylab, d_beta, groups ? 0 : p, pdata, intercept, tab, ntab, d_active, d_partial, d_gpart, d_clk, stream,
                                             grad_passes ? 0 : 1);
            if (r) return r;
            hipLaunchKernelGGL(batch_unpack_kernel, dim3(nunpack), dim3(256), 0, stream, (const double*)d_partial, (const double*)d_gpart,
                               begin, (const int*)d_active, PP, GP, ll_at, p, intercept ? 1 : 0, grad_passes ? (double*)nullptr : Sig_inv, d_g, d_ll);
            if (groups > 1) hipLaunchKernelGGL(batch_pool_sum_kernel, dim3((unsigned)((p * p + p + 1 + 255) / 256)), dim3(256), 0, stream, groups, p, Sig_inv, d_g, d_ll);
            if (grad_passes)
                hipLaunchKernelGGL(batch_pool_step_kernel, dim3(K), dim3(128), 0, stream, p, (const double*)d_hinv, (const double*)d_scale, (const double*)d_g,
                                   (const double*)d_beta, (const int*)d_active, d_delta, d_stats);
            else
                r = launch_chol_small_batched(nprob, Sig_inv, p, (int64_t)p * p, p, d_g, d_beta, p, d_hinv, (int64_t)p * p, d_delta, d_stats, 3, d_active, stream);
            if (r) return r;
            if (hipMemsetAsync(d_live, 0, sizeof(int), stream) != hipSuccess) return DLSA_ERR_HIP;
            hipLaunchKernelGGL(batch_update_kernel, dim3(nprob), dim3(128), 0, stream, p, ptol, iter_cap, (const double*)Sig_inv, (const double*)d_ll,
                               (const double*)d_delta, (const double*)d_stats, d_beta, d_prev, d_step, d_state, d_active, d_live, coef,
                               Sig_invMcoef, d_llout, d_iter, d_status, grad_passes ? 2 : in_a, ptol, used_start ? 1 : 0);
            if (hipGetLastError() != hipSuccess || hipMemcpyAsync(&live, d_live, sizeof(int), hipMemcpyDeviceToHost, stream) != hipSuccess ||
                hipStreamSynchronize(stream) != hipSuccess) {
                set_error("irls_fit (batched): a launch or the read-back of the live count failed");
                return DLSA_ERR_HIP;
            }
            if (trace) {       // (dlsa_irls_options.trace: the iteration's relative steps over the partitions that took part in it)
                std::vector<double> hs((size_t)nprob * 3);
                if (hipMemcpyAsync(hs.data(), d_stats, hs.size() * sizeof(double), hipMemcpyDeviceToHost, stream) != hipSuccess ||
                    hipStreamSynchronize(stream) != hipSuccess) return DLSA_ERR_HIP;
                std::vector<double> rel;
                for (int k = 0; k < nprob; ++k) rel.push_back(hs[3 * (size_t)k] / std::max(1.0, hs[3 * (size_t)k + 1]));
                std::sort(rel.begin(), rel.end());
                fprintf(stderr, "[dlsa lock step] %s pass %d: live after it %d of %d; relative step (all partitions' last) median %.2e max %.2e\n",
                        groups ? "pooled" : grad_passes ? "gradient-only" : in_a ? "subsample" : "Newton", it, live, nprob, rel[rel.size() / 2], rel.back());
            }
        }
        return DLSA_OK;
    };
    bool started = false;
    if (pooled) {
        // the groups' slabs stay in the pass while active[0 .. G) are set: only the pooled problem's own flag (slot 0) changes
        rc = run_phase(d_slabsP, nslabP, d_beginP, std::max(tol, 3e-2), 1, G, 0);
        if (rc) return fail(rc);
        BatchState sp;
        DLSA_BATCH_CHECK(hipMemcpyAsync(&sp, d_state, sizeof(BatchState), hipMemcpyDeviceToHost, stream));
        DLSA_BATCH_CHECK(hipStreamSynchronize(stream));
        started = live == 0 && sp.status == DLSA_PART_OK;
        used_start = started;
        if (started) hipLaunchKernelGGL(batch_pool_spread_kernel, dim3((unsigned)(((int64_t)K * p + 255) / 256)), dim3(256), 0, stream, K, p, d_beta);
        else DLSA_BATCH_CHECK(hipMemsetAsync(d_beta, 0, pb, stream));
        hipLaunchKernelGGL(batch_restart_kernel, dim3((K + 255) / 256), dim3(256), 0, stream, K, d_state, d_active);
        // ---- gradient-only iterations: passes over all rows that cost half a fused pass (no H: HBM-bound), steps with the pooled
        // Hessian (scaled to the partition's row count) in place of the partition's own.  They contract the distance to the partition's
        // MLE by the relative distance of its Hessian from the pooled one, ~sqrt(p / rows_k) -- 0.17 -> 0.015 -> 1.5e-3 at 2e4 x 100 --
        // and a partition leaves this phase once three Newton passes will do from where it is (steps e, c e^2, c^3 e^4 <= tol with
        // c ~ 1/2): G G G F F F (tol 1e-13; G G for 1e-10) for F F F F F.  bench/lockstep_start_study.py, bench/grad_passes_ab.py.
        // The safeguard is the Newton phase's own (a step that lowers the log-likelihood is halved).
        const char* ge = knob("DLSA_IRLS_GRAD_PASSES");
        const int gpasses = ge ? atoi(ge) : 4;
        const double etarget = std::min(3e-3, std::max(1e-4, std::pow(0.8 * tol, 0.25)));
        if (started && gpasses > 0) {
            rc = run_phase(d_slabs, nslab, d_begin, etarget, 1, 0, gpasses);
            if (rc) return fail(rc);
            hipLaunchKernelGGL(batch_restart_kernel, dim3((K + 255) / 256), dim3(256), 0, stream, K, d_state, d_active);
        }
    }
    if (phase_a && !started) {
        used_start = true;
        // (the subsample's MLE is ~2 sqrt(p / rows) away from the partition's own whatever happens here: a step of 3e-2 is close enough --
        // round 5, same box: 1000 x 2e4 x 100 39.6 -> 37.0 ms, 200 x 1e5 x 100 31.6 -> 30.7, the full-row iterations unchanged at 5)
        rc = run_phase(d_slabsA, nslabA, d_beginA, std::max(tol, 3e-2), 1, 0, 0);
        if (rc) return fail(rc);
        // (a partition still live after the cap restarts like a failed one would: from where it is)
        hipLaunchKernelGGL(batch_restart_kernel, dim3((K + 255) / 256), dim3(256), 0, stream, K, d_state, d_active);
    }
    rc = run_phase(d_slabs, nslab, d_begin, tol, 0, 0, 0);
    if (rc) return fail(rc);
    if (live > 0) { set_error("irls_fit (batched): %d partitions still live after %d passes", live, cap); return fail(DLSA_ERR_INVALID); }
#if FP_TIMELINE
    // (experiment builds, bench/lockstep_timeline.py: the last pass's per-workgroup time stamps ride in the free slots of the g partials)
    if (const char* dump = getenv("DLSA_TL_DUMP")) {
        std::vector<double> hostg((size_t)nslab * GP);
        DLSA_BATCH_CHECK(hipMemcpyAsync(hostg.data(), d_gpart, hostg.size() * sizeof(double), hipMemcpyDeviceToHost, stream));
        DLSA_BATCH_CHECK(hipStreamSynchronize(stream));
        if (FILE* fh = fopen(dump, "wb")) {
            const int hdr[4] = {nslab, GP, p, 0};
            fwrite(hdr, sizeof(int), 4, fh); fwrite(hostg.data(), sizeof(double), hostg.size(), fh); fclose(fh);
        }
    }
#endif
    if (n_iter_host) DLSA_BATCH_CHECK(hipMemcpyAsync(n_iter_host, d_iter, (size_t)K * sizeof(int), hipMemcpyDeviceToHost, stream));
    if (status_host) DLSA_BATCH_CHECK(hipMemcpyAsync(status_host, d_status, (size_t)K * sizeof(int), hipMemcpyDeviceToHost, stream));
    if (loglik_host) DLSA_BATCH_CHECK(hipMemcpyAsync(loglik_host, d_llout, (size_t)K * sizeof(double), hipMemcpyDeviceToHost, stream));
    DLSA_BATCH_CHECK(hipStreamSynchronize(stream));
    release();
    return DLSA_OK;
#undef DLSA_BATCH_CHECK
}

}  // namespace dlsa
