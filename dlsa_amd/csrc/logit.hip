// Fused logit pass over the rows (HBM-bound; one read of X):
//   eta_i = x_i . beta,  mu_i = sigmoid(eta_i),  w_i = mu_i (1 - mu_i),
//   g = X'(y - mu),  loglik = sum_i y_i log mu_i + (1 - y_i) log(1 - mu_i).
// Reference call sites: dlsa/models.py:113-114 (the inner products inside sklearn's newton-cg
// fit and predict_proba), :130 (the p(1-p) weights), :217-222 (log-likelihood).
//
// Layout: a wave owns RB consecutive rows at a time.  Lane l holds columns 128c + 2l + {0,1}
// (c < NC), i.e. one 16-byte load per lane per 128-column chunk = 1 KiB coalesced per wave
// instruction.  The RB row dot products are reduced with ONE merged butterfly (the first
// log2(RB) exchange steps halve the number of live rows), the transcendental part runs once
// per RB rows instead of once per row, and the residuals are broadcast back through SGPRs for
// the rank-1 update of g, which stays in registers for the whole kernel.
#include "common.h"
#include <algorithm>

#ifndef DLSA_LOGIT_NT
#define DLSA_LOGIT_NT 1             // non-temporal loads of the rows (each is read once per pass): 6.1 -> 6.55 TB/s at p = 500, 5.8 -> 6.55 at p = 256 (same box); 0 = default cache policy
#endif
typedef double dlsa_d2v __attribute__((ext_vector_type(2)));
typedef float dlsa_f4v __attribute__((ext_vector_type(4)));
#if DLSA_LOGIT_NT
#define DLSA_STREAM_LOAD(ptr) __builtin_nontemporal_load(ptr)
#else
#define DLSA_STREAM_LOAD(ptr) (*(ptr))
#endif
static __device__ __forceinline__ double2 dlsa_stream_ld2(const double* ptr) {
    const dlsa_d2v t = DLSA_STREAM_LOAD(reinterpret_cast<const dlsa_d2v*>(ptr));
    double2 r; r.x = t.x; r.y = t.y; return r;
}
static __device__ __forceinline__ float4 dlsa_stream_ld4f(const float* ptr) {
    const dlsa_f4v t = DLSA_STREAM_LOAD(reinterpret_cast<const dlsa_f4v*>(ptr));
    float4 r; r.x = t.x; r.y = t.y; r.z = t.z; r.w = t.w; return r;
}
#ifndef DLSA_IMG_NT
#define DLSA_IMG_NT 1            // non-temporal stores of the bf16 chunk images (same-box A/B at 1e6 x 500: 905 -> 875 us; 0 = default policy)
#endif
#define DLSA_LOGIT_LD2(ptr) dlsa_stream_ld2(ptr)
#define DLSA_LOGIT_LD4F(ptr) dlsa_stream_ld4f(ptr)

namespace dlsa {

constexpr int LOGIT_THREADS = 256;
constexpr int LOGIT_WAVES = LOGIT_THREADS / 64;
constexpr int LOGIT_MAX_BLOCKS = 2048;

#include "rowdot.h"       // exch_add, merged_reduce, row_of_lane, lane_of_row, rep_mask, read_lane_f64
#include "logistic.h"      // rcp_newton, exp_neg, logistic_terms: shared with the fused Newton pass (irls_pass.hip)

struct LogitArgs {
    const double* X;
    const double* y;
    const double* beta;
    double* w_out;     // nullable
    double* gpart;     // [nblocks][NC*128]
    double* llpart;    // [nblocks]
    int64_t ldx;
    int64_t n;
    int p;
    // implicit intercept (the ones column of models.py:121-122 is never materialised): beta0 = its coefficient (nullable =
    // no intercept); s0part [nblocks] = per-block sum of the residuals (logit) / of v (xtv), i.e. the ones column's entry of
    // X'(y - mu) resp. X'v.  For xtv, a null y stands for the all-ones vector.
    const double* beta0;
    double* s0part;
    // BORDER form: the intercept's border of the Hessian [1 | X]' W [1 | X] (models.py:121-130) from the weights this pass computes
    // anyway -- hpart [nblocks][NC*128] = X'w per block, swpart [nblocks] = sum w -- so that the Gram that follows needs no pass of its own
    double* hpart = nullptr;
    double* swpart = nullptr;
    // IMG form (irls_wide.hip): the pass also leaves S = sqrt(w) [X | 1] rounded to bf16 as chunk images of 16 rows in the layout the
    // bf16 Gram kernel's MFMA fragments read -- [chunk][k-group of 8 rows][column block of 32][half of 4 rows][32 columns][4 rows x 2 B]
    // (img_nb column blocks; img_ones: the design's column p is the intercept's ones column)
    unsigned* img = nullptr;
    int img_nb = 0;
    int img_ones = 0;
};

// IMG form: the batch's RB rows (RB = 4 or 8) scaled by sqrt(w) and rounded to bf16, a lane's two columns of a chunk half (4 rows x 2 B
// each) as 16 contiguous bytes.  Columns past p (clamped loads) are masked; column p is the ones column when img_ones.
typedef __bf16 logit_bf16x2 __attribute__((ext_vector_type(2)));
typedef float logit_f2v __attribute__((ext_vector_type(2)));
typedef unsigned logit_u4v __attribute__((ext_vector_type(4)));
#ifndef DLSA_IMG_F16
#define DLSA_IMG_F16 0
#endif
typedef _Float16 logit_f16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned logit_pack_bf16(float lo, float hi) {
    const logit_f2v v = {lo, hi};
#if DLSA_IMG_F16
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, logit_f16x2));
#else
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, logit_bf16x2));
#endif
}
template <int RB, int I>
__device__ __forceinline__ void logit_bcast_rows(float swf, float (&sw)[RB]) {
    if constexpr (I < RB) {
        sw[I] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, swf), lane_of_row<RB>(I)));
        logit_bcast_rows<RB, I + 1>(swf, sw);
    }
}
template <int RB, int NC, class Args>
__device__ __forceinline__ void logit_image(const Args& a, int64_t bt, int lane, float swf, const double2 (&x)[RB][NC]) {
    static_assert(RB == 4 || RB == 8, "image: a batch is one or two halves of a k-group");
    float sw[RB];
    logit_bcast_rows<RB, 0>(swf, sw);
    const int64_t row0 = bt * RB;
    const int64_t chunk = row0 >> 4;
    const int sub = (int)(row0 & 15) >> 2;                 // first 4-row piece of the batch within its chunk: k-group sub >> 1, half sub & 1
    unsigned* base = a.img + chunk * ((int64_t)a.img_nb * 256);        // (img_nb * 1024 bytes per chunk)
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int col = c * 128 + 2 * lane, blk = 4 * c + (lane >> 4);
        if (blk < a.img_nb) {
            const bool in0 = col < a.p, in1 = col + 1 < a.p;
            const float o0 = (a.img_ones && col == a.p) ? 1.0f : 0.0f, o1 = (a.img_ones && col + 1 == a.p) ? 1.0f : 0.0f;
#pragma unroll
            for (int h = 0; h < RB / 4; ++h) {
                float f0[4], f1[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    f0[i] = (in0 ? (float)x[4 * h + i][c].x : o0) * sw[4 * h + i];
                    f1[i] = (in1 ? (float)x[4 * h + i][c].y : o1) * sw[4 * h + i];
                }
                const int piece = sub + h;
                const logit_u4v v = {logit_pack_bf16(f0[0], f0[1]), logit_pack_bf16(f0[2], f0[3]), logit_pack_bf16(f1[0], f1[1]), logit_pack_bf16(f1[2], f1[3])};
#if DLSA_IMG_NT
                __builtin_nontemporal_store(v, reinterpret_cast<logit_u4v*>(base + ((((piece >> 1) * a.img_nb + blk) * 2 + (piece & 1)) * 64 + (lane & 15) * 4)));
#else
                *reinterpret_cast<logit_u4v*>(base + ((((piece >> 1) * a.img_nb + blk) * 2 + (piece & 1)) * 64 + (lane & 15) * 4)) = v;
#endif
            }
        }
    }
}

template <int NC, int RB, bool VEC, bool BORDER = false, bool IMG = false>
__global__ __launch_bounds__(LOGIT_THREADS) void logit_kernel(LogitArgs a) {
    __shared__ double red[NC * 128 + 2];
    __shared__ double redh[BORDER ? NC * 128 + 1 : 1];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const double b0 = a.beta0 ? *a.beta0 : 0.0;
    double s0 = 0.0;

    double2 b[NC], g[NC], h[BORDER ? NC : 1];
    double sw = 0.0;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int col = c * 128 + 2 * lane;
        b[c].x = col < a.p ? a.beta[col] : 0.0;
        b[c].y = col + 1 < a.p ? a.beta[col + 1] : 0.0;
        g[c].x = 0.0; g[c].y = 0.0;
        if constexpr (BORDER) { h[c].x = 0.0; h[c].y = 0.0; }
    }
    double ll = 0.0;
    const int myrow = row_of_lane<RB>(lane);
    const bool rep = (lane & rep_mask<RB>()) == 0;

    // (IMG: whole chunks of 16 rows -- a batch past n writes zeros into its share of the last chunk's image)
    const int64_t nbatch = IMG ? (a.n + 15) / 16 * (16 / RB) : (a.n + RB - 1) / RB;
    const int64_t stride = (int64_t)gridDim.x * LOGIT_WAVES;

    // (the label of the lane's own row travels with the batch: a load issued later would make its s_waitcnt
    //  drain the prefetch of the next batch as well -- vmcnt retires in order)
    auto load_batch = [&](int64_t bt, double2 (&x)[RB][NC], double& yv) {
        // Branch-free: rows past n and columns past p are CLAMPED to a valid address and zeroed by a select
        // afterwards, so every load is unconditional straight-line code.  (With the loads inside per-row / per-column
        // branches hipcc can no longer count them and falls back to s_waitcnt vmcnt(0), which drains the prefetch.)
        const int64_t row0 = bt * RB;
        const int64_t ry = min(row0 + myrow, a.n - 1);
        const double ytmp = a.y[ry];
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            const int64_t r = min(row0 + i, a.n - 1);
            const double* rowp = a.X + r * a.ldx;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const int col = c * 128 + 2 * lane;
                const int c0 = col < a.p ? col : 0;                   // clamped first column
                if (VEC) {                                            // VEC implies p even: a pair never straddles p
                    x[i][c] = DLSA_LOGIT_LD2(rowp + c0);
                } else {
                    x[i][c].x = rowp[c0];
                    x[i][c].y = rowp[col + 1 < a.p ? col + 1 : 0];
                }
            }
        }
        yv = ytmp;      // masked in process(): touching the loaded values here would wait for them right away
    };
    auto process = [&](int64_t bt, const double2 (&xraw)[RB][NC], const double yraw) {
        const int64_t row0 = bt * RB;
        // No masking of x needed: a clamped column meets beta = 0 in the dot product and only pollutes entries
        // of g beyond p, which nobody reads; a clamped row (past n) gets residual 0 below.
        const double2 (&x)[RB][NC] = xraw;
        const double yv = (row0 + myrow < a.n) ? yraw : 0.0;
        double dot[RB];
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            double s = 0.0;
#pragma unroll
            for (int c = 0; c < NC; ++c) s = fma(x[i][c].x, b[c].x, fma(x[i][c].y, b[c].y, s));
            dot[i] = s;
        }
        const double eta = merged_reduce<RB>(dot, lane) + b0;
        const int64_t r = row0 + myrow;
        const bool valid = r < a.n;
        // e = exp(-|eta|);  mu = sigmoid(eta);  w = mu(1-mu) = e/(1+e)^2
        double mu, wgt, sp;
        logistic_terms<true>(eta, mu, wgt, sp);
        const double resid = valid ? (yv - mu) : 0.0;
        if (valid && rep) {
            if (a.w_out) a.w_out[r] = wgt;
            // y log mu + (1-y) log(1-mu) = y*eta - softplus(eta)
            ll += yv * eta - sp;
            s0 += resid;
        }
        rank1_update<RB, NC, 0>(resid, x, g);
        if constexpr (IMG) logit_image<RB, NC>(a, bt, lane, valid ? __builtin_sqrtf((float)wgt) : 0.0f, x);
        if constexpr (BORDER) {                 // X'w and sum w of the same rows (a clamped row past n weighs nothing)
            const double wv = valid ? wgt : 0.0;
            if (rep) sw += wv;
            rank1_update<RB, NC, 0>(wv, x, h);
        }
    };

    int64_t bt = (int64_t)blockIdx.x * LOGIT_WAVES + wave;
    if constexpr (NC == 1) {
        // Narrow rows (p <= 128; at NC = 2 the second register set drops the kernel to one wave per SIMD: slower): a batch is only RB*8p bytes, so the serial chain load -> butterfly ->
        // transcendentals -> rank-1 update leaves HBM idle at the two waves per SIMD this kernel gets.  The next
        // batch is therefore loaded into a second register set before the current one is processed.
        double2 xa[RB][NC], xb[RB][NC];
        double ya = 0.0, yb = 0.0;
        // the prefetch is UNCONDITIONAL (a batch past the end re-reads row n-1 and is never processed): a load
        // inside "if (next < nbatch)" makes the wait counts conservative again
        if (a.n > 0) {
            load_batch(bt, xa, ya);
            for (; bt < nbatch; bt += 2 * stride) {
                const int64_t b1 = bt + stride, b2 = bt + 2 * stride;
                load_batch(b1, xb, yb);
                process(bt, xa, ya);
                load_batch(b2, xa, ya);
                if (b1 < nbatch) process(b1, xb, yb);
            }
        }
    } else {
        for (; bt < nbatch; bt += stride) {
            double2 x[RB][NC];
            double yv;
            load_batch(bt, x, yv);
            process(bt, x, yv);
        }
    }

    // block reduction of g and loglik: waves add into LDS one after another (fixed order)
    for (int m = 32; m >= 1; m >>= 1) { ll += __shfl_xor(ll, m, 64); s0 += __shfl_xor(s0, m, 64); }
    if constexpr (BORDER)
        for (int m = 32; m >= 1; m >>= 1) sw += __shfl_xor(sw, m, 64);
    for (int wv = 0; wv < LOGIT_WAVES; ++wv) {
        if (wave == wv) {
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                double* dst = red + c * 128 + 2 * lane;
                if (wv == 0) { dst[0] = g[c].x; dst[1] = g[c].y; }
                else { dst[0] += g[c].x; dst[1] += g[c].y; }
                if constexpr (BORDER) {
                    double* dh = redh + c * 128 + 2 * lane;
                    if (wv == 0) { dh[0] = h[c].x; dh[1] = h[c].y; }
                    else { dh[0] += h[c].x; dh[1] += h[c].y; }
                }
            }
            if constexpr (BORDER) { if (lane == 0) { if (wv == 0) redh[NC * 128] = sw; else redh[NC * 128] += sw; } }
            if (lane == 0) {
                if (wv == 0) { red[NC * 128] = ll; red[NC * 128 + 1] = s0; }
                else { red[NC * 128] += ll; red[NC * 128 + 1] += s0; }
            }
        }
        __syncthreads();
    }
    double* gp = a.gpart + (int64_t)blockIdx.x * (NC * 128);
    for (int col = tid; col < NC * 128; col += LOGIT_THREADS) gp[col] = red[col];
    if (tid == 0) { a.llpart[blockIdx.x] = red[NC * 128]; if (a.s0part) a.s0part[blockIdx.x] = red[NC * 128 + 1]; }
    if constexpr (BORDER) {
        double* hp = a.hpart + (int64_t)blockIdx.x * (NC * 128);
        for (int col = tid; col < NC * 128; col += LOGIT_THREADS) hp[col] = redh[col];
        if (tid == 0) a.swpart[blockIdx.x] = redh[NC * 128];
    }
}

// g[col] = sum_b gpart[b][col], loglik = sum_b llpart[b].  One workgroup per 16 columns (32 workgroups at p = 500:
// a 1e6-row partition pays this once per Newton iteration, so it is latency that counts); 64 row groups sum
// interleaved block ranges with four loads in flight each and are combined through LDS in a fixed order
// (deterministic).  The last workgroup also sums the log-likelihood partials.
constexpr int FINISH_COLS = 16, FINISH_GROUPS = 64;
__global__ __launch_bounds__(FINISH_COLS * FINISH_GROUPS) void logit_finish_kernel(const double* __restrict__ gpart,
                                                                                    const double* __restrict__ llpart,
                                                                                    int nblocks, int pitch, int p,
                                                                                    double* __restrict__ g,
                                                                                    double* __restrict__ loglik,
                                                                                    const double* __restrict__ s0part,
                                                                                    double* __restrict__ s0) {
    __shared__ double red[FINISH_GROUPS][FINISH_COLS + 1];
    const int cx = threadIdx.x % FINISH_COLS, ry = threadIdx.x / FINISH_COLS;
    const int col = blockIdx.x * FINISH_COLS + cx;
    if (g && blockIdx.x * FINISH_COLS < p) {
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        if (col < p) {
            const double* src = gpart + col;
            int b = ry;
            for (; b + 3 * FINISH_GROUPS < nblocks; b += 4 * FINISH_GROUPS) {
                s0 += src[(int64_t)b * pitch];
                s1 += src[(int64_t)(b + FINISH_GROUPS) * pitch];
                s2 += src[(int64_t)(b + 2 * FINISH_GROUPS) * pitch];
                s3 += src[(int64_t)(b + 3 * FINISH_GROUPS) * pitch];
            }
            for (; b < nblocks; b += FINISH_GROUPS) s0 += src[(int64_t)b * pitch];
        }
        red[ry][cx] = (s0 + s1) + (s2 + s3);
        __syncthreads();
        if (ry == 0 && col < p) {
            double t = red[0][cx];
            for (int k = 1; k < FINISH_GROUPS; ++k) t += red[k][cx];
            g[col] = t;
        }
    }
    if (blockIdx.x == gridDim.x - 1) {
        // the last workgroup sums the scalar partials: the log-likelihood and the ones column's entry
        for (int which = 0; which < 2; ++which) {
            const double* part = which == 0 ? llpart : s0part;
            double* out = which == 0 ? loglik : s0;
            if (!out || !part) continue;
            __syncthreads();
            double t = 0.0;
            for (int b = threadIdx.x; b < nblocks; b += FINISH_COLS * FINISH_GROUPS) t += part[b];
            t = wave_allreduce_sum(t);
            if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][FINISH_COLS] = t;
            __syncthreads();
            if (threadIdx.x == 0) {
                double u = 0.0;
                for (int k = 0; k < FINISH_COLS * FINISH_GROUPS / 64; ++k) u += red[k][FINISH_COLS];
                *out = u;
            }
        }
    }
}

// (also the finish step of onehot.hip's structured logit pass)
void logit_finish_launch(const double* gpart, const double* llpart, int nblocks, int pitch, int p, double* g,
                         double* loglik, hipStream_t stream, const double* s0part, double* s0) {
    hipLaunchKernelGGL(logit_finish_kernel, dim3((p + FINISH_COLS - 1) / FINISH_COLS + 1), dim3(FINISH_COLS * FINISH_GROUPS),
                       0, stream, gpart, llpart, nblocks, pitch, p, g, loglik, s0part, s0);
}

static int logit_nc(int p) {
    const int chunks = (p + 127) / 128;
    int nc = 1;
    while (nc < chunks) nc *= 2;
    return nc;
}

static int logit_blocks(int64_t n, int rb) {
    const int64_t nbatch = (n + rb - 1) / rb;
    int64_t blocks = (nbatch + LOGIT_WAVES * 4 - 1) / (LOGIT_WAVES * 4);   // >= 4 batches per wave
    if (blocks < 1) blocks = 1;
    if (blocks > LOGIT_MAX_BLOCKS) blocks = LOGIT_MAX_BLOCKS;
    return (int)blocks;
}

template <int NC, int RB>
static void launch_logit(const LogitArgs& a, bool vec, int blocks, hipStream_t s) {
    if constexpr (RB == 4 || RB == 8) {
        if (a.img) {        // IMG form (irls_wide_pass_impl)
            if (vec) hipLaunchKernelGGL((logit_kernel<NC, RB, true, false, true>), dim3(blocks), dim3(LOGIT_THREADS), 0, s, a);
            else hipLaunchKernelGGL((logit_kernel<NC, RB, false, false, true>), dim3(blocks), dim3(LOGIT_THREADS), 0, s, a);
            return;
        }
    }
    if (a.hpart) {          // BORDER form (vector loads only: logit_border_ok)
        hipLaunchKernelGGL((logit_kernel<NC, RB, true, true>), dim3(blocks), dim3(LOGIT_THREADS), 0, s, a);
        return;
    }
    if (vec) hipLaunchKernelGGL((logit_kernel<NC, RB, true>), dim3(blocks), dim3(LOGIT_THREADS), 0, s, a);
    else hipLaunchKernelGGL((logit_kernel<NC, RB, false>), dim3(blocks), dim3(LOGIT_THREADS), 0, s, a);
}

size_t logit_workspace_bytes_impl(int64_t n, int p) {
    (void)n;
    const int nc = logit_nc(p);
    // (twice the g partials and three scalar partials: the BORDER form's X'w / sum w share the arena)
    return 2 * align_up((size_t)LOGIT_MAX_BLOCKS * nc * 128 * sizeof(double), 256) +
           3 * align_up((size_t)LOGIT_MAX_BLOCKS * sizeof(double), 256);
}

// irls_pass.hip: the streaming skeleton of the fused Newton pass (rows through an LDS-DMA ring) without the Hessian
bool irls_pass_fused_eligible(const double* X, int64_t ldx, const double* y, int64_t n, int p);
int irls_pass_impl(const double* X, int64_t ldx, const double* y, const double* beta, int64_t n, int p, double* H, int64_t ldh,
                   double* g, double* loglik, double* w_out, double* w_scratch, void* ws, size_t ws_bytes, hipStream_t stream,
                   int* fused_out);

static int logit_pass_run(const double* X, int64_t ldx, const double* y, const double* beta, int64_t n, int p,
                          double* w_out, double* g, double* loglik, void* ws, size_t ws_bytes, hipStream_t stream, int intercept, double* border,
                          unsigned* img = nullptr, int img_nb = 0);

// intercept != 0: beta and g have p + 1 entries, [intercept | the p columns of X]; X itself has p columns.
int logit_pass_impl(const double* X, int64_t ldx, const double* y, const double* beta, int64_t n, int p,
                    double* w_out, double* g, double* loglik, void* ws, size_t ws_bytes, hipStream_t stream, int intercept) {
    return logit_pass_run(X, ldx, y, beta, n, p, w_out, g, loglik, ws, ws_bytes, stream, intercept, nullptr);
}

// The logit pass of a fit with the implicit intercept that ALSO leaves border[0] = sum w, border[1 .. p] = X'w (the first row of
// [1 | X]' W [1 | X], models.py:121-130) from the weights it computes: the Gram pass that follows then needs no border pass of its
// own (config 3's reference-faithful call: one 100 GB read less per fresh Hessian).  Vector-load shapes only (logit_border_ok).
bool logit_border_ok(const double* X, int64_t ldx, int p) { return (ldx % 2 == 0) && (p % 2 == 0) && (((uintptr_t)X & 15) == 0); }
int logit_pass_border_impl(const double* X, int64_t ldx, const double* y, const double* beta, int64_t n, int p,
                           double* w_out, double* g, double* loglik, double* border, void* ws, size_t ws_bytes, hipStream_t stream) {
    DLSA_REQUIRE(border && logit_border_ok(X, ldx, p), "logit_pass (border): needs a border buffer and 16-byte aligned rows of even length");
    return logit_pass_run(X, ldx, y, beta, n, p, w_out, g, loglik, ws, ws_bytes, stream, 1, border);
}

// The logit pass that also leaves the bf16 chunk images of sqrt(w) [X | 1] (IMG form; 129 <= p <= 512 runs four rows per wave, narrower eight).
int logit_pass_image_impl(const double* X, int64_t ldx, const double* y, const double* beta, int64_t n, int p, double* w_out, double* g,
                          double* loglik, void* ws, size_t ws_bytes, hipStream_t stream, int intercept, unsigned* img, int img_nb) {
    DLSA_REQUIRE(img && img_nb > 0 && p <= 512, "logit_pass (image): needs an image buffer and p <= 512");
    return logit_pass_run(X, ldx, y, beta, n, p, w_out, g, loglik, ws, ws_bytes, stream, intercept, nullptr, img, img_nb);
}

static int logit_pass_run(const double* X, int64_t ldx, const double* y, const double* beta, int64_t n, int p,
                          double* w_out, double* g, double* loglik, void* ws, size_t ws_bytes, hipStream_t stream, int intercept, double* border,
                          unsigned* img, int img_nb) {
    DLSA_REQUIRE(X && y && beta, "logit_pass: null X, y or beta");
    DLSA_REQUIRE(p > 0 && n >= 0 && ldx >= p, "logit_pass: bad shape n=%lld p=%d ldx=%lld", (long long)n, p, (long long)ldx);
    DLSA_REQUIRE(p <= 2048, "logit_pass: p=%d > 2048 not supported", p);
    // (IMG with the intercept: the ones column is column p of the image, so the lanes' column chunks must reach it -- p = 128, 256: one chunk more)
    const int pw = img ? p + (intercept ? 1 : 0) : p;
    const int nc = logit_nc(pw);
    if (!ws || ws_bytes < logit_workspace_bytes_impl(n, pw) || ((uintptr_t)ws & 255)) {
        set_error("logit_pass: workspace %zu bytes needed (256-aligned), got %zu", logit_workspace_bytes_impl(n, pw), ws_bytes);
        return DLSA_ERR_WORKSPACE;
    }
    // Narrow designs (49 <= p <= 120, aligned rows): the rows come through the LDS-DMA ring of the fused Newton pass, which streams
    // at the HBM rate where this kernel's register loads reach 5.2-5.5 TB/s (DLSA_LOGIT_RING=0: keep the register-load kernel).
    if (!img && !intercept && irls_pass_fused_eligible(X, ldx, y, n, p) && (!w_out || ((uintptr_t)w_out % 8) == 0)) {
        const char* e = kernel_knob("DLSA_LOGIT_RING");
        if (!e || atoi(e) != 0)
            return irls_pass_impl(X, ldx, y, beta, n, p, nullptr, p, g, loglik, w_out, nullptr, ws, ws_bytes, stream, nullptr);
    }
    Arena ar(ws, ws_bytes);
    LogitArgs a;
    a.X = X; a.y = y; a.beta = intercept ? beta + 1 : beta; a.w_out = w_out; a.ldx = ldx; a.n = n; a.p = p;
    a.gpart = (double*)ar.take((size_t)LOGIT_MAX_BLOCKS * nc * 128 * sizeof(double));
    a.llpart = (double*)ar.take((size_t)LOGIT_MAX_BLOCKS * sizeof(double));
    a.s0part = (double*)ar.take((size_t)LOGIT_MAX_BLOCKS * sizeof(double));
    a.beta0 = intercept ? beta : nullptr;
    a.hpart = border ? (double*)ar.take((size_t)LOGIT_MAX_BLOCKS * nc * 128 * sizeof(double)) : nullptr;
    a.swpart = border ? (double*)ar.take((size_t)LOGIT_MAX_BLOCKS * sizeof(double)) : nullptr;
    a.img = img; a.img_nb = img_nb; a.img_ones = (img && intercept) ? 1 : 0;
    const bool vec = (ldx % 2 == 0) && (p % 2 == 0) && (((uintptr_t)X & 15) == 0);
    int blocks;
    switch (nc) {
        case 1: blocks = logit_blocks(n, 8); launch_logit<1, 8>(a, vec, blocks, stream); break;
        case 2: blocks = logit_blocks(n, 8); launch_logit<2, 8>(a, vec, blocks, stream); break;
        case 4: blocks = logit_blocks(n, 4); launch_logit<4, 4>(a, vec, blocks, stream); break;
        case 8: blocks = logit_blocks(n, 2); launch_logit<8, 2>(a, vec, blocks, stream); break;
        default: blocks = logit_blocks(n, 1); launch_logit<16, 1>(a, vec, blocks, stream); break;
    }
    DLSA_HIP_CHECK(hipGetLastError());
    if (g || loglik) {
        logit_finish_launch((const double*)a.gpart, (const double*)a.llpart, blocks, nc * 128, p, (g && intercept) ? g + 1 : g,
                            loglik, stream, (g && intercept) ? (const double*)a.s0part : nullptr, (g && intercept) ? g : nullptr);
        DLSA_HIP_CHECK(hipGetLastError());
    }
    if (border) {           // the same fixed-order sums for X'w (-> border[1 ..]) and sum w (-> border[0], through the kernel's loglik slot)
        logit_finish_launch((const double*)a.hpart, (const double*)a.swpart, blocks, nc * 128, p, border + 1, border, stream, nullptr, nullptr);
        DLSA_HIP_CHECK(hipGetLastError());
    }
    return DLSA_OK;
}

// ---------------------------------------------------------------------------------------------
// N1: log-likelihood of C estimator columns in ONE read of X (dlsa/models.py:217-222).
// Same row layout as logit_kernel; the RB x C dot products of a batch are reduced with one merged
// butterfly, after which every lane holds eta for one fixed (row-in-batch, column) pair.
// ---------------------------------------------------------------------------------------------
struct LoglikArgs {
    const double* X;
    const double* y;
    const double* par;     // p x c row-major (the rows of the p columns of X)
    const double* par0;    // implicit intercept: its row of c coefficients (nullable)
    double* llpart;        // [nblocks][C]
    int64_t ldx, ldpar, n;
    int p, c;
};

template <int NC, int RB, int C, bool VEC>
__global__ __launch_bounds__(LOGIT_THREADS) void loglik_kernel(LoglikArgs a) {
    __shared__ double red[LOGIT_WAVES][C];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double2 b[C][NC];
#pragma unroll
    for (int j = 0; j < C; ++j)
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int col = c * 128 + 2 * lane;
            b[j][c].x = (col < a.p && j < a.c) ? a.par[(int64_t)col * a.ldpar + j] : 0.0;
            b[j][c].y = (col + 1 < a.p && j < a.c) ? a.par[(int64_t)(col + 1) * a.ldpar + j] : 0.0;
        }
    const int myv = row_of_lane<RB * C>(lane);           // value index = row * C + column
    const int myrow = myv / C, mycol = myv % C;
    const bool rep = (lane & rep_mask<RB * C>()) == 0;
    const double icpt = (a.par0 && mycol < a.c) ? a.par0[mycol] : 0.0;
    double ll = 0.0;
    const int64_t nbatch = (a.n + RB - 1) / RB;
    const int64_t stride = (int64_t)gridDim.x * LOGIT_WAVES;
    for (int64_t bt = (int64_t)blockIdx.x * LOGIT_WAVES + wave; bt < nbatch; bt += stride) {
        const int64_t row0 = bt * RB;
        double dot[RB * C];
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            // branch-free clamped loads (see logit_kernel): clamped columns meet par = 0, clamped rows are dropped below
            const int64_t r = min(row0 + i, a.n - 1);
            const double* rowp = a.X + r * a.ldx;
            double2 x[NC];
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const int col = c * 128 + 2 * lane;
                const int c0 = col < a.p ? col : 0;
                if (VEC) x[c] = DLSA_LOGIT_LD2(rowp + c0);
                else { x[c].x = rowp[c0]; x[c].y = rowp[col + 1 < a.p ? col + 1 : 0]; }
            }
#pragma unroll
            for (int j = 0; j < C; ++j) {
                double s = 0.0;
#pragma unroll
                for (int c = 0; c < NC; ++c) s = fma(x[c].x, b[j][c].x, fma(x[c].y, b[j][c].y, s));
                dot[i * C + j] = s;
            }
        }
        const double eta = merged_reduce<RB * C>(dot, lane) + icpt;
        const int64_t r = row0 + myrow;
        if (rep && r < a.n && mycol < a.c) {
            const double yv = a.y[r];
            double mu_unused, w_unused, sp;
            logistic_terms<false>(eta, mu_unused, w_unused, sp);
            ll += yv * eta - sp;
        }
    }
#pragma unroll
    for (int j = 0; j < C; ++j) {
        double s = (rep && mycol == j) ? ll : 0.0;
        for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m, 64);
        if (lane == 0) red[wave][j] = s;
    }
    __syncthreads();
    if (tid < C) {
        double s = red[0][tid];
#pragma unroll
        for (int wv = 1; wv < LOGIT_WAVES; ++wv) s += red[wv][tid];
        a.llpart[(int64_t)blockIdx.x * C + tid] = s;
    }
}

template <int C>
__global__ void loglik_finish_kernel(const double* __restrict__ llpart, int nblocks, int c, double* __restrict__ out) {
    const int j = threadIdx.x >> 6, lane = threadIdx.x & 63;      // one wave per column
    if (j >= c) return;
    double s = 0.0;
    for (int b = lane; b < nblocks; b += 64) s += llpart[(int64_t)b * C + j];
    for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m, 64);
    if (lane == 0) out[j] = s;
}

// ---------------------------------------------------------------------------------------------
// N3 (linear-model map): g = X'v and vv = v'v in one read of X (v = the response).  Same row layout.
// ---------------------------------------------------------------------------------------------
template <int NC, int RB, bool VEC>
__global__ __launch_bounds__(LOGIT_THREADS) void xtv_kernel(LogitArgs a) {
    __shared__ double red[NC * 128 + 2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double2 g[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) { g[c].x = 0.0; g[c].y = 0.0; }
    double vv = 0.0, sv = 0.0;
    const int64_t nbatch = (a.n + RB - 1) / RB;
    const int64_t stride = (int64_t)gridDim.x * LOGIT_WAVES;
    for (int64_t bt = (int64_t)blockIdx.x * LOGIT_WAVES + wave; bt < nbatch; bt += stride) {
        const int64_t row0 = bt * RB;
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            // branch-free clamped loads (see logit_kernel): a row past n gets weight 0, columns past p land in
            // entries of g that nobody reads
            const int64_t r = min(row0 + i, a.n - 1);
            const double yraw = a.y ? a.y[r] : 1.0;
            const double* rowp = a.X + r * a.ldx;
            double2 v[NC];
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const int col = c * 128 + 2 * lane;
                const int c0 = col < a.p ? col : 0;
                if (VEC) v[c] = DLSA_LOGIT_LD2(rowp + c0);
                else { v[c].x = rowp[c0]; v[c].y = rowp[col + 1 < a.p ? col + 1 : 0]; }
            }
            const double yv = (row0 + i < a.n) ? yraw : 0.0;
            if (lane == 0) { vv = fma(yv, yv, vv); sv += yv; }
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                g[c].x = fma(yv, v[c].x, g[c].x);
                g[c].y = fma(yv, v[c].y, g[c].y);
            }
        }
    }
    for (int m = 32; m >= 1; m >>= 1) { vv += __shfl_xor(vv, m, 64); sv += __shfl_xor(sv, m, 64); }
    for (int wv = 0; wv < LOGIT_WAVES; ++wv) {
        if (wave == wv) {
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                double* dst = red + c * 128 + 2 * lane;
                if (wv == 0) { dst[0] = g[c].x; dst[1] = g[c].y; }
                else { dst[0] += g[c].x; dst[1] += g[c].y; }
            }
            if (lane == 0) {
                if (wv == 0) { red[NC * 128] = vv; red[NC * 128 + 1] = sv; }
                else { red[NC * 128] += vv; red[NC * 128 + 1] += sv; }
            }
        }
        __syncthreads();
    }
    double* gp = a.gpart + (int64_t)blockIdx.x * (NC * 128);
    for (int col = tid; col < NC * 128; col += LOGIT_THREADS) gp[col] = red[col];
    if (tid == 0) { a.llpart[blockIdx.x] = red[NC * 128]; if (a.s0part) a.s0part[blockIdx.x] = red[NC * 128 + 1]; }
}

// fp32 rows (config 5, wide-p linear model): lane l holds columns 256c + 4l + {0..3} (16-byte loads);
// products are accumulated in fp64 and written back as fp32.
struct XtvF32Args {
    const float* X;
    const float* v;
    double* gpart;     // [nblocks][NC*256]
    double* vvpart;    // [nblocks]
    int64_t ldx, n;
    int p;
};

template <int NC, bool VEC>
__global__ __launch_bounds__(LOGIT_THREADS) void xtv_f32_kernel(XtvF32Args a) {
    __shared__ double red[NC * 256 + 1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double g[NC][4];
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e) g[c][e] = 0.0;
    double vv = 0.0;
    const int64_t stride = (int64_t)gridDim.x * LOGIT_WAVES;
    for (int64_t r = (int64_t)blockIdx.x * LOGIT_WAVES + wave; r < a.n; r += stride) {
        const double yv = (double)a.v[r];
        if (lane == 0) vv = fma(yv, yv, vv);
        const float* src = a.X + r * a.ldx + 4 * lane;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int col = c * 256 + 4 * lane;
            float4 x; x.x = x.y = x.z = x.w = 0.f;
            if (col + 3 < a.p) {
                if (VEC) x = DLSA_LOGIT_LD4F(src + c * 256);
                else { x.x = src[c * 256]; x.y = src[c * 256 + 1]; x.z = src[c * 256 + 2]; x.w = src[c * 256 + 3]; }
            } else {
                if (col < a.p) x.x = src[c * 256];
                if (col + 1 < a.p) x.y = src[c * 256 + 1];
                if (col + 2 < a.p) x.z = src[c * 256 + 2];
            }
            g[c][0] = fma(yv, (double)x.x, g[c][0]);
            g[c][1] = fma(yv, (double)x.y, g[c][1]);
            g[c][2] = fma(yv, (double)x.z, g[c][2]);
            g[c][3] = fma(yv, (double)x.w, g[c][3]);
        }
    }
    for (int m = 32; m >= 1; m >>= 1) vv += __shfl_xor(vv, m, 64);
    for (int wv = 0; wv < LOGIT_WAVES; ++wv) {
        if (wave == wv) {
#pragma unroll
            for (int c = 0; c < NC; ++c)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    double* dst = red + c * 256 + 4 * lane + e;
                    if (wv == 0) *dst = g[c][e]; else *dst += g[c][e];
                }
            if (lane == 0) { if (wv == 0) red[NC * 256] = vv; else red[NC * 256] += vv; }
        }
        __syncthreads();
    }
    double* gp = a.gpart + (int64_t)blockIdx.x * (NC * 256);
    for (int col = tid; col < NC * 256; col += LOGIT_THREADS) gp[col] = red[col];
    if (tid == 0) a.vvpart[blockIdx.x] = red[NC * 256];
}

__global__ __launch_bounds__(1024) void xtv_f32_finish_kernel(const double* __restrict__ gpart, const double* __restrict__ vvpart,
                                                              int nblocks, int pitch, int p, float* __restrict__ g,
                                                              float* __restrict__ vv) {
    __shared__ double red[16][65];
    const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + cx;
    double s = 0.0;
    if (col < p)
        for (int b = ry; b < nblocks; b += 16) s += gpart[(int64_t)b * pitch + col];
    red[ry][cx] = s;
    __syncthreads();
    if (ry == 0 && col < p) {
        double t = red[0][cx];
        for (int k = 1; k < 16; ++k) t += red[k][cx];
        g[col] = (float)t;
    }
    if (vv && blockIdx.x == 0 && threadIdx.x == 0) {
        double t = 0.0;
        for (int b = 0; b < nblocks; ++b) t += vvpart[b];
        *vv = (float)t;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Streaming linear-model statistics (config 5's map step, chunk by chunk): ONE read of X gives, in fp64 whatever the rows'
// type,  g = X'v,  c = X'1 (the column sums: the intercept's border of [1 | X]'[1 | X], dlsa/models.py:121-122 -- the ones
// column stays implicit),  v'v  and  sum v;  accumulate != 0 ADDS them to what the outputs hold (the previous chunks).
// A wave owns two rows per trip (two sets of loads in flight); lane l holds columns CW c + E l + {0..E-1} (16-byte loads).
// ---------------------------------------------------------------------------------------------------------------------
template <typename T>
struct XtvStatsArgs {
    const T* X;
    const T* v;
    double* part;      // [nblocks][2 * NC * CW + 2]: g | colsum | v'v | sum v
    int64_t ldx, n;
    int p, want_colsum;
};

template <typename T, int NC, bool VEC>
__global__ __launch_bounds__(LOGIT_THREADS) void xtv_stats_kernel(XtvStatsArgs<T> a) {
    constexpr int E = 16 / (int)sizeof(T), CW = 64 * E;          // elements per lane and load, columns per wave pass
    extern __shared__ double red_stats[];                        // [2 * NC * CW + 2]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double g[NC][E], cs[NC][E];
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int e = 0; e < E; ++e) { g[c][e] = 0.0; cs[c][e] = 0.0; }
    double vv = 0.0, sv = 0.0;
    auto load_row = [&](int64_t r, T (&x)[NC][E]) {
        const T* src = a.X + r * a.ldx + E * lane;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int col = c * CW + E * lane;
            if (VEC && col + E - 1 < a.p) {
                typedef T vecT __attribute__((ext_vector_type(E)));
                const vecT q = DLSA_STREAM_LOAD(reinterpret_cast<const vecT*>(src + c * CW));
#pragma unroll
                for (int e = 0; e < E; ++e) x[c][e] = q[e];
            } else {
#pragma unroll
                for (int e = 0; e < E; ++e) x[c][e] = (col + e < a.p) ? src[c * CW + e] : T(0);
            }
        }
    };
    auto add_row = [&](double yv, const T (&x)[NC][E]) {
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int e = 0; e < E; ++e) {
                g[c][e] = fma(yv, (double)x[c][e], g[c][e]);
                if (a.want_colsum) cs[c][e] += (double)x[c][e];
            }
    };
    const int64_t stride = (int64_t)gridDim.x * LOGIT_WAVES * 2;
    for (int64_t r = ((int64_t)blockIdx.x * LOGIT_WAVES + wave) * 2; r < a.n; r += stride) {
        T x0[NC][E], x1[NC][E];
        const bool two = r + 1 < a.n;
        load_row(r, x0);
        load_row(two ? r + 1 : r, x1);
        const double y0 = (double)a.v[r], y1 = two ? (double)a.v[r + 1] : 0.0;
        if (lane == 0) { vv = fma(y0, y0, vv); vv = fma(y1, y1, vv); sv += y0 + y1; }
        add_row(y0, x0);
        if (two) add_row(y1, x1);
    }
    vv = wave_allreduce_sum(vv);
    sv = wave_allreduce_sum(sv);
    constexpr int NG = NC * CW;
    for (int wv = 0; wv < LOGIT_WAVES; ++wv) {                   // the waves meet in a fixed order: bit-reproducible
        if (wave == wv) {
#pragma unroll
            for (int c = 0; c < NC; ++c)
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const int k = c * CW + E * lane + e;
                    if (wv == 0) { red_stats[k] = g[c][e]; red_stats[NG + k] = cs[c][e]; }
                    else { red_stats[k] += g[c][e]; red_stats[NG + k] += cs[c][e]; }
                }
            if (lane == 0) {
                if (wv == 0) { red_stats[2 * NG] = vv; red_stats[2 * NG + 1] = sv; }
                else { red_stats[2 * NG] += vv; red_stats[2 * NG + 1] += sv; }
            }
        }
        __syncthreads();
    }
    double* gp = a.part + (int64_t)blockIdx.x * (2 * NG + 2);
    for (int k = tid; k < 2 * NG + 2; k += LOGIT_THREADS) gp[k] = red_stats[k];
}

// sums the per-block partials in a fixed order; out = g [p] | colsum [p] (nullable) | stats[0] = v'v, stats[1] = sum v
__global__ __launch_bounds__(1024) void xtv_stats_finish_kernel(const double* __restrict__ part, int nblocks, int pitch, int ng, int p,
                                                                double* __restrict__ g, double* __restrict__ colsum,
                                                                double* __restrict__ stats, int accumulate) {
    __shared__ double red[16][65];
    const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
    const int which = blockIdx.y;                                // 0: g, 1: colsum, 2: the two scalars
    if (which == 1 && !colsum) return;
    int col = blockIdx.x * 64 + cx, src_col;
    bool live;
    if (which < 2) { live = col < p; src_col = which * ng + col; }
    else { if (blockIdx.x) return; live = cx < 2; src_col = 2 * ng + cx; }
    double s = 0.0;
    if (live)
        for (int b = ry; b < nblocks; b += 16) s += part[(int64_t)b * pitch + src_col];
    red[ry][cx] = s;
    __syncthreads();
    if (ry == 0 && live) {
        double t = red[0][cx];
        for (int k = 1; k < 16; ++k) t += red[k][cx];
        double* dst = which == 0 ? g + col : which == 1 ? colsum + col : stats + cx;
        *dst = accumulate ? *dst + t : t;
    }
}

template <typename T>
static size_t xtv_stats_ws_bytes(int p) {
    constexpr int CW = 64 * (16 / (int)sizeof(T));
    int nc = 1;
    while (nc * CW < p) nc *= 2;
    return align_up((size_t)LOGIT_MAX_BLOCKS * (2 * nc * CW + 2) * sizeof(double), 256);
}

template <typename T>
int xtv_stats_impl(const T* X, int64_t ldx, const T* v, int64_t n, int p, double* g, double* colsum, double* stats,
                   int accumulate, void* ws, size_t ws_bytes, hipStream_t s) {
    constexpr int E = 16 / (int)sizeof(T), CW = 64 * E;
    DLSA_REQUIRE((X || n == 0) && (v || n == 0) && g && stats, "xtv_stats: null argument");
    DLSA_REQUIRE(p > 0 && p <= 2048 && n >= 0 && ldx >= p, "xtv_stats: bad shape n=%lld p=%d ldx=%lld", (long long)n, p, (long long)ldx);
    const size_t need = xtv_stats_ws_bytes<T>(p);
    if (!ws || ws_bytes < need || ((uintptr_t)ws & 255)) {
        set_error("xtv_stats: workspace %zu bytes needed (256-aligned), got %zu", need, ws_bytes);
        return DLSA_ERR_WORKSPACE;
    }
    int nc = 1;
    while (nc * CW < p) nc *= 2;
    XtvStatsArgs<T> a;
    a.X = X; a.v = v; a.part = (double*)ws; a.ldx = ldx; a.n = n; a.p = p; a.want_colsum = colsum ? 1 : 0;
    const bool vec = (ldx % E == 0) && (((uintptr_t)X & 15) == 0);
    const int64_t blocks64 = (n + LOGIT_WAVES * 32 - 1) / (LOGIT_WAVES * 32);
    const int blocks = (int)std::min<int64_t>(std::max<int64_t>(blocks64, 1), LOGIT_MAX_BLOCKS);
    const size_t shm = (size_t)(2 * nc * CW + 2) * sizeof(double);
#define DLSA_XTV_STATS(NCV) do { \
        if (shm > 48 * 1024) { \
            DLSA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(xtv_stats_kernel<T, NCV, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm)); \
            DLSA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(xtv_stats_kernel<T, NCV, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm)); \
        } \
        if (vec) hipLaunchKernelGGL((xtv_stats_kernel<T, NCV, true>), dim3(blocks), dim3(LOGIT_THREADS), shm, s, a); \
        else hipLaunchKernelGGL((xtv_stats_kernel<T, NCV, false>), dim3(blocks), dim3(LOGIT_THREADS), shm, s, a); } while (0)
    if (n > 0) {
        switch (nc) {
            case 1: DLSA_XTV_STATS(1); break;
            case 2: DLSA_XTV_STATS(2); break;
            case 4: DLSA_XTV_STATS(4); break;
            case 8: DLSA_XTV_STATS(8); break;
            default:
                if constexpr (sizeof(T) == 8) { DLSA_XTV_STATS(16); }
                else { set_error("xtv_stats: p too wide"); return DLSA_ERR_INVALID; }
        }
    }
#undef DLSA_XTV_STATS
    DLSA_HIP_CHECK(hipGetLastError());
    hipLaunchKernelGGL(xtv_stats_finish_kernel, dim3((p + 63) / 64, 3), dim3(1024), 0, s, (const double*)ws, n > 0 ? blocks : 0,
                       2 * nc * CW + 2, nc * CW, p, g, colsum, stats, accumulate);
    DLSA_HIP_CHECK(hipGetLastError());
    return DLSA_OK;
}

template <int NC>
static void launch_xtv_f32(const XtvF32Args& a, bool vec, int blocks, hipStream_t s) {
    if (vec) hipLaunchKernelGGL((xtv_f32_kernel<NC, true>), dim3(blocks), dim3(LOGIT_THREADS), 0, s, a);
    else hipLaunchKernelGGL((xtv_f32_kernel<NC, false>), dim3(blocks), dim3(LOGIT_THREADS), 0, s, a);
}

template <int NC, int RB>
static void launch_xtv(const LogitArgs& a, bool vec, int blocks, hipStream_t s) {
    if (vec) hipLaunchKernelGGL((xtv_kernel<NC, RB, true>), dim3(blocks), dim3(LOGIT_THREADS), 0, s, a);
    else hipLaunchKernelGGL((xtv_kernel<NC, RB, false>), dim3(blocks), dim3(LOGIT_THREADS), 0, s, a);
}

template <int NC, int RB, int C>
static void launch_loglik(const LoglikArgs& a, bool vec, int blocks, hipStream_t s) {
    if (vec) hipLaunchKernelGGL((loglik_kernel<NC, RB, C, true>), dim3(blocks), dim3(LOGIT_THREADS), 0, s, a);
    else hipLaunchKernelGGL((loglik_kernel<NC, RB, C, false>), dim3(blocks), dim3(LOGIT_THREADS), 0, s, a);
}

}  // namespace dlsa

extern "C" {

// N1 (dlsa/models.py:217-222): all c <= 8 estimator columns in one read of X.  intercept != 0: par has p + 1 rows, the
// first one the intercepts (the ones column of models.py:162-165 is implicit).
static int loglik_impl(const double* X, int64_t ldx, const double* y, int64_t n, int p, const double* par,
                       int64_t ldpar, int c, double* out, void* ws, size_t ws_bytes, void* stream, int intercept) {
    using namespace dlsa;
    DLSA_REQUIRE(X && y && par && out, "loglik: null argument");
    DLSA_REQUIRE(p > 0 && p <= 2048 && n >= 0 && ldx >= p, "loglik: bad shape n=%lld p=%d", (long long)n, p);
    DLSA_REQUIRE(c > 0 && c <= 8 && ldpar >= c, "loglik: need 1 <= c <= 8 and ldpar >= c");
    const size_t need = align_up((size_t)LOGIT_MAX_BLOCKS * 8 * sizeof(double), 256);
    if (!ws || ws_bytes < need || ((uintptr_t)ws & 255)) {
        set_error("loglik: workspace %zu bytes needed (256-aligned), got %zu", need, ws_bytes);
        return DLSA_ERR_WORKSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    LoglikArgs a;
    a.X = X; a.y = y; a.par = intercept ? par + ldpar : par; a.par0 = intercept ? par : nullptr;
    a.llpart = (double*)ws; a.ldx = ldx; a.ldpar = ldpar; a.n = n; a.p = p; a.c = c;
    const bool vec = (ldx % 2 == 0) && (p % 2 == 0) && (((uintptr_t)X & 15) == 0);
    const int nc = logit_nc(p);
    int blocks;
    // RB*C = 16 values per merged butterfly (8 for the widest rows)
    if (c <= 4) {
        switch (nc) {
            case 1: blocks = logit_blocks(n, 4); launch_loglik<1, 4, 4>(a, vec, blocks, s); break;
            case 2: blocks = logit_blocks(n, 4); launch_loglik<2, 4, 4>(a, vec, blocks, s); break;
            case 4: blocks = logit_blocks(n, 4); launch_loglik<4, 4, 4>(a, vec, blocks, s); break;
            case 8: blocks = logit_blocks(n, 2); launch_loglik<8, 2, 4>(a, vec, blocks, s); break;
            default: blocks = logit_blocks(n, 1); launch_loglik<16, 1, 4>(a, vec, blocks, s); break;
        }
        hipLaunchKernelGGL((loglik_finish_kernel<4>), dim3(1), dim3(64 * 4), 0, s, (const double*)a.llpart, blocks, c, out);
    } else {
        switch (nc) {
            case 1: blocks = logit_blocks(n, 2); launch_loglik<1, 2, 8>(a, vec, blocks, s); break;
            case 2: blocks = logit_blocks(n, 2); launch_loglik<2, 2, 8>(a, vec, blocks, s); break;
            case 4: blocks = logit_blocks(n, 2); launch_loglik<4, 2, 8>(a, vec, blocks, s); break;
            case 8: blocks = logit_blocks(n, 1); launch_loglik<8, 1, 8>(a, vec, blocks, s); break;
            default: {      // p > 1024 with more than 4 columns: two passes of 4 columns
                int rc = loglik_impl(X, ldx, y, n, p, par, ldpar, 4, out, ws, ws_bytes, stream, intercept);
                if (rc) return rc;
                return loglik_impl(X, ldx, y, n, p, par + 4, ldpar, c - 4, out + 4, ws, ws_bytes, stream, intercept);
            }
        }
        hipLaunchKernelGGL((loglik_finish_kernel<8>), dim3(1), dim3(64 * 8), 0, s, (const double*)a.llpart, blocks, c, out);
    }
    DLSA_HIP_CHECK(hipGetLastError());
    return DLSA_OK;
}

int dlsa_loglik_f64(const double* X, int64_t ldx, const double* y, int64_t n, int p, const double* par,
                    int64_t ldpar, int c, double* out, void* ws, size_t ws_bytes, void* stream) {
    return loglik_impl(X, ldx, y, n, p, par, ldpar, c, out, ws, ws_bytes, stream, 0);
}
int dlsa_loglik_icpt_f64(const double* X, int64_t ldx, const double* y, int64_t n, int p, const double* par,
                         int64_t ldpar, int c, double* out, void* ws, size_t ws_bytes, void* stream) {
    return loglik_impl(X, ldx, y, n, p, par, ldpar, c, out, ws, ws_bytes, stream, 1);
}

}  // extern "C"

namespace dlsa {
// g = X'v (p values), vv = v'v, sv = sum v (both nullable) in one read of X; v == nullptr stands for the all-ones vector
int xtv_impl(const double* X, int64_t ldx, const double* v, int64_t n, int p, double* g, double* vv, double* sv,
             void* ws, size_t ws_bytes, hipStream_t s) {
    DLSA_REQUIRE(X && g, "xtv: null argument");
    DLSA_REQUIRE(p > 0 && p <= 2048 && n >= 0 && ldx >= p, "xtv: bad shape n=%lld p=%d", (long long)n, p);
    const int nc = logit_nc(p);
    if (!ws || ws_bytes < logit_workspace_bytes_impl(n, p) || ((uintptr_t)ws & 255)) {
        set_error("xtv: workspace %zu bytes needed (256-aligned), got %zu", logit_workspace_bytes_impl(n, p), ws_bytes);
        return DLSA_ERR_WORKSPACE;
    }
    Arena ar(ws, ws_bytes);
    LogitArgs a;
    a.X = X; a.y = v; a.beta = nullptr; a.w_out = nullptr; a.ldx = ldx; a.n = n; a.p = p;
    a.gpart = (double*)ar.take((size_t)LOGIT_MAX_BLOCKS * nc * 128 * sizeof(double));
    a.llpart = (double*)ar.take((size_t)LOGIT_MAX_BLOCKS * sizeof(double));
    a.s0part = (double*)ar.take((size_t)LOGIT_MAX_BLOCKS * sizeof(double));
    a.beta0 = nullptr;
    const bool vec = (ldx % 2 == 0) && (p % 2 == 0) && (((uintptr_t)X & 15) == 0);
    int blocks;
    switch (nc) {
        case 1: blocks = logit_blocks(n, 8); launch_xtv<1, 8>(a, vec, blocks, s); break;
        case 2: blocks = logit_blocks(n, 8); launch_xtv<2, 8>(a, vec, blocks, s); break;
        case 4: blocks = logit_blocks(n, 4); launch_xtv<4, 4>(a, vec, blocks, s); break;
        case 8: blocks = logit_blocks(n, 2); launch_xtv<8, 2>(a, vec, blocks, s); break;
        default: blocks = logit_blocks(n, 1); launch_xtv<16, 1>(a, vec, blocks, s); break;
    }
    logit_finish_launch((const double*)a.gpart, (const double*)a.llpart, blocks, nc * 128, p, g, vv, s,
                        (const double*)a.s0part, sv);
    DLSA_HIP_CHECK(hipGetLastError());
    return DLSA_OK;
}
}  // namespace dlsa

extern "C" {

// N3: g = X'v (p values) and vv = v'v (1 value, nullable) in one read of X -- the linear-model map
// step's X'y (no DLSA implementation in the reference; README.md:6 claims the method).
int dlsa_xtv_f64(const double* X, int64_t ldx, const double* v, int64_t n, int p, double* g, double* vv,
                 void* ws, size_t ws_bytes, void* stream) {
    DLSA_REQUIRE(v, "xtv: null argument");
    return dlsa::xtv_impl(X, ldx, v, n, p, g, vv, nullptr, ws, ws_bytes, (hipStream_t)stream);
}

int dlsa_xtv_f32(const float* X, int64_t ldx, const float* v, int64_t n, int p, float* g, float* vv,
                 void* ws, size_t ws_bytes, void* stream) {
    using namespace dlsa;
    DLSA_REQUIRE(X && v && g, "xtv_f32: null argument");
    DLSA_REQUIRE(p > 0 && p <= 2048 && n >= 0 && ldx >= p, "xtv_f32: bad shape n=%lld p=%d", (long long)n, p);
    const int chunks = (p + 255) / 256;
    int nc = 1;
    while (nc < chunks) nc *= 2;
    const size_t need = align_up((size_t)LOGIT_MAX_BLOCKS * nc * 256 * sizeof(double), 256) + align_up((size_t)LOGIT_MAX_BLOCKS * 8, 256);
    if (!ws || ws_bytes < need || ((uintptr_t)ws & 255)) {
        set_error("xtv_f32: workspace %zu bytes needed (256-aligned), got %zu", need, ws_bytes);
        return DLSA_ERR_WORKSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    Arena ar(ws, ws_bytes);
    XtvF32Args a;
    a.X = X; a.v = v; a.ldx = ldx; a.n = n; a.p = p;
    a.gpart = (double*)ar.take((size_t)LOGIT_MAX_BLOCKS * nc * 256 * sizeof(double));
    a.vvpart = (double*)ar.take((size_t)LOGIT_MAX_BLOCKS * sizeof(double));
    const bool vec = (ldx % 4 == 0) && (((uintptr_t)X & 15) == 0);
    int64_t blocks64 = (n + LOGIT_WAVES * 16 - 1) / (LOGIT_WAVES * 16);
    const int blocks = (int)std::min<int64_t>(std::max<int64_t>(blocks64, 1), LOGIT_MAX_BLOCKS);
    switch (nc) {
        case 1: launch_xtv_f32<1>(a, vec, blocks, s); break;
        case 2: launch_xtv_f32<2>(a, vec, blocks, s); break;
        case 4: launch_xtv_f32<4>(a, vec, blocks, s); break;
        default: launch_xtv_f32<8>(a, vec, blocks, s); break;
    }
    hipLaunchKernelGGL(xtv_f32_finish_kernel, dim3((p + 63) / 64), dim3(1024), 0, s, (const double*)a.gpart,
                       (const double*)a.vvpart, blocks, nc * 256, p, g, vv);
    DLSA_HIP_CHECK(hipGetLastError());
    return DLSA_OK;
}

size_t dlsa_xtv_stats_workspace_bytes(int p, int elem_bytes) {
    if (p <= 0 || p > 2048 || (elem_bytes != 4 && elem_bytes != 8)) return 0;
    return elem_bytes == 8 ? dlsa::xtv_stats_ws_bytes<double>(p) : dlsa::xtv_stats_ws_bytes<float>(p);
}
int dlsa_xtv_stats_f64(const double* X, int64_t ldx, const double* v, int64_t n, int p, double* g, double* colsum,
                       double* stats, int accumulate, void* ws, size_t ws_bytes, void* stream) {
    return dlsa::xtv_stats_impl<double>(X, ldx, v, n, p, g, colsum, stats, accumulate, ws, ws_bytes, (hipStream_t)stream);
}
int dlsa_xtv_stats_f32(const float* X, int64_t ldx, const float* v, int64_t n, int p, double* g, double* colsum,
                       double* stats, int accumulate, void* ws, size_t ws_bytes, void* stream) {
    return dlsa::xtv_stats_impl<float>(X, ldx, v, n, p, g, colsum, stats, accumulate, ws, ws_bytes, (hipStream_t)stream);
}

size_t dlsa_logit_workspace_bytes(int64_t n, int p) {
    if (p <= 0 || p > 2048 || n < 0) return 0;
    return dlsa::logit_workspace_bytes_impl(n, p);
}

int dlsa_logit_pass_f64(const double* X, int64_t ldx, const double* y, const double* beta, int64_t n,
                        int p, double* w_out, double* g, double* loglik, void* ws, size_t ws_bytes,
                        void* stream) {
    return dlsa::logit_pass_impl(X, ldx, y, beta, n, p, w_out, g, loglik, ws, ws_bytes, (hipStream_t)stream, 0);
}

int dlsa_logit_pass_icpt_f64(const double* X, int64_t ldx, const double* y, const double* beta, int64_t n,
                             int p, double* w_out, double* g, double* loglik, void* ws, size_t ws_bytes,
                             void* stream) {
    return dlsa::logit_pass_impl(X, ldx, y, beta, n, p, w_out, g, loglik, ws, ws_bytes, (hipStream_t)stream, 1);
}

}  // extern "C"
