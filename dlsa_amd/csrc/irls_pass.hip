// One FUSED Newton pass for narrow fp64 designs (49 <= p <= 120): in ONE read of the rows
//     eta = X beta,  mu = sigmoid(eta),  w = mu (1 - mu),  g = X'(y - mu),  loglik,  H = X' diag(w) X.
// Reference call sites: dlsa/models.py:110-114 (the solver's inner products and predict_proba), :130 (the Hessian).  The
// reference reads a partition's rows three times per evaluation (fit iteration, predict_proba, the .dot of :130); the
// unfused engine twice (logit.hip's pass writes w, gram_narrow.hip's pass reads X and w).  At p = 100 the Gram pass sits at
// the chip's power limit with 3.8 TB/s streaming, and the logit pass costs two thirds of a Gram pass: every fresh Hessian
// pays 1.66 passes.  Here the rows a workgroup has staged in LDS for the MFMAs also feed the logistic terms:
//
//   * gram_narrow.hip's layout: the whole upper triangle of H (NT full tiles + G tail groups) lives in ONE wave's AGPRs, the
//     four waves of a workgroup split the ROWS (wave m takes k-steps m and m + 4 of every 32-row chunk), rows stream through a
//     4-stage LDS-DMA ring three chunks ahead;
//   * while a wave's MFMAs of chunk c run, the same wave evaluates the logistic terms of ITS OWN eight rows of chunk c + 1 (already
//     landed) from LDS: lane (j = l >> 3, s = l & 7) holds columns 16 q + 2 s + {0, 1} of row j -- NTC 16-byte LDS reads,
//     2 NTC FMAs against beta (registers), a 3-step DPP butterfly over the 8 lanes of a row, the lean transcendentals of
//     logistic.h, the rank-one update g += (y - mu) x on the registers it still holds.  The pieces are issued one behind
//     each SEGMENT of the generated MFMA blocks (gram_narrow_asm.inc), so their latencies hide under the matrix pipe;
//   * w goes to the stage's w slot in LDS, where the MFMA part of the next iteration reads it exactly as gram_narrow.hip
//     reads the DMA'd weights -- by the same wave that wrote it (no cross-wave dependency, no extra barrier);
//   * y arrives through the same LDS-DMA ring (one piece per wave and chunk in place of gram_narrow's w piece).
// One workgroup per CU, one wave per SIMD (the wave needs both register sets: 182 accumulators + fragments + beta + g).
// Outputs per slab: the H partial (summed by gram.hip's reduce kernel), g and loglik partials (summed in a fixed order by
// irls_pass_finish_kernel): bit-reproducible run to run.
#include "common.h"
#include "options.h"
#include "irls_batch.h"
#include <algorithm>
#include <type_traits>

#ifndef DLSA_FUSED_PRIVATE
#define DLSA_FUSED_PRIVATE 1        // 1: the fused pass's waves stage their own rows / labels and run without a barrier in the chunk loop (round 4); 0: round-robin rows, one barrier per chunk
#endif
#ifndef DLSA_FUSED_EARLY_FRAGS
#define DLSA_FUSED_EARLY_FRAGS 1    // 1: the next chunk's fragments are requested behind segment 1 of the second k-step instead of at the iteration's end
#endif
#ifndef DLSA_FUSED_BATCH
#define DLSA_FUSED_BATCH 1          // 1: the transcendentals of a wave's own rows are evaluated once per TWO chunks, 16 rows x 4 copies instead of 8 rows x 8 copies (round 4); needs a fifth LDS stage
#endif
#ifndef FP_TIMELINE
#define FP_TIMELINE 0               // profiling builds only (bench/fused_timeline.py): every workgroup leaves its s_memrealtime stamps in the free slots of its g partial
#endif
#ifndef FP_ABL
#define FP_ABL 0                    // ABLATION builds only (wrong results; bench/build_variant.sh ... -DFP_ABL=n): 1 no log1p / loglik, 2 no logistic terms, 4 no DMA in the loop, 8 no MFMAs
#endif
#ifndef DLSA_FUSED_SCONST
#define DLSA_FUSED_SCONST 1         // 1: polynomial coefficients of the logistic terms in SGPRs (no v_mov per Horner step)
#endif
#ifndef DLSA_STREAM_AUX
#define DLSA_STREAM_AUX 2           // nt on the LDS-DMA row stream (each row is read by one workgroup, once): ring logit pass -7..-8 % at p = 100-112, narrow Gram +3 % at p = 64, neutral at p = 100; 0 = default policy
#endif

namespace dlsa {

template <typename T>
void gram_reduce_launch(const T* partial, int nslab, int PP, int p, T* H, int64_t ldh, int accumulate, hipStream_t stream);   // gram.hip
int logit_pass_impl(const double* X, int64_t ldx, const double* y, const double* beta, int64_t n, int p, double* w_out, double* g,
                    double* loglik, void* ws, size_t ws_bytes, hipStream_t s, int intercept);                                   // logit.hip
size_t logit_workspace_bytes_impl(int64_t n, int p);
int gram_impl_f64(const double* X, int64_t ldx, const double* w, int64_t n, int p, double* H, int64_t ldh, int accumulate, void* ws,
                  size_t ws_bytes, hipStream_t stream);                                                                         // gram.hip
size_t gram_workspace_bytes_impl(int64_t n, int p, int elem_bytes);

#include "logistic.h"
#include "gram_narrow_asm.inc"

// q * r + C with the fp64 coefficient C as a SCALAR operand.  hipcc turns fma(q, r, literal) into v_mov_b64 + v_fmac_f64 (and
// an SGPR-held coefficient into s_mov + v_mov + v_fmac): two or three VALU slots per Horner step -- and next to fp64 MFMAs EVERY
// VALU instruction, a move included, costs ~4.7 cycles of the matrix pipe while SALU instructions cost nothing
// (bench/ubench_gap.hip, profiles/r04_ubench_gap.txt).  The asm form is one VALU slot: v_fma_f64 with the coefficient in the one
// scalar operand a VOP3 instruction may have (the s_mov_b32 pair that fills it is SALU).  Same arithmetic, same bits.
template <bool SC = true>
__device__ __forceinline__ double fp_fma_sc2(double a, double c, double b) {      // a * C + b
#if DLSA_FUSED_SCONST
    if constexpr (!SC) return fma(a, c, b);
    double d;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "s"(c), "v"(b));
    return d;
#else
    return fma(a, c, b);
#endif
}
// a 32-bit / 64-bit constant pinned in VGPRs (opaque to hipcc: it cannot re-create it with a v_mov inside the loop)
// (PIN = false: the plain constant -- the widest shape, 7 tiles + 2 tail groups with w_out, has no 14 registers to spare)
template <bool PIN> __device__ __forceinline__ int fp_pin32(int v) { if constexpr (!PIN) return v; int d; asm("v_mov_b32 %0, %1" : "=v"(d) : "s"(v)); return d; }
template <bool PIN> __device__ __forceinline__ double fp_pin64(double v) { if constexpr (!PIN) return v; double d; asm("v_mov_b64 %0, %1" : "=v"(d) : "s"(v)); return d; }
template <bool SC = true>
__device__ __forceinline__ double fp_fma_sc(double q, double r, double c) {
#if DLSA_FUSED_SCONST
    if constexpr (!SC) return fma(q, r, c);
    double d;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(q), "v"(r), "s"(c));
    return d;
#else
    return fma(q, r, c);
#endif
}

constexpr int FP_KC = 32;                 // rows per chunk: two k-steps per wave
constexpr int FP_NST = 4;                 // LDS stages of the fused pass: the DMA runs three chunks ahead of the MFMAs, two ahead of the logistic terms
// ... of the logit-only pass: nothing reads chunk c once its logistic terms are done, so D stages keep D - 1 chunks between
// "requested" and "in use".  Three everywhere: five stages for the one-workgroup shapes (102 instead of 51 KB in flight per CU at
// p = 100) measured no gain (p = 100 1.35-1.40 ms, p = 112 1.54 ms either way), and the two-workgroup shapes (below) fit three twice.
constexpr int fp_logit_stages(int ntc) { (void)ntc; return 3; }
// The logit-only pass evaluates ~100 dependent fp64 operations per 8 rows in ONE wave per SIMD: for short rows (p <= 80) that
// chain, not HBM, sets the pace (p = 50: 57 cycles per row and CU where the stream needs 38).  Its three-stage ring is <= 63 KB
// there, so two workgroups share a CU: two chains per SIMD, twice the bytes in flight.
constexpr int fp_logit_wgs(int ntc) { return ntc <= 5 ? 2 : 1; }
constexpr int FP_MIN_P = 49, FP_MAX_P = 120;
constexpr int64_t FP_MIN_ROWS = 8192;

// BATCHED form (irls_batch.hip: the lock-step driver for many partitions): a workgroup's slab is described by a table entry (FusedSlab,
// irls_batch.h) instead of slab * rows_per_slab -- rows of ONE partition, that partition's own beta, nothing to do when the partition
// has converged.

struct FusedArgs {
    const FusedSlab* slabs;   // nullable: the batched form
    const int* active;        // [partitions]: 0 = the partition's fit has ended (batched form)
    int64_t beta_stride;      // beta of partition k = beta + k * beta_stride (batched form)
    const double* X;
    const double* y;
    const double* beta;
    double* w_out;        // nullable
    double* partial;      // [nslab][PP][PP]
    double* gpart;        // [nslab][GP]: g (16 NTC) | loglik
    int64_t ldx, n, rows_per_slab;
    int p, PP;
    unsigned long long* clk;
};

constexpr int fp_pitch(int ntc) { return (ntc % 2) ? ntc * 16 : ntc * 16 + 16; }       // = 16 mod 32 (gram_narrow.hip)
constexpr int fp_buf(int ntc) { return FP_KC * fp_pitch(ntc) + 2 * FP_KC; }            // a chunk + its w (computed) + its y (DMA)
constexpr int fp_nreg(int nt, int g) { return 8 * (nt * (nt + 1) / 2) + 2 * (nt + 1) * g; }
constexpr int fp_gp(int ntc) { return 16 * ntc + 8; }
// BATCHED logistic terms (fused pass only): five stages + a scratch corner per wave (eight partial sums of 16 rows at a pitch of
// ten doubles -- conflict-free 16-byte reads -- and the 16 residuals)
constexpr int FP_NST_B = 5, FP_SCR = 192;
// ... for shapes whose five stages fit the LDS and whose accumulators leave the VGPRs the batched form needs (<= 190 accumulator registers:
// up to 6 tiles + 1 tail group, p <= 100; the wider shapes' two fragment sets alone are 80-90 VGPRs and keep the per-chunk form)
constexpr bool fp_batch_ok(int nt, int g) {
    return DLSA_FUSED_BATCH && DLSA_FUSED_PRIVATE && fp_nreg(nt, g) <= 190 &&
           (size_t)(FP_NST_B * fp_buf(nt + (g > 0 ? 1 : 0)) + 4 * FP_SCR) * 8 <= (size_t)kLdsBytes;
}
constexpr int fp_hess_stages(int nt, int g) { return fp_batch_ok(nt, g) ? FP_NST_B : FP_NST; }
constexpr size_t fp_hess_lds(int nt, int g) { return (size_t)(fp_hess_stages(nt, g) * fp_buf(nt + (g > 0 ? 1 : 0)) + (fp_batch_ok(nt, g) ? 4 * FP_SCR : 0)) * 8; }

template <int T, int TEND, typename F>
__device__ __forceinline__ void fp_for_tiles(F&& fn) {
    if constexpr (T < TEND) {
        double v[4];
        narrow_tile_read<T>(v);
        fn(T, v);
        fp_for_tiles<T + 1, TEND>(fn);
    }
}
template <int NTRI, int NTAIL, int K, typename F>
__device__ __forceinline__ void fp_for_tails(F&& fn) {
    if constexpr (K < NTAIL) {
        fn(K, narrow_pair_read<8 * NTRI + 2 * K>());
        fp_for_tails<NTRI, NTAIL, K + 1>(fn);
    }
}

// the four waves' partial tiles [T0, T1) meet in LDS (gram_narrow.hip's narrow_meet): waves 1..3 park, wave 0 adds in a fixed order
template <int T0, int T1, int MEETN>
__device__ __forceinline__ void fp_meet(double* lds, int wave, int lane, double* __restrict__ P, int PP) {
    if constexpr (T0 < T1) {
        if (wave != 0)
            fp_for_tiles<T0, T1>([&](int t, double (&v)[4]) {
                double* d = lds + ((wave - 1) * MEETN + (t - T0)) * 256 + lane * 4;
                d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3];
            });
        __syncthreads();
        if (wave == 0)
            fp_for_tiles<T0, T1>([&](int t, double (&v)[4]) {
                int tj = 0;
                while ((tj + 1) * (tj + 2) / 2 <= t) ++tj;
                const int ti = t - tj * (tj + 1) / 2;
                const double* s1 = lds + (t - T0) * 256 + lane * 4;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const double sum = ((v[r] + s1[r]) + s1[MEETN * 256 + r]) + s1[2 * MEETN * 256 + r];
                    P[(int64_t)(ti * 16 + 4 * r + (lane >> 4)) * PP + tj * 16 + (lane & 15)] = sum;
                }
            });
        __syncthreads();
    }
}
template <int NT, int G>
__device__ __forceinline__ void fp_meet_tails(double* lds, int wave, int lane, double* __restrict__ P, int PP) {
    constexpr int NTRI = NT * (NT + 1) / 2, NTAIL = (NT + 1) * G;
    if constexpr (NTAIL > 0) {
        if (wave != 0)
            fp_for_tails<NTRI, NTAIL, 0>([&](int k, double v) { lds[((wave - 1) * NTAIL + k) * 64 + lane] = v; });
        __syncthreads();
        if (wave == 0)
            fp_for_tails<NTRI, NTAIL, 0>([&](int k, double v) {
                const double* s1 = lds + k * 64 + lane;
                const double sum = ((v + s1[0]) + s1[NTAIL * 64]) + s1[2 * NTAIL * 64];
                const int gi = k / (NT + 1), t = k - gi * (NT + 1);
                const int row = 16 * t + 4 * ((lane & 15) >> 2) + (lane >> 4), col = 16 * NT + 4 * gi + (lane & 3);
                P[(int64_t)row * PP + col] = sum;
            });
        __syncthreads();
    }
}

// segments SEG .. NARROW_NSEG - 1 of a k-step's MFMA block, with between(q) issued behind segment q
template <int NT, int G, int SEG, typename F>
__device__ __forceinline__ void fp_kstep_spread(const double (&f)[NT + (G > 0 ? 1 : 0)], const double (&g)[NT],
                                                const double (&bt)[G > 0 ? G : 1], F&& between) {
    if constexpr (SEG < NARROW_NSEG) {
        if constexpr (!(FP_ABL & 8)) narrow_kstep_seg<NT, G, SEG>(f, g, bt);
        else asm volatile("" ::"v"(f[0]), "v"(g[NT - 1]), "v"(bt[0]));
        __builtin_amdgcn_sched_barrier(0);
        between(std::integral_constant<int, SEG>{});
        __builtin_amdgcn_sched_barrier(0);
        fp_kstep_spread<NT, G, SEG + 1>(f, g, bt, between);
    }
}

// The logistic terms of eight rows (lane (j, s): row j, columns 16 q + 2 s + {0, 1}), cut into the pieces that ride behind
// the ten MFMA segments of a chunk.  Everything a later piece needs lives in this struct's registers.
template <int NTC>
struct LogitState {
    double x[NTC][2];         // the lane's columns of its row
    double yv, eta, e, inv, mu, wgt, sv, num, den, q, kf, rr, resid;
    double ps[8];             // batched form: the eight partial sums of the lane's row
    double vmask;             // ICPT: 1.0 for a row of the slab, 0.0 past its end (a factor costs no scalar register pair to keep)
    bool big, valid;
};

// HESS = false: the same streaming skeleton without the Hessian -- the logit pass of narrow designs (w, g, loglik) fed by the
// LDS-DMA ring instead of logit.hip's register loads
// BF: the batched form (its own instantiations: the slab table's extra arguments cost the widest w_out shapes their last scalar registers).
// ICPT: the design is [X | 1] -- the implicit intercept of models.py:121-122 as the LAST column inside the kernel (NT, G are the shape of
// p + 1 columns): the DMA never writes column p of a stage (its lanes end at the even p), so that column is set to 1.0 ONCE; rows past
// the slab's end (zeros through the descriptor's bounds check, but a one in that column) are masked out of w, the residual and the
// log-likelihood by the row's validity instead.  The host side rotates the intercept to the FRONT of g and H (models.py:136-142).
template <bool WOUT, bool HESS, int NT, int G, bool BF = false, bool ICPT = false>
__global__ __launch_bounds__(256, (!HESS ? fp_logit_wgs(NT + (G > 0 ? 1 : 0)) : 1)) void irls_pass_narrow_kernel(FusedArgs a) {
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    typedef double d2 __attribute__((ext_vector_type(2)));
    constexpr int KC = FP_KC, NWAVES = 4, THREADS = 256;
    constexpr int NTC = NT + (G > 0 ? 1 : 0);
    constexpr int LDP = fp_pitch(NTC), BUF = fp_buf(NTC), NTRI = NT * (NT + 1) / 2;
    constexpr int GA = G > 0 ? G : 1;
    constexpr int WOFF = KC * LDP, YOFF = KC * LDP + KC;               // the stage's w and y slots
    constexpr bool PIN = fp_nreg(NT, G) <= 224;                        // registers to spare (all shapes but 7 tiles + 1 or 2 tail groups, 240 / 256 accumulators): pinned constants, scalar coefficients, early fragment loads
    constexpr int DMA_PER_CHUNK = KC / NWAVES + 1;                    // 8 row pieces + the y piece, per wave
    // BATCH: the logistic terms of the wave's own rows of TWO chunks in one evaluation (lane = row (lane & 15), four copies)
    constexpr bool BATCH = HESS && fp_batch_ok(NT, G);
    constexpr int NST = HESS ? fp_hess_stages(NT, G) : fp_logit_stages(NTC);
    constexpr int MEETN_FIT = (int)((size_t)NST * BUF * 8 / (3 * 2048));
    constexpr int MEETN = MEETN_FIT < NTRI ? MEETN_FIT : NTRI;
    static_assert(3 * MEETN >= NTRI, "the tiles meet in at most three passes");
    static_assert((size_t)3 * (NT + 1) * G * 64 <= (size_t)NST * BUF, "tail meeting fits the ring");
    static_assert(NARROW_NSEG == 5 && KC == 32, "piece schedule below: two k-steps of five segments per chunk");
    // (the logistic terms ride behind segments 1 and 3 of the first k-step and 0, 2 and 4 of the second: five VALU groups per chunk --
    // a group costs ~11 pipe cycles whatever its size -- with the LDS reads one segment ahead of their use)
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool probe = blockIdx.x == 0 && wave == 0;
    const unsigned long long t_begin = probe ? __builtin_readcyclecounter() : 0ull;
    const int slab = blockIdx.x;
#if FP_TIMELINE
    unsigned long long tl0 = __builtin_amdgcn_s_memrealtime(), tl1 = 0, tl2 = 0;
#endif
    int64_t rbeg = BF ? 0 : (int64_t)slab * a.rows_per_slab, gxoff, gyoff;
    int nrows;
    const double* __restrict__ beta_in = a.beta;
    if constexpr (BF) {                                                // batched form: the slab is a table entry
        const FusedSlab sd = a.slabs[slab];
        if (a.active && !a.active[sd.part]) return;                    // that partition's fit has ended: nobody reads this slab's partials
        gxoff = sd.xoff; gyoff = sd.yoff; nrows = sd.nrows;
        beta_in += (int64_t)sd.part * a.beta_stride;
    } else {
        const int64_t rend = min(rbeg + a.rows_per_slab, a.n);
        nrows = (int)(rend > rbeg ? rend - rbeg : 0);
        gxoff = rbeg * a.ldx; gyoff = rbeg;
    }
    const int nchunks = (nrows + KC - 1) / KC;
    double* const wout_base = WOUT ? a.w_out + rbeg : nullptr;        // (one pointer instead of a pointer and a row offset kept through the loop)

    const unsigned xbytes = nrows > 0 ? (unsigned)(((int64_t)(nrows - 1) * a.ldx + a.p) * 8) : 0u;
    __amdgpu_buffer_rsrc_t rsrcX = __builtin_amdgcn_make_buffer_rsrc((void*)(a.X + gxoff), 0, (int)xbytes, 0x00020000);
    __amdgpu_buffer_rsrc_t rsrcY = __builtin_amdgcn_make_buffer_rsrc((void*)(a.y + gyoff), 0, nrows * 8, 0x00020000);
    // columns p .. 16 NTC - 1 are never written by the DMA (lanes masked): the ring is zeroed once
    for (int e = tid; e < NST * BUF; e += THREADS) lds[e] = 0.0;
    __syncthreads();
    if constexpr (ICPT) {
        for (int e = tid; e < NST * KC; e += THREADS) lds[(e / KC) * BUF + (e % KC) * LDP + a.p] = 1.0;
        __syncthreads();
    }

    const bool col_in = 2 * lane < a.p;
    constexpr int RQ = KC / NWAVES / 4;                               // rows per wave and DMA part (two)
    // PRIVATE (HESS): a wave stages the rows and labels of ITS OWN two k-steps (rows 4 wave .. + 3 and 16 + 4 wave .. + 3), which are
    // also the rows whose logistic terms it evaluates -- nothing in the chunk loop crosses waves, so the loop carries no s_barrier
    // and the four waves drift freely.  The logit-only ring keeps the round-robin rows and its barrier.
    constexpr bool PRIV = HESS && DLSA_FUSED_PRIVATE;
    auto stage_rows = [&](int chunk, int buf, int q) {
        double* base = lds + buf * BUF;
        // (the row pitch goes through an opaque copy: hipcc otherwise keeps row * pitch for every row of the chunk in SGPRs across the
        // loop -- 16 registers this kernel needs for the polynomial coefficients -- instead of one s_mul per DMA; 32-bit: the slab's
        // bytes fit, irls_pass_fused_eligible)
        int pitch_b = (int)a.ldx * 8, wrow = PRIV ? 4 * wave : wave;       // (the wave's first row likewise: its LDS offsets per row stay out of SGPRs)
        asm volatile("" : "+s"(pitch_b), "+s"(wrow));
#pragma unroll
        for (int ps = q * RQ; ps < (q + 1) * RQ; ++ps) {
            const int rrow = PRIV ? 16 * (ps >> 2) + (ps & 3) : NWAVES * ps;          // row = wrow + rrow
            const int soff = (chunk * KC + wrow + rrow) * pitch_b;
            if (col_in) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcX, (lds_ptr_t)(base + wrow * LDP + rrow * LDP), 16, lane * 16, soff, 0, DLSA_STREAM_AUX);
        }
    };
    // one row piece (BATCH: one DMA instruction behind each MFMA segment -- two in a row cost the pipe four times what two apart do, bench/ubench_gap.hip)
    auto stage_row1 = [&](int chunk, int buf, int ps) {
        double* base = lds + buf * BUF;
        int pitch_b = (int)a.ldx * 8, wrow = 4 * wave;
        asm volatile("" : "+s"(pitch_b), "+s"(wrow));
        const int rrow = 16 * (ps >> 2) + (ps & 3);
        const int soff = (chunk * KC + wrow + rrow) * pitch_b;
        if (col_in) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcX, (lds_ptr_t)(base + wrow * LDP + rrow * LDP), 16, lane * 16, soff, 0, DLSA_STREAM_AUX);
    };
    // the chunk's labels.  PRIVATE: this wave's eight only, into its own corner of the y slot ([wave][8]: lane L fetches the two labels
    // 2 (L & 1) .. + 1 of k-step wave + 4 (L >> 1)).  Else every wave fetches all 32 (same bytes to the same place, the same count).
    auto stage_y = [&](int chunk, int buf) {
        if constexpr (PRIV) {
            if (lane < 4)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcY, (lds_ptr_t)(lds + buf * BUF + YOFF + 8 * wave), 16,
                                                         (4 * (wave + 4 * (lane >> 1)) + 2 * (lane & 1)) * 8, chunk * KC * 8, 0, 0);
        } else {
            if (lane < KC / 2)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcY, (lds_ptr_t)(lds + buf * BUF + YOFF), 16, lane * 16, chunk * KC * 8, 0, 0);
        }
    };

    // ---- the lane's share of beta and of g (columns 16 q + 2 s + {0, 1}); loglik.
    // Lane (j, s): row slot j = 0..7 of the wave's eight rows of a chunk, s = 0..7 the eighth of the row.  s sits on lane bits 0, 1, 3 and
    // j on bits 2, 4, 5, so that the three butterfly steps over a row's lanes are xor 1, xor 2 (quad_perm) and xor 8 (row_ror:8): one
    // DPP move per dword each (xor 4 would need two).
    const int ls = (lane & 3) | ((lane >> 1) & 4), lj = ((lane >> 2) & 1) | ((lane >> 3) & 6);
    double bq[NTC][2], gacc[NTC][2];
#pragma unroll
    for (int q = 0; q < NTC; ++q)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int col = 16 * q + 2 * ls + e;
            if constexpr (ICPT) bq[q][e] = col < a.p ? beta_in[col + 1] : (col == a.p ? beta_in[0] : 0.0);      // beta arrives intercept FIRST
            else bq[q][e] = col < a.p ? beta_in[col] : 0.0;
            gacc[q][e] = 0.0;
        }
    // every lane of a row accumulates the row's loglik term (the eight copies are thinned out after the loop); rows past the slab's
    // end arrive as zeros (x and y: the descriptors' bounds check), add nothing to g and exactly -softplus(0) each to loglik, which
    // the epilogue takes out again -- no per-row validity test in the loop
    double llacc = 0.0;
    // own rows of a chunk: k-step `wave` holds rows 4 wave .. + 3 (j < 4), k-step wave + 4 rows 16 + 4 wave .. + 3 (j >= 4)
    const int own_row = (lj < 4 ? 4 * wave + lj : 16 + 4 * wave + (lj - 4));
    const int own_slot = PRIV ? 8 * wave + lj : own_row;  // the row's place in the stage's w and y slots
    const int xoff = own_row * LDP + 2 * ls;              // + 16 q: the lane's b128 of tile column q
    // leading coefficients of the two polynomials, pinned in VGPRs for the whole kernel (the asm keeps hipcc from re-creating them
    // with a v_mov_b64 per evaluation)
    const double c_exp13 = fp_pin64<PIN>(1.6059043836821613e-10), c_log21 = fp_pin64<PIN>(1.0 / 21.0), c_sqrt2m1 = fp_pin64<PIN>(0.41421356237309503),
                 c_ln2 = fp_pin64<PIN>(6.931471805599453094e-01);
    // (BATCH: the five stage offsets live in SGPRs too -- eight of the log polynomial's coefficients move to pinned VGPRs, of which
    // these shapes have > 100 to spare, instead of spilling scalar registers)
    constexpr bool VC = PIN && HESS && fp_batch_ok(NT, G);
    constexpr bool VC8 = VC;
    const double c_l19 = fp_pin64<VC>(1.0 / 19.0), c_l17 = fp_pin64<VC>(1.0 / 17.0), c_l15 = fp_pin64<VC>(1.0 / 15.0), c_l13 = fp_pin64<VC>(1.0 / 13.0),
                 c_l11 = fp_pin64<VC8>(1.0 / 11.0), c_l9 = fp_pin64<VC8>(1.0 / 9.0), c_l7 = fp_pin64<VC8>(1.0 / 7.0), c_l5 = fp_pin64<VC8>(1.0 / 5.0);
    const int k_half = fp_pin32<PIN>(0x3fe00000), k_one = fp_pin32<PIN>(0x3ff00000), k_mhalf = fp_pin32<PIN>((int)0xbfe00000), k_zero = fp_pin32<PIN>(0),
              k_1p5 = fp_pin32<PIN>(0x3ff80000), k_two = fp_pin32<PIN>(0x40000000);

    LogitState<NTC> L;
    // The pieces below are the operations of logistic.h (exp_neg, logistic_terms) in the same order on the same values -- w and mu
    // come out to the last bit as logit.hip's pass gives them -- written so that every step is ONE VALU instruction: next to fp64
    // MFMAs each VALU instruction, a move included, costs ~4.7 cycles of the matrix pipe and each separate group of them ~11 more
    // (bench/ubench_gap.hip), so coefficients are scalar operands (fp_fma_sc), selects pick one dword, and the pieces form few groups.
    // piece 0: the row's columns and its label out of LDS (chunk `chunk` in stage `buf`)
    auto lp_read = [&](int chunk, int buf) {
        const double* base = lds + buf * BUF;
#pragma unroll
        for (int q = 0; q < NTC; ++q) {
            const d2 v = *(const d2*)(base + xoff + 16 * q);
            L.x[q][0] = v.x; L.x[q][1] = v.y;
        }
        L.yv = base[YOFF + own_slot];
        if constexpr (WOUT || ICPT) L.valid = chunk * KC + own_row < nrows;
        if constexpr (ICPT) L.vmask = __hiloint2double(L.valid ? 0x3ff00000 : 0, 0);
    };
    auto lp_split = [&](double s) {               // the exponent split of exp(-|eta|)
        L.eta = s;
        double aabs;
        asm("v_min_f64 %0, |%1|, %2" : "=v"(aabs) : "v"(s), "s"(745.2));
        double t;
        asm("v_mul_f64 %0, %1, %2" : "=v"(t) : "v"(aabs), "s"(1.4426950408889634));
        L.kf = rint(t);
        double r;
        asm("v_fma_f64 %0, %1, %2, -%3" : "=v"(r) : "v"(L.kf), "s"(6.93147180369123816490e-01), "v"(aabs));
        L.rr = fp_fma_sc2(L.kf, 1.90821492927058770002e-10, r);
    };
    auto lp_dot = [&]() {                         // eta over the 8 lanes of the row (every lane ends with it), then the exponent split
        double s0 = L.x[0][0] * bq[0][0], s1 = L.x[0][1] * bq[0][1];
#pragma unroll
        for (int q = 1; q < NTC; ++q) { s0 = fma(L.x[q][0], bq[q][0], s0); s1 = fma(L.x[q][1], bq[q][1], s1); }
        double s = s0 + s1;
        s += dpp_xor_f64<1>(s);
        s += dpp_xor_f64<2>(s);
        s += dpp_xor_f64<8>(s);
        lp_split(s);
    };
    auto lp_exp = [&]() {                         // e = exp(-|eta|): degree-13 polynomial, ldexp (logistic.h: exp_neg)
        const double r = L.rr;
        double q = c_exp13;
        q = fp_fma_sc<PIN>(q, r, 2.08767569878681e-09);
        q = fp_fma_sc<PIN>(q, r, 2.505210838544172e-08);
        q = fp_fma_sc<PIN>(q, r, 2.755731922398589e-07);
        q = fp_fma_sc<PIN>(q, r, 2.7557319223985893e-06);
        q = fp_fma_sc<PIN>(q, r, 2.48015873015873e-05);
        q = fp_fma_sc<PIN>(q, r, 1.984126984126984e-04);
        q = fp_fma_sc<PIN>(q, r, 1.388888888888889e-03);
        q = fp_fma_sc<PIN>(q, r, 8.333333333333333e-03);
        q = fp_fma_sc<PIN>(q, r, 4.1666666666666664e-02);
        q = fp_fma_sc<PIN>(q, r, 1.6666666666666666e-01);
        q = fma(q, r, 0.5);
        q = fma(q, r, 1.0);
        q = fma(q, r, 1.0);
        int kneg;
        asm("v_cvt_i32_f64 %0, -%1" : "=v"(kneg) : "v"(L.kf));
        L.e = ldexp(q, kneg);
    };
    auto lp_mu_core = [&]() {
        const double inv = rcp_newton(1.0 + L.e);
        const double einv = L.e * inv;
        L.mu = L.eta >= 0.0 ? inv : einv;
        L.wgt = einv * inv;
        L.resid = L.yv - L.mu;
        if constexpr (ICPT) {                     // a row past the slab's end is all zeros but for the ones column: it must weigh nothing
            L.wgt *= L.vmask;
            L.resid *= L.vmask;
        }
    };
    auto lp_mu = [&](int buf) {                   // mu, w -> the stage's w slot (read back by this wave's MFMA part next chunk)
        lp_mu_core();
        if (HESS && ls == 0) lds[buf * BUF + WOFF + own_slot] = L.wgt;
    };
    auto lp_log_core = [&]() {                    // log1p(e) and the row's loglik term (logistic.h: logistic_terms)
        // t' = h (1 + e) with h = 1/2 above sqrt 2: num = t' - 1 = h e + (h - 1), den = t' + 1 = h e + (h + 1) -- the same values as
        // logistic.h's selects between fma(0.5, e, -0.5) / e and fma(0.5, e, 1.5) / 2 + e, from three one-dword selects
        const bool big = L.e > c_sqrt2m1;
        const double h = __hiloint2double(big ? k_half : k_one, 0);
        const double hm1 = __hiloint2double(big ? k_mhalf : k_zero, 0);
        const double hp1 = __hiloint2double(big ? k_1p5 : k_two, 0);
        const double num = fma(h, L.e, hm1), den = fma(h, L.e, hp1);
        const double sv = num * rcp_newton(den);
        const double z = sv * sv;
        double q = c_log21;
        if constexpr (VC) {
            q = fma(q, z, c_l19); q = fma(q, z, c_l17); q = fma(q, z, c_l15); q = fma(q, z, c_l13);
            if constexpr (VC8) { q = fma(q, z, c_l11); q = fma(q, z, c_l9); q = fma(q, z, c_l7); q = fma(q, z, c_l5); }
            else {
                q = fp_fma_sc<PIN>(q, z, 1.0 / 11.0); q = fp_fma_sc<PIN>(q, z, 1.0 / 9.0);
                q = fp_fma_sc<PIN>(q, z, 1.0 / 7.0); q = fp_fma_sc<PIN>(q, z, 1.0 / 5.0);
            }
        } else {
            q = fp_fma_sc<PIN>(q, z, 1.0 / 19.0);
            q = fp_fma_sc<PIN>(q, z, 1.0 / 17.0);
            q = fp_fma_sc<PIN>(q, z, 1.0 / 15.0);
            q = fp_fma_sc<PIN>(q, z, 1.0 / 13.0);
            q = fp_fma_sc<PIN>(q, z, 1.0 / 11.0);
            q = fp_fma_sc<PIN>(q, z, 1.0 / 9.0);
            q = fp_fma_sc<PIN>(q, z, 1.0 / 7.0);
            q = fp_fma_sc<PIN>(q, z, 1.0 / 5.0);
        }
        q = fp_fma_sc<PIN>(q, z, 1.0 / 3.0);
        q = fma(q, z, 1.0);
        const double l1p = fma(sv + sv, q, big ? c_ln2 : 0.0);
        double pos;
        asm("v_max_f64 %0, %1, 0" : "=v"(pos) : "v"(L.eta));              // (fmax() puts a canonicalising v_max in front)
        const double softplus = pos + l1p;
        if constexpr (ICPT) llacc = fma(fma(L.yv, L.eta, -softplus), L.vmask, llacc);
        else llacc += fma(L.yv, L.eta, -softplus);
    };
    auto lp_log = [&](int chunk) {
        lp_log_core();
        if constexpr (WOUT) {
            if (ls == 0 && L.valid) wout_base[(int64_t)chunk * KC + own_row] = L.wgt;
        }
    };
    auto lp_grad = [&]() {
#pragma unroll
        for (int q = 0; q < NTC; ++q) {
            gacc[q][0] = fma(L.resid, L.x[q][0], gacc[q][0]);
            gacc[q][1] = fma(L.resid, L.x[q][1], gacc[q][1]);
        }
    };
    auto logit_all = [&](int chunk, int buf) {    // the whole of it in one go (prologue: chunk 0; the logit-only ring)
        lp_read(chunk, buf); lp_dot(); lp_exp(); lp_mu(buf); lp_log(chunk); lp_grad();
    };

    // ---- BATCH: the same terms for the wave's own rows of TWO chunks at once.  The dot products keep the 8-lanes-per-row layout (one
    // pass per chunk) and leave their eight partial sums per row in the wave's scratch corner; then lane L evaluates row (L & 15) --
    // rows 0..7: the first chunk's, 8..15: the second's; four copies -- summing the partials in the butterfly's order (the same eta to
    // the last bit), so the ~64 instructions of the transcendentals run once per 16 rows instead of once per 8.  w goes to the stages'
    // w slots, the residuals to the scratch corner, from where the gradient passes (rows re-read from LDS: DS instructions cost the
    // matrix pipe next to nothing) pick them up.  Everything stays inside the wave: no barrier, no flag.
    double* const scr = lds + (BATCH ? NST * BUF + wave * FP_SCR : 0);
    const int b_r = lane & 15, b_h = b_r >> 3, b_s = b_r & 7;
    const int b_row = b_s < 4 ? 4 * wave + b_s : 16 + 4 * wave + (b_s - 4);
    const int b_slot = 8 * wave + b_s;
    double resid_g[2] = {0.0, 0.0};
    double xb[BATCH ? NTC : 1][2];            // the second chunk's row (the first chunk's sits in L.x): both stay in registers up to the gradient
    auto bp_read = [&](auto hc, int buf) {
        constexpr int h = decltype(hc)::value;
        const double* base = lds + buf * BUF;
#pragma unroll
        for (int q = 0; q < NTC; ++q) {
            const d2 v = *(const d2*)(base + xoff + 16 * q);
            if constexpr (h == 0) { L.x[q][0] = v.x; L.x[q][1] = v.y; }
            else { xb[BATCH ? q : 0][0] = v.x; xb[BATCH ? q : 0][1] = v.y; }
        }
    };
    auto bp_dot = [&]() {                         // both chunks' partial sums -> the scratch corner
        double s0 = L.x[0][0] * bq[0][0], s1 = L.x[0][1] * bq[0][1];
        double t0 = xb[0][0] * bq[0][0], t1 = xb[0][1] * bq[0][1];
#pragma unroll
        for (int q = 1; q < NTC; ++q) {
            s0 = fma(L.x[q][0], bq[q][0], s0); s1 = fma(L.x[q][1], bq[q][1], s1);
            t0 = fma(xb[BATCH ? q : 0][0], bq[q][0], t0); t1 = fma(xb[BATCH ? q : 0][1], bq[q][1], t1);
        }
        scr[lj * 10 + ls] = s0 + s1;
        scr[(8 + lj) * 10 + ls] = t0 + t1;
    };
    auto bp_sum_read = [&](int chunk0, int buf0, int buf1) {
        const d2* pr = (const d2*)(scr + b_r * 10);
#pragma unroll
        for (int k = 0; k < 4; ++k) { const d2 v = pr[k]; L.ps[2 * k] = v.x; L.ps[2 * k + 1] = v.y; }
        L.yv = lds[(b_h ? buf1 : buf0) * BUF + YOFF + b_slot];
        if constexpr (WOUT || ICPT) L.valid = (chunk0 + b_h) * KC + b_row < nrows;
        if constexpr (ICPT) L.vmask = __hiloint2double(L.valid ? 0x3ff00000 : 0, 0);
    };
    auto bp_sum = [&]() {
        lp_split(((L.ps[0] + L.ps[1]) + (L.ps[2] + L.ps[3])) + ((L.ps[4] + L.ps[5]) + (L.ps[6] + L.ps[7])));
    };
    auto bp_mu = [&](int buf0, int buf1) {
        lp_mu_core();
        if (lane < 16) {
            lds[(b_h ? buf1 : buf0) * BUF + WOFF + b_slot] = L.wgt;
            scr[160 + b_r] = L.resid;
        }
    };
    auto bp_log = [&](int chunk0) {
        lp_log_core();
        if constexpr (WOUT) {
            if (lane < 16 && L.valid) wout_base[(int64_t)(chunk0 + b_h) * KC + b_row] = L.wgt;
        }
    };
    auto bp_gread = [&]() { resid_g[0] = scr[160 + lj]; resid_g[1] = scr[168 + lj]; };
    auto bp_grad = [&]() {
#pragma unroll
        for (int q = 0; q < NTC; ++q) {
            gacc[q][0] = fma(resid_g[0], L.x[q][0], gacc[q][0]);
            gacc[q][1] = fma(resid_g[0], L.x[q][1], gacc[q][1]);
        }
#pragma unroll
        for (int q = 0; q < NTC; ++q) {
            gacc[q][0] = fma(resid_g[1], xb[BATCH ? q : 0][0], gacc[q][0]);
            gacc[q][1] = fma(resid_g[1], xb[BATCH ? q : 0][1], gacc[q][1]);
        }
    };

    if constexpr (HESS) narrow_acc_zero<fp_nreg(NT, G)>();

    // ---- prologue: chunks 0 .. 2 in flight (logit only: 0 .. D - 1); chunks 0 and 1 landed; the logistic terms of chunk 0
    // (BATCH: chunks 0 .. 4 in flight -- all five stages --, 0 and 1 landed, their logistic terms)
    constexpr int D = BATCH ? 5 : HESS ? 3 : fp_logit_stages(NTC);
#pragma unroll
    for (int ch = 0; ch < D; ++ch) {
#pragma unroll
        for (int q = 0; q < 4; ++q) stage_rows(ch, ch, q);
        stage_y(ch, ch);
    }
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 2) * DMA_PER_CHUNK) : "memory");
    if constexpr (!PRIV) asm volatile("s_barrier" ::: "memory");
    if constexpr (BATCH) {
        bp_read(std::integral_constant<int, 0>{}, 0); bp_read(std::integral_constant<int, 1>{}, 1); bp_dot();
        bp_sum_read(0, 0, 1); bp_sum(); lp_exp(); bp_mu(0, 1); bp_log(0);
        bp_gread(); bp_grad();
    } else {
        logit_all(0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (!HESS) asm volatile("s_barrier" ::: "memory");       // the loop's first DMA overwrites chunk 0, which every wave must have left

#if FP_TIMELINE
    tl1 = __builtin_amdgcn_s_memrealtime();
#endif
    const int frag_off = (lane >> 4) * LDP + (lane & 15);
    const int tail_off = (lane >> 4) * LDP + 16 * NT + (lane & 3);
    int cur = 0;
    if constexpr (!HESS) {
        for (int c = 0; c < nchunks; ++c) {
            const int nxt = cur == D - 1 ? 0 : cur + 1;                 // D stages: chunk c + D replaces chunk c, done with
#pragma unroll
            for (int q = 0; q < 4; ++q) stage_rows(c + D, cur, q);
            stage_y(c + D, cur);
            __builtin_amdgcn_sched_barrier(0);
            logit_all(c + 1, nxt);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 2) * DMA_PER_CHUNK) : "memory");     // chunk c + 2 has landed
            asm volatile("s_barrier" ::: "memory");
            cur = nxt;
        }
    }
    // The fragments of chunk c + 1 (landed since the last barrier) and the weights just computed for it are requested at the END of
    // iteration c, in front of the barrier: the LDS latency passes while the wave waits there, and the next iteration starts on its
    // MFMAs at once.  Two register sets, the loop unrolled over them (no copies).
    struct Frags { double f[2][NTC], wv[2], bt[2][GA]; };
    auto load_frags = [&](int buf, Frags& fr) {
        const double* base = lds + buf * BUF;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int ks = wave + 4 * kk;
            const double* kb = base + ks * 4 * LDP;
#pragma unroll
            for (int t = 0; t < NTC; ++t) fr.f[kk][t] = kb[frag_off + t * 16];
#pragma unroll
            for (int gi = 0; gi < G; ++gi) fr.bt[kk][gi] = kb[tail_off + 4 * gi];
            fr.wv[kk] = base[WOFF + (PRIV ? 8 * wave + 4 * kk : ks * 4) + (lane >> 4)];
        }
    };
    auto iteration = [&](int c, const Frags& fr, Frags& fnext) {
        const int nxt = (cur + 1) & 3, nxt3 = (cur + 3) & 3;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            double g[NT], btw[GA];
#pragma unroll
            for (int t = 0; t < NT; ++t) g[t] = fr.f[kk][t] * fr.wv[kk];
#pragma unroll
            for (int gi = 0; gi < GA; ++gi) btw[gi] = (G > 0) ? fr.bt[kk][gi] * fr.wv[kk] : 0.0;
            if (kk == 0) {
                // behind the five segments of the first k-step: the DMA of chunk c + 3 in five parts (every wave has left chunk c - 1,
                // whose stage it overwrites) and the first half of the logistic terms of chunk c + 1
                fp_kstep_spread<NT, G, 0>(fr.f[kk], g, btw, [&](auto qc) {
                    constexpr int q = decltype(qc)::value;
                    if constexpr (!(FP_ABL & 4)) { if constexpr (q < 4) stage_rows(c + 3, nxt3, q); else stage_y(c + 3, nxt3); }
                    if constexpr (FP_ABL & 2) return;
                    if constexpr (q == 0) lp_read(c + 1, nxt);        // (LDS reads only: their latency passes under segment 1)
                    else if constexpr (q == 1) lp_dot();
                    else if constexpr (q == 3) lp_exp();
                });
            } else {
                fp_kstep_spread<NT, G, 0>(fr.f[kk], g, btw, [&](auto qc) {
                    constexpr int q = decltype(qc)::value;
                    // the NEXT chunk's fragments and the weights lp_mu has just stored for it are requested here, three segments before
                    // the iteration ends: without a barrier to wait at, a request at the very end would be waited for with the pipe idle
                    if constexpr (q == 1 && DLSA_FUSED_EARLY_FRAGS && PIN) load_frags(nxt, fnext);
                    if constexpr (FP_ABL & 2) return;
                    if constexpr (q == 0) lp_mu(nxt);
                    else if constexpr (q == 2) { if constexpr (!(FP_ABL & 1)) lp_log(c + 1); }
                    else if constexpr (q == 4) lp_grad();
                });
            }
        }
        if constexpr (!(DLSA_FUSED_EARLY_FRAGS && PIN)) load_frags(nxt, fnext);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DMA_PER_CHUNK) : "memory");      // chunk c + 2 has landed (c + 3 is in flight)
        if constexpr (!PRIV) asm volatile("s_barrier" ::: "memory");
        cur = nxt;
    };
    // BATCH: one trip = the MFMAs of chunks c and c + 1 (stages s0, s1), the logistic terms of chunks c + 2 and c + 3 (landed / landing
    // in s2, s3), the DMA of chunks c + 5 and c + 6.  A stage is dead as soon as its chunk's fragments sit in registers (the wave's two
    // k-steps ARE its rows of the chunk; their logistic terms were done a trip earlier): chunk c + 5 goes into chunk c's stage during
    // the MFMAs of chunk c, chunk c + 6 into chunk c + 1's during the MFMAs of chunk c + 1 -- every row is requested more than a
    // trip (> 3.5 us) before its first use (four chunks ahead left the odd chunk ~2 us: SQ_WAIT_ANY 6.7 % of the wave cycles).
    auto half = [&](auto hc, int c, int s0, int s1, int s2, int s3, int s4, const Frags& fr, Frags& fnext) {
        constexpr int HALF = decltype(hc)::value;                       // 0: the MFMAs of chunk c (stage s0), 1: of chunk c + 1 (s1)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            double g[NT], btw[GA];
#pragma unroll
            for (int t = 0; t < NT; ++t) g[t] = fr.f[kk][t] * fr.wv[kk];
#pragma unroll
            for (int gi = 0; gi < GA; ++gi) btw[gi] = (G > 0) ? fr.bt[kk][gi] * fr.wv[kk] : 0.0;
            fp_kstep_spread<NT, G, 0>(fr.f[kk], g, btw, [&](auto qc) {
                constexpr int q = decltype(qc)::value;
                // chunk c + 3 has landed when only chunk c + 4 and the two row pieces of chunk c + 5 just issued are still out
                if (kk == 0) { if constexpr (HALF == 0 && q == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((FP_ABL & 4) ? 0 : DMA_PER_CHUNK + 2) : "memory"); }
                if constexpr (!(FP_ABL & 4)) {      // the eight rows and the labels of chunk c + 5 (+ 1): one piece behind each of nine segments
                    const int piece = 5 * kk + q;
                    if (piece < 8) stage_row1(c + 5 + HALF, HALF == 0 ? s0 : s1, piece);
                    else if (piece == 8) stage_y(c + 5 + HALF, HALF == 0 ? s0 : s1);
                }
                // the next MFMA block's fragments (and the weights bp_mu has stored for them), three segments before they are needed
                if (kk == 1) { if constexpr (q == 1) load_frags(HALF == 0 ? s1 : s2, fnext); }
                if constexpr (FP_ABL & 2) return;
                if constexpr (HALF == 0) {
                    // (few, large VALU groups: a group costs ~11 pipe cycles whatever its size; every LDS read one segment ahead of its use)
                    if (kk == 0) {
                        if constexpr (q == 0) bp_read(std::integral_constant<int, 0>{}, s2);
                        else if constexpr (q == 2) bp_read(std::integral_constant<int, 1>{}, s3);
                        else if constexpr (q == 3) bp_dot();
                        else if constexpr (q == 4) bp_sum_read(c + 2, s2, s3);
                    } else {
                        if constexpr (q == 0) { bp_sum(); lp_exp(); }
                        else if constexpr (q == 2) bp_mu(s2, s3);
                        else if constexpr (q == 4) { if constexpr (!(FP_ABL & 1)) bp_log(c + 2); }
                    }
                } else {
                    if (kk == 0) {
                        if constexpr (q == 1) bp_gread();
                        else if constexpr (q == 2) bp_grad();
                    }
                }
            });
        }
    };
    if constexpr (HESS && BATCH) {
        Frags fa, fb;
        load_frags(0, fa);
        int s0 = 0;
        // an odd chunk count runs one chunk past the slab's end: zero rows through the DMA's bounds check, taken out of the logistic sums
        for (int c = 0; c < nchunks; c += 2) {
            const int s1 = s0 + 1 - (s0 >= 4 ? 5 : 0), s2 = s0 + 2 - (s0 >= 3 ? 5 : 0), s3 = s0 + 3 - (s0 >= 2 ? 5 : 0), s4 = s0 + 4 - (s0 >= 1 ? 5 : 0);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((FP_ABL & 4) ? 0 : 2 * DMA_PER_CHUNK) : "memory");       // chunk c + 2 has landed (c + 3, c + 4 may be in flight)
            half(std::integral_constant<int, 0>{}, c, s0, s1, s2, s3, s4, fa, fb);
            half(std::integral_constant<int, 1>{}, c, s0, s1, s2, s3, s4, fb, fa);
            s0 = s2;
        }
    } else if constexpr (HESS) {
        Frags fa, fb;
        load_frags(0, fa);
        // an odd chunk count runs one chunk past the slab's end: zero rows through the DMA's bounds check, masked in the logistic sums
        for (int c = 0; c < nchunks; c += 2) {
            iteration(c, fa, fb);
            iteration(c + 1, fb, fa);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
#if FP_TIMELINE
    tl2 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long tl2w = tl2;
#endif
    __syncthreads();

    // ---- H: the four waves' triangles meet in LDS; wave 0 stores the slab's partial
    if constexpr (HESS) {
        double* __restrict__ P = a.partial + (int64_t)slab * a.PP * a.PP;
        fp_meet<0, (MEETN < NTRI ? MEETN : NTRI), MEETN>(lds, wave, lane, P, a.PP);
        fp_meet<MEETN, (2 * MEETN < NTRI ? 2 * MEETN : NTRI), MEETN>(lds, wave, lane, P, a.PP);
        fp_meet<2 * MEETN, NTRI, MEETN>(lds, wave, lane, P, a.PP);
        fp_meet_tails<NT, G>(lds, wave, lane, P, a.PP);
    }

    // ---- g and loglik: over the 8 rows of a wave (lanes with the same s: the row bits are lane bits 2, 4, 5), then over the four waves in LDS
#pragma unroll
    for (int q = 0; q < NTC; ++q)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            double s = gacc[q][e], u, v;
            s += dpp_xor_f64<4>(s);
            swap_f64<16>(s, s, u, v); s = u + v;
            swap_f64<32>(s, s, u, v); s = u + v;
            gacc[q][e] = s;
        }
    // loglik: one lane per row counts.  The zero rows evaluated past the slab's end -- chunks 0 .. nch_eval - 1 were evaluated -- each
    // added fma(0, 0, -softplus(0)) = -ln 2: taken out again (to the rounding of that one product).
    if constexpr (ICPT) {
        llacc = (BATCH ? lane < 16 : ls == 0) ? llacc : 0.0;          // (invalid rows added nothing: nothing to take out)
    } else if constexpr (BATCH) {
        // lane L < 16 evaluated row b_row of the chunks b_h, b_h + 2, ... below nch_eval (even)
        const int nch_eval = ((nchunks + 1) & ~1) + 2;
        int first_bad = nrows > b_row ? (nrows - b_row + KC - 1) / KC : 0;
        first_bad += (first_bad ^ b_h) & 1;                                                // ... of the lane's parity
        const int nbad = nch_eval > first_bad ? (nch_eval - first_bad + 1) / 2 : 0;
        llacc = lane < 16 ? fma((double)nbad, 6.931471805599453094e-01, llacc) : 0.0;
    } else {
        const int nch_eval = (HESS ? ((nchunks + 1) & ~1) : nchunks) + 1;
        const int first_bad = nrows > own_row ? (nrows - own_row + KC - 1) / KC : 0;      // first chunk whose row `own_row` lies past the end
        const int nbad = nch_eval > first_bad ? nch_eval - first_bad : 0;
        llacc = ls == 0 ? fma((double)nbad, 6.931471805599453094e-01, llacc) : 0.0;
    }
    llacc = wave_allreduce_sum(llacc);
    constexpr int GP = fp_gp(NTC);
    if (lj == 0) {                                // the eight lanes of row slot 0 hold the wave's sums for their s
#pragma unroll
        for (int q = 0; q < NTC; ++q) {
            lds[wave * GP + 16 * q + 2 * ls] = gacc[q][0];
            lds[wave * GP + 16 * q + 2 * ls + 1] = gacc[q][1];
        }
        if (ls == 0) lds[wave * GP + 16 * NTC] = llacc;
    }
    __syncthreads();
    if (tid <= 16 * NTC)
        a.gpart[(int64_t)slab * GP + tid] = ((lds[tid] + lds[GP + tid]) + lds[2 * GP + tid]) + lds[3 * GP + tid];
    if (probe && lane == 0) *a.clk = __builtin_readcyclecounter() - t_begin;
#if FP_TIMELINE
    if (lane == 0) {       // slots 16 NTC + 1 .. + 7 of the workgroup's g partial are free: wave 0 leaves start / prologue end / loop end / end, waves 1..3 their loop ends
        unsigned long long* tl = (unsigned long long*)(a.gpart + (int64_t)slab * GP + 16 * NTC + 1);
        if (wave == 0) { tl[0] = tl0; tl[1] = tl1; tl[2] = tl2w; tl[3] = __builtin_amdgcn_s_memrealtime(); }
        else tl[3 + wave] = tl2w;
    }
#endif
}

// g [p] and loglik: the slab partials in a fixed order
// (rot: the kernel's LAST column -- the implicit intercept -- is entry 0 of g, the others move up by one: models.py:136-142)
__global__ __launch_bounds__(256) void irls_pass_finish_kernel(const double* __restrict__ gpart, int nslab, int GP, int p, int ll_at,
                                                               double* __restrict__ g, double* __restrict__ loglik, int rot) {
    __shared__ double red[16][17];
    const int cl = threadIdx.x & 15, kg = threadIdx.x >> 4;
    const int col = blockIdx.x * 16 + cl;
    const bool is_ll = col == p;                  // one extra "column": the log-likelihood
    const int src = is_ll ? ll_at : col;
    double s = 0.0;
    if (col <= p)
        for (int k = kg; k < nslab; k += 16) s += gpart[(int64_t)k * GP + src];
    red[kg][cl] = s;
    __syncthreads();
    if (kg == 0 && col <= p) {
        double t = red[0][cl];
#pragma unroll
        for (int k = 1; k < 16; ++k) t += red[k][cl];
        if (is_ll) { if (loglik) *loglik = t; }
        else if (g) g[rot ? (col == p - 1 ? 0 : col + 1) : col] = t;
    }
}

static void fp_shape(int p, int& nt, int& g) {
    nt = p / 16;
    g = (p - 16 * nt + 3) / 4;
    if (g == 4) { ++nt; g = 0; }
}

static int fp_slabs(int64_t n, int64_t& rows_per_slab, int wgs_per_cu = 1) {
    int64_t ns = std::min<int64_t>((int64_t)kNumCU * wgs_per_cu, std::max<int64_t>(1, n / 2048));
    rows_per_slab = ((n + ns - 1) / ns + FP_KC - 1) / FP_KC * FP_KC;
    return (int)((n + rows_per_slab - 1) / rows_per_slab);
}

static size_t fp_pp(int p) { return ((size_t)(p + 15) / 16 * 16 + 63) / 64 * 64; }

// Rows the 16-byte DMA pieces may stream.  Even p: rows of even pitch from a 16-byte aligned base (every piece aligned, none leaves its
// row).  ODD p (round 5; dummy-encoded widths are arbitrary, models.py:56-104): the piece of the row's last column also carries the
// 8 bytes behind it -- harmless when those are DATA: a packed matrix (ldx == p), where they are the next row's first element (finite
// whenever the data is; beta is zero there, and H's column p is never written out) or, behind the slab's last row, out of the
// descriptor's range (zero).  A padded odd-width row would put the caller's padding bytes there (possibly NaN): not served.  The
// pieces of such rows sit at 8-byte offsets, which the LDS-DMA takes (bench: p = 99, 101, 111 exact against the oracle).
static bool fp_rows_ok(const double* X, int64_t ldx, int p, int64_t base_ldx = 0) {
    // (base_ldx: the matrix's own row pitch when `ldx` is the pitch of a strided VIEW of it -- rows k, k + K, ...: the bytes behind a
    // row of the view are the next row of the matrix)
    if (p & 1) return (base_ldx ? base_ldx : ldx) == p && ((uintptr_t)X % 8) == 0;
    return ldx % 2 == 0 && ((uintptr_t)X % 16) == 0;
}

bool irls_pass_fused_eligible(const double* X, int64_t ldx, const double* y, int64_t n, int p) {
    if (p < FP_MIN_P || p > FP_MAX_P || n < FP_MIN_ROWS) return false;
    const char* e = knob("DLSA_IRLS_FUSED");
    if (e && atoi(e) == 0) return false;          // A/B runs: the two-launch form
    int64_t rps;
    fp_slabs(n, rps);
    return fp_rows_ok(X, ldx, p) && ((uintptr_t)y % 8) == 0 &&          // (the y pieces are dword-aligned buffer loads)
           (double)(rps + 8 * FP_KC) * (double)ldx * 8.0 < 2.0e9;                    // 32-bit DMA offsets
}

static size_t fp_fused_ws_bytes(int64_t n, int p) {
    int nt, g;
    fp_shape(p, nt, g);
    const int ntc = nt + (g > 0 ? 1 : 0);
    // first estimates of fp_slabs: upper bounds of its results that are monotone in n (one workspace serves every row count up to n)
    const int64_t base = std::max<int64_t>(1, n / 2048);
    const int ns = (int)std::min<int64_t>(kNumCU, base);
    const int ns_logit = (int)std::min<int64_t>((int64_t)kNumCU * fp_logit_wgs(ntc), base);    // the logit-only launch of short rows: two workgroups per CU
    return align_up((size_t)ns * fp_pp(p) * fp_pp(p) * 8, 256) + align_up((size_t)std::max(ns, ns_logit) * fp_gp(ntc) * 8, 256) + kGramProbeBytes;
}

size_t irls_pass_workspace_bytes_impl(int64_t n, int p) {
    size_t b = std::max(gram_workspace_bytes_impl(n, p, 8), logit_workspace_bytes_impl(n, p));
    if (p >= FP_MIN_P && p <= FP_MAX_P && n >= FP_MIN_ROWS) b = std::max(b, fp_fused_ws_bytes(n, p));
    return b;
}

// One Newton pass: the fused kernel where the shape allows it, else the logit pass + the Gram pass (two launches, w through
// `w_out` or scratch).  `w_out` nullable (w_scratch [n] then carries w between the two launches of the unfused form).
int irls_pass_impl(const double* X, int64_t ldx, const double* y, const double* beta, int64_t n, int p, double* H, int64_t ldh,
                   double* g, double* loglik, double* w_out, double* w_scratch, void* ws, size_t ws_bytes, hipStream_t stream,
                   int* fused_out) {
    DLSA_REQUIRE((X || n == 0) && (y || n == 0) && beta, "irls_pass: null argument");      // H == nullptr: the logit pass alone
    DLSA_REQUIRE(p > 0 && p <= 2048 && n >= 0 && ldx >= p && ldh >= p, "irls_pass: bad shape n=%lld p=%d ldx=%lld ldh=%lld",
                 (long long)n, p, (long long)ldx, (long long)ldh);
    const bool fused = irls_pass_fused_eligible(X, ldx, y, n, p) && (!w_out || ((uintptr_t)w_out % 8) == 0);
    if (fused_out) *fused_out = fused ? 1 : 0;
    if (!fused) {
        double* w = w_out ? w_out : w_scratch;
        DLSA_REQUIRE(w || n == 0, "irls_pass: this shape takes the two-launch form, which needs w_out (or scratch) for the weights");
        int rc = logit_pass_impl(X, ldx, y, beta, n, p, w, g, loglik, ws, ws_bytes, stream, 0);
        if (rc || !H) return rc;
        return gram_impl_f64(X, ldx, w, n, p, H, ldh, 0, ws, ws_bytes, stream);
    }
    FusedArgs a;
    a.slabs = nullptr; a.active = nullptr; a.beta_stride = 0;
    a.X = X; a.y = y; a.beta = beta; a.w_out = w_out; a.ldx = ldx; a.n = n; a.p = p; a.PP = (int)fp_pp(p);
    int nt, gt;
    fp_shape(p, nt, gt);
    const int ntc = nt + (gt > 0 ? 1 : 0), GP = fp_gp(ntc);
    const int nslab = fp_slabs(n, a.rows_per_slab, H ? 1 : fp_logit_wgs(ntc));
    const size_t part = H ? align_up((size_t)nslab * a.PP * a.PP * 8, 256) : 0, gpb = align_up((size_t)nslab * GP * 8, 256);
    const size_t need = part + gpb + kGramProbeBytes;
    if (!ws || ws_bytes < need || ((uintptr_t)ws & 255)) {
        set_error("irls_pass: workspace %zu bytes needed (256-aligned), got %zu", need, ws_bytes);
        return DLSA_ERR_WORKSPACE;
    }
    a.partial = (double*)ws;
    a.gpart = (double*)((char*)ws + part);
    a.clk = (unsigned long long*)((char*)ws + part + gpb);
#define DLSA_LAUNCH_FP2(WO, HS, NTV, GV) do { \
        const size_t shm = (HS) ? fp_hess_lds(NTV, GV) : (size_t)fp_logit_stages(NTV + (GV > 0 ? 1 : 0)) * fp_buf(NTV + (GV > 0 ? 1 : 0)) * 8; \
        DLSA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(irls_pass_narrow_kernel<WO, HS, NTV, GV>), \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm)); \
        hipLaunchKernelGGL((irls_pass_narrow_kernel<WO, HS, NTV, GV>), dim3(nslab), dim3(256), shm, stream, a); } while (0)
#define DLSA_LAUNCH_FP(WO, NTV, GV) do { if (H) DLSA_LAUNCH_FP2(WO, true, NTV, GV); else DLSA_LAUNCH_FP2(WO, false, NTV, GV); } while (0)
#define DLSA_LAUNCH_FP_G(WO, NTV) do { switch (gt) { \
        case 0: DLSA_LAUNCH_FP(WO, NTV, 0); break; case 1: DLSA_LAUNCH_FP(WO, NTV, 1); break; \
        case 2: DLSA_LAUNCH_FP(WO, NTV, 2); break; default: DLSA_LAUNCH_FP(WO, NTV, 3); break; } } while (0)
#define DLSA_LAUNCH_FP_NT(WO) do { switch (nt) { \
        case 3: DLSA_LAUNCH_FP_G(WO, 3); break; case 4: DLSA_LAUNCH_FP_G(WO, 4); break; \
        case 5: DLSA_LAUNCH_FP_G(WO, 5); break; case 6: DLSA_LAUNCH_FP_G(WO, 6); break; \
        default: switch (gt) { case 0: DLSA_LAUNCH_FP(WO, 7, 0); break; case 1: DLSA_LAUNCH_FP(WO, 7, 1); break; \
                               default: DLSA_LAUNCH_FP(WO, 7, 2); break; } break; } } while (0)
    if (w_out) DLSA_LAUNCH_FP_NT(true);
    else DLSA_LAUNCH_FP_NT(false);
#undef DLSA_LAUNCH_FP_NT
#undef DLSA_LAUNCH_FP_G
#undef DLSA_LAUNCH_FP
#undef DLSA_LAUNCH_FP2
    DLSA_HIP_CHECK(hipGetLastError());
    note_gram_kernel(a.clk, stream, "irls_pass_narrow_kernel<%s,%s,%d,%d>", w_out ? "true" : "false", H ? "true" : "false", nt > 6 ? 7 : nt,
                     nt > 6 ? (gt > 2 ? 2 : gt) : gt);
    if (H) gram_reduce_launch<double>((const double*)ws, nslab, a.PP, p, H, ldh, 0, stream);
    hipLaunchKernelGGL(irls_pass_finish_kernel, dim3((p + 1 + 15) / 16), dim3(256), 0, stream, (const double*)a.gpart, nslab, GP, p,
                       16 * ntc, g, loglik, 0);
    DLSA_HIP_CHECK(hipGetLastError());
    return DLSA_OK;
}

// ---- the fused pass for a design with the IMPLICIT intercept (dlsa_irls_fit_ex_f64 with intercept != 0; models.py:121-122): p data
// columns, p + 1 = pe columns of beta / g / H with the intercept FIRST.  Kernel instantiations <true, true, NT, G, false, true>.
bool irls_pass_fused_icpt_eligible(const double* X, int64_t ldx, const double* y, int64_t n, int p) {
    const int pe = p + 1;
    if (pe < FP_MIN_P || pe > FP_MAX_P || (p & 1) || n < FP_MIN_ROWS) return false;
    const char* e = knob("DLSA_IRLS_FUSED");
    if (e && atoi(e) == 0) return false;
    int nt, gt;
    fp_shape(pe, nt, gt);
    if (nt == 6 && gt == 3) return false;         // (p + 1 = 105 .. 108 with w_out: five VGPRs short; those fits keep the two launches)
    int64_t rps;
    fp_slabs(n, rps);
    return ldx % 2 == 0 && ((uintptr_t)X % 16) == 0 && ((uintptr_t)y % 8) == 0 && (double)(rps + 8 * FP_KC) * (double)ldx * 8.0 < 2.0e9;
}

// slab partials (kernel column order: intercept LAST) -> H with the intercept first, both triangles; fixed order over the slabs
__global__ __launch_bounds__(256) void fused_reduce_icpt_kernel(const double* __restrict__ partial, int nslab, int PP, int pe,
                                                                double* __restrict__ H, int64_t ldh) {
    __shared__ double part[16][17];
    const int jl = threadIdx.x & 15, kg = threadIdx.x >> 4;
    const int j = blockIdx.x * 16 + jl, i = blockIdx.y;
    if (blockIdx.x * 16 + 15 < i) return;
    const bool live = j < pe && j >= i;
    double s0 = 0.0;
    if (live)
        for (int k = kg; k < nslab; k += 16) s0 += partial[(int64_t)k * PP * PP + (int64_t)i * PP + j];
    part[kg][jl] = s0;
    __syncthreads();
    if (kg == 0 && live) {
        double s = part[0][jl];
#pragma unroll
        for (int g2 = 1; g2 < 16; ++g2) s += part[g2][jl];
        const int io = i == pe - 1 ? 0 : i + 1, jo = j == pe - 1 ? 0 : j + 1;
        H[(int64_t)io * ldh + jo] = s;
        H[(int64_t)jo * ldh + io] = s;
    }
}

int irls_pass_icpt_impl(const double* X, int64_t ldx, const double* y, const double* beta, int64_t n, int p, double* H, int64_t ldh,
                        double* g, double* loglik, double* w_out, void* ws, size_t ws_bytes, hipStream_t stream) {
    const int pe = p + 1;
    DLSA_REQUIRE(X && y && beta && H && ldh >= pe, "irls_pass (intercept): null argument or ldh < p + 1");      // (w_out may be null: no weights written)
    FusedArgs a;
    a.slabs = nullptr; a.active = nullptr; a.beta_stride = 0;
    a.X = X; a.y = y; a.beta = beta; a.w_out = w_out; a.ldx = ldx; a.n = n; a.p = p; a.PP = (int)fp_pp(pe);
    int nt, gt;
    fp_shape(pe, nt, gt);
    const int ntc = nt + (gt > 0 ? 1 : 0), GP = fp_gp(ntc);
    const int nslab = fp_slabs(n, a.rows_per_slab, 1);
    const size_t part = align_up((size_t)nslab * a.PP * a.PP * 8, 256), gpb = align_up((size_t)nslab * GP * 8, 256);
    const size_t need = part + gpb + kGramProbeBytes;
    if (!ws || ws_bytes < need || ((uintptr_t)ws & 255)) {
        set_error("irls_pass (intercept): workspace %zu bytes needed (256-aligned), got %zu", need, ws_bytes);
        return DLSA_ERR_WORKSPACE;
    }
    a.partial = (double*)ws;
    a.gpart = (double*)((char*)ws + part);
    a.clk = (unsigned long long*)((char*)ws + part + gpb);
#define DLSA_LAUNCH_FPI_W(WO, NTV, GV) do { \
        const size_t shm = fp_hess_lds(NTV, GV); \
        DLSA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(irls_pass_narrow_kernel<WO, true, NTV, GV, false, true>), \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm)); \
        hipLaunchKernelGGL((irls_pass_narrow_kernel<WO, true, NTV, GV, false, true>), dim3(nslab), dim3(256), shm, stream, a); } while (0)
#define DLSA_LAUNCH_FPI(NTV, GV) do { if (w_out) DLSA_LAUNCH_FPI_W(true, NTV, GV); else DLSA_LAUNCH_FPI_W(false, NTV, GV); } while (0)
#define DLSA_LAUNCH_FPI_G(NTV) do { switch (gt) { \
        case 0: DLSA_LAUNCH_FPI(NTV, 0); break; case 1: DLSA_LAUNCH_FPI(NTV, 1); break; \
        case 2: DLSA_LAUNCH_FPI(NTV, 2); break; default: DLSA_LAUNCH_FPI(NTV, 3); break; } } while (0)
    switch (nt) {
        case 3: DLSA_LAUNCH_FPI_G(3); break; case 4: DLSA_LAUNCH_FPI_G(4); break;
        case 5: DLSA_LAUNCH_FPI_G(5); break;
        case 6: switch (gt) { case 0: DLSA_LAUNCH_FPI(6, 0); break; case 1: DLSA_LAUNCH_FPI(6, 1); break; default: DLSA_LAUNCH_FPI(6, 2); break; } break;
        default: switch (gt) { case 0: DLSA_LAUNCH_FPI(7, 0); break; case 1: DLSA_LAUNCH_FPI(7, 1); break; default: DLSA_LAUNCH_FPI(7, 2); break; } break;
    }
#undef DLSA_LAUNCH_FPI_G
#undef DLSA_LAUNCH_FPI
#undef DLSA_LAUNCH_FPI_W
    DLSA_HIP_CHECK(hipGetLastError());
    note_gram_kernel(a.clk, stream, "irls_pass_narrow_kernel<%s,true,%d,%d,icpt>", w_out ? "true" : "false", nt > 6 ? 7 : nt, nt > 6 ? (gt > 2 ? 2 : gt) : gt);
    hipLaunchKernelGGL(fused_reduce_icpt_kernel, dim3((pe + 15) / 16, pe), dim3(256), 0, stream, (const double*)ws, nslab, a.PP, pe, H, ldh);
    hipLaunchKernelGGL(irls_pass_finish_kernel, dim3((pe + 1 + 15) / 16), dim3(256), 0, stream, (const double*)a.gpart, nslab, GP, pe,
                       16 * ntc, g, loglik, 1);
    DLSA_HIP_CHECK(hipGetLastError());
    return DLSA_OK;
}

// ---- the batched form (irls_batch.hip): ONE launch of the fused kernel over a table of slabs, each with its partition's own beta
int irls_pass_batched_pp(int p) { return (int)fp_pp(p); }
int irls_pass_batched_gp(int p) {
    int nt, gt;
    fp_shape(p, nt, gt);
    return fp_gp(nt + (gt > 0 ? 1 : 0));
}
int irls_pass_batched_ll_at(int p) {
    int nt, gt;
    fp_shape(p, nt, gt);
    return 16 * (nt + (gt > 0 ? 1 : 0));
}
// (p = data columns; with the implicit intercept the kernel's shape is that of p + 1 columns, the PP / GP / ll_at queries take p + 1)
bool irls_pass_batched_shape_ok(const double* X, int64_t ldx, const double* y, int p, int intercept, int64_t base_ldx) {
    const int pe = p + (intercept ? 1 : 0);
    // (the implicit intercept's ones column is column p of a stage, which the DMA must never write: even p only)
    if (intercept && (p & 1)) return false;
    return pe >= FP_MIN_P && pe <= FP_MAX_P && fp_rows_ok(X, ldx, p, base_ldx) && ((uintptr_t)y % 8) == 0;
}
// want_h == 0: the logit-only form (g and loglik per slab, no H partial: `partial` unused) -- the lock step's gradient-only iterations
int irls_pass_batched_launch(const double* X, int64_t ldx, const double* y, const double* beta, int64_t beta_stride, int p, int intercept,
                             const FusedSlab* d_slabs, int nslab, const int* d_active, double* partial, double* gpart,
                             unsigned long long* clk, hipStream_t stream, int want_h) {
    const int pe = p + (intercept ? 1 : 0);
    FusedArgs a;
    a.slabs = d_slabs; a.active = d_active; a.beta_stride = beta_stride;
    a.X = X; a.y = y; a.beta = beta; a.w_out = nullptr; a.ldx = ldx; a.n = 0; a.rows_per_slab = 0; a.p = p; a.PP = (int)fp_pp(pe);
    a.partial = partial; a.gpart = gpart; a.clk = clk;
    int nt, gt;
    fp_shape(pe, nt, gt);
#define DLSA_LAUNCH_FPB2(HS, NTV, GV, IC) do { \
        const size_t shm = (HS) ? fp_hess_lds(NTV, GV) : (size_t)fp_logit_stages(NTV + (GV > 0 ? 1 : 0)) * fp_buf(NTV + (GV > 0 ? 1 : 0)) * 8; \
        DLSA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(irls_pass_narrow_kernel<false, HS, NTV, GV, true, IC>), \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm)); \
        hipLaunchKernelGGL((irls_pass_narrow_kernel<false, HS, NTV, GV, true, IC>), dim3(nslab), dim3(256), shm, stream, a); } while (0)
#define DLSA_LAUNCH_FPB(NTV, GV) do { \
        if (want_h) { if (intercept) DLSA_LAUNCH_FPB2(true, NTV, GV, true); else DLSA_LAUNCH_FPB2(true, NTV, GV, false); } \
        else { if (intercept) DLSA_LAUNCH_FPB2(false, NTV, GV, true); else DLSA_LAUNCH_FPB2(false, NTV, GV, false); } } while (0)
#define DLSA_LAUNCH_FPB_G(NTV) do { switch (gt) { \
        case 0: DLSA_LAUNCH_FPB(NTV, 0); break; case 1: DLSA_LAUNCH_FPB(NTV, 1); break; \
        case 2: DLSA_LAUNCH_FPB(NTV, 2); break; default: DLSA_LAUNCH_FPB(NTV, 3); break; } } while (0)
    switch (nt) {
        case 3: DLSA_LAUNCH_FPB_G(3); break; case 4: DLSA_LAUNCH_FPB_G(4); break;
        case 5: DLSA_LAUNCH_FPB_G(5); break; case 6: DLSA_LAUNCH_FPB_G(6); break;
        default: switch (gt) { case 0: DLSA_LAUNCH_FPB(7, 0); break; case 1: DLSA_LAUNCH_FPB(7, 1); break; default: DLSA_LAUNCH_FPB(7, 2); break; } break;
    }
#undef DLSA_LAUNCH_FPB_G
#undef DLSA_LAUNCH_FPB
#undef DLSA_LAUNCH_FPB2
    DLSA_HIP_CHECK(hipGetLastError());
    return DLSA_OK;
}

}  // namespace dlsa

extern "C" {

size_t dlsa_irls_pass_workspace_bytes(int64_t n, int p) {
    if (p <= 0 || p > 2048 || n < 0) return 0;
    // the two-launch form keeps w in the workspace when the caller does not ask for it
    return dlsa::irls_pass_workspace_bytes_impl(n, p) + dlsa::align_up((size_t)n * sizeof(double), 256);
}

int dlsa_irls_pass_f64(const double* X, int64_t ldx, const double* y, const double* beta, int64_t n, int p, double* H,
                       int64_t ldh, double* g, double* loglik, double* w_out, void* ws, size_t ws_bytes, void* stream) {
    using namespace dlsa;
    const size_t pass = irls_pass_workspace_bytes_impl(n, p), wbytes = align_up((size_t)n * sizeof(double), 256);
    if (!ws || ws_bytes < pass + wbytes || ((uintptr_t)ws & 255)) {
        set_error("irls_pass: workspace %zu bytes needed (256-aligned), got %zu", pass + wbytes, ws_bytes);
        return DLSA_ERR_WORKSPACE;
    }
    const size_t pass_al = align_up(pass, 256);
    return irls_pass_impl(X, ldx, y, beta, n, p, H, ldh, g, loglik, w_out, (double*)((char*)ws + pass_al), ws, pass_al,
                          (hipStream_t)stream, nullptr);
}

}  // extern "C"
