// Weighted Gram H = X' diag(w) X for NARROW fp64 designs (49 <= p <= 120 in an even row pitch: config 2's p = 100,
// config 1's p = 50, and the same with an intercept column),
// reference call site dlsa/models.py:130.
//
// At these widths the whole upper triangle of H is at most 28 tiles of 16x16, i.e. <= 224 accumulator registers
// in fp64 -- it fits in ONE wave's AGPRs.  So instead of cutting the tiles over the four waves of a workgroup (gram.hip's
// list plan: one fragment pair from LDS per MFMA, 86 instead of 64 cycles per tile-step, and 7 real tiles in 8
// slots at p = 100), the four waves split the ROWS: every wave owns all tiles and takes every fourth 4-row k-step of a
// staged chunk.  Per k-step a wave reads its fragments once (A and B are the same fragments, B scaled by w) and issues
// its MFMAs with no wasted slot.  The accumulators live in AGPRs; the four partial triangles meet in LDS once per
// workgroup, at the end.
//
// Columns are NOT padded to whole 16-wide tiles: p = 16 NT + (up to 12) columns are covered by NT full tiles
// (NT (NT + 1) / 2 v_mfma_f64_16x16x4_f64 per k-step) plus G <= 3 four-column TAIL GROUPS.  A tail group costs NT + 1
// v_mfma_f64_4x4x4_4b_f64: that instruction computes the four diagonal 4 x 4 blocks of the 16 x 16 outer product of two
// fragments in the big MFMA's own lane layout (A[i = l & 15][k = l >> 4], probed in bench/probe_mfma4x4.hip), at a quarter
// of the big MFMA's time (16 vs 64 cycles) -- so with the tail's four columns broadcast to the four blocks of B, fragment t
// as A gives H[16 t .. 16 t + 15][tail columns] from registers the wave already holds.  p = 100: 21 tiles + 7 small MFMAs
// = 1456 pipe cycles per k-step instead of 28 x 64 = 1792 (2.61 -> 2.24 ms per 1e7 rows); p = 50: 6 + 4 instead of 10 tiles.
// PMC at p = 100 (profiles/r02_pmc_p100.json): SQ_VALU_MFMA_BUSY_CYCLES = 1456 per k-step exactly, the fp64 pipe 87 % busy
// at the 1.9-2.0 GHz the chip sustains under this load (the clock, not the issue rate, is what is left).
//
// Rows stream global -> LDS with the LDS-DMA (buffer_load ... lds, 16 bytes per lane, one instruction per row)
// through a ring of NSTAGE chunk buffers; two chunks are in flight while the third is consumed, with a counted
// s_waitcnt vmcnt(N) + raw s_barrier per chunk (the DMA completes in order).  Rows cut into as many slabs as workgroups
// fit at once.
#include "common.h"
#include <algorithm>
#include <stdlib.h>

namespace dlsa {

// gram.hip
template <typename T>
void gram_reduce_launch(const T* partial, int nslab, int PP, int p, T* H, int64_t ldh, int accumulate, hipStream_t stream);

#ifndef DLSA_NARROW_SPREAD
#define DLSA_NARROW_SPREAD 1          // 1: the DMA two chunks ahead is issued in parts behind the segments of the first k-step's MFMA block
#endif
#ifndef DLSA_NARROW_KC
#define DLSA_NARROW_KC 32
#endif
constexpr int NARROW_KC = DLSA_NARROW_KC;         // rows per chunk: a multiple of 16 (KC/16 k-steps per wave)
#ifndef DLSA_STREAM_AUX
#define DLSA_STREAM_AUX 2           // nt on the LDS-DMA row stream (each row is read by one workgroup, once): ring logit pass -7..-8 % at p = 100-112, narrow Gram +3 % at p = 64, neutral at p = 100; 0 = default policy
#endif
#ifndef DLSA_NARROW_STAGES
#define DLSA_NARROW_STAGES 3          // experiment: 2 = two 32-row stages (half the barriers per row, the DMA one chunk ahead) where two workgroups share a CU
#endif
constexpr int NARROW_STAGES = DLSA_NARROW_STAGES;
#ifndef DLSA_NARROW_PRIVATE
#define DLSA_NARROW_PRIVATE 1         // 1: every wave stages the rows (and weights) of ITS OWN k-steps, so nothing in the chunk loop crosses waves and the loop has no s_barrier (round 4); 0: rows dealt round-robin, one barrier per chunk
#endif
constexpr int NARROW_AHEAD = NARROW_STAGES - 1;      // chunks the DMA runs ahead
constexpr int NARROW_MIN_P = 49, NARROW_MAX_P = 120;      // 3..7 tiles (+ tail groups while the triangle fits the 256 AGPRs: 7 tiles + 2 groups)
constexpr int64_t NARROW_MIN_ROWS = 8192;

struct NarrowArgs {
    const double* X;
    const double* w;
    double* partial;      // [nslab][PP][PP]
    int64_t ldx, n, rows_per_slab;
    int p, PP;
    int dbg;              // DLSA_GRAM_DBG (timing experiments only, wrong results): 1 = no DMA after the prologue, 128 = no MFMAs
    unsigned long long* clk;   // clock probe: wave 0 of workgroup 0 stores its s_memtime delta (dlsa_gram_last_kernel)
};

// LDS row pitch in elements: a k-step's fragment read is 4 rows x 16 columns of 8 bytes; 32 lanes (two rows) are
// served per cycle, so consecutive rows must sit 128 bytes apart modulo 256: pitch = 16 mod 32.
constexpr int narrow_pitch(int nt) { return (nt % 2) ? nt * 16 : nt * 16 + 16; }
constexpr int narrow_buf_elems(int nt, int kc) { return kc * narrow_pitch(nt) + kc; }     // chunk + its w
// The MFMA block is inline assembly on explicitly numbered AGPRs (generated: tools/gen_gram_narrow_asm.py).  With the
// builtin -- or with "+a"-constrained operands -- hipcc carries the 224 accumulator registers of the loop in VGPRs
// and copies all of them into AGPRs and back around every k-step (448 v_accvgpr moves per 28 MFMAs: measured no
// faster than the tile-list kernel).  Named registers stay where the MFMAs want them.
#include "gram_narrow_asm.inc"
#ifndef DLSA_NARROW_DIAG4
#define DLSA_NARROW_DIAG4 0           // 1 (round 6, measured, NOT the default -- same-box A/B: MFMA busy cycles -6.6 % at p = 100, -10.7 % at p = 50 as planned, launch time +-1 %, +2.5 % at p = 112; profiles/r06_narrow_diag4.txt): diagonal tiles on three v_mfma_f64_4x4x4_4b_f64 against column-rotated fragments (48 pipe cycles instead of 64; tools/gen_gram_narrow_d4_asm.py); 0: one 16x16x4 MFMA per diagonal tile
#endif
#include "gram_narrow_d4_asm.inc"

// segments SEG .. NARROW_NSEG - 1 of a k-step's MFMA block, with between(q) issued behind segment q
template <int NT, int G, int SEG, typename F>
__device__ __forceinline__ void narrow_kstep_spread(const double (&f)[NT + (G > 0 ? 1 : 0)], const double (&g)[NT],
                                                    const double (&bt)[G > 0 ? G : 1], F&& between) {
    if constexpr (SEG < NARROW_NSEG) {
        narrow_kstep_seg<NT, G, SEG>(f, g, bt);
        __builtin_amdgcn_sched_barrier(0);
        between(SEG);
        __builtin_amdgcn_sched_barrier(0);
        narrow_kstep_spread<NT, G, SEG + 1>(f, g, bt, between);
    }
}

template <int NT, int G, int SEG, typename F>
__device__ __forceinline__ void narrowd_kstep_spread(const double (&f)[NT + (G > 0 ? 1 : 0)], const double (&g)[NT],
                                                     const double (&bt)[G > 0 ? G : 1], const double (&r1)[NT], const double (&r2)[NT], F&& between) {
    if constexpr (SEG < NARROWD_NSEG) {
        narrowd_kstep_seg<NT, G, SEG>(f, g, bt, r1, r2);
        __builtin_amdgcn_sched_barrier(0);
        between(SEG);
        __builtin_amdgcn_sched_barrier(0);
        narrowd_kstep_spread<NT, G, SEG + 1>(f, g, bt, r1, r2, between);
    }
}

// fn(t, v) for the local tiles t in [T, TEND) with v = this lane's four doubles of tile t (AGPRs a[8t : 8t + 7])
template <int T, int TEND, typename F>
__device__ __forceinline__ void narrow_for_tiles(F&& fn) {
    if constexpr (T < TEND) {
        double v[4];
        narrow_tile_read<T>(v);
        fn(T, v);
        narrow_for_tiles<T + 1, TEND>(fn);
    }
}
// fn(k, v) for the tail accumulators k in [K, NTAIL): v = the lane's double of accumulator k (AGPRs a[8 NTRI + 2k : +1])
template <int NTRI, int NTAIL, int K, typename F>
__device__ __forceinline__ void narrow_for_tails(F&& fn) {
    if constexpr (K < NTAIL) {
        fn(K, narrow_pair_read<8 * NTRI + 2 * K>());
        narrow_for_tails<NTRI, NTAIL, K + 1>(fn);
    }
}

// The four waves' partial tiles [T0, T1) meet in LDS: waves 1..3 park theirs, wave 0 adds them to its own in a fixed
// order and stores the slab's partial.
template <int T0, int T1, int MEETN>
__device__ __forceinline__ void narrow_meet(double* lds, int wave, int lane, double* __restrict__ P, int PP) {
    if constexpr (T0 < T1) {
        if (wave != 0)
            narrow_for_tiles<T0, T1>([&](int t, double (&v)[4]) {
                double* d = lds + ((wave - 1) * MEETN + (t - T0)) * 256 + lane * 4;
                d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3];
            });
        __syncthreads();
        if (wave == 0)
            narrow_for_tiles<T0, T1>([&](int t, double (&v)[4]) {
                int tj = 0;
                while ((tj + 1) * (tj + 2) / 2 <= t) ++tj;
                const int ti = t - tj * (tj + 1) / 2;
                const double* s1 = lds + (t - T0) * 256 + lane * 4;
#pragma unroll
                for (int r = 0; r < 4; ++r) {     // C/D register r of lane l = C[4r + (l >> 4)][l & 15]
                    const double sum = ((v[r] + s1[r]) + s1[MEETN * 256 + r]) + s1[2 * MEETN * 256 + r];
                    P[(int64_t)(ti * 16 + 4 * r + (lane >> 4)) * PP + tj * 16 + (lane & 15)] = sum;
                }
            });
        __syncthreads();
    }
}

// The tail accumulators meet the same way (one double per lane and accumulator).  Accumulator k = gi (NT + 1) + t, lane l:
// H[16 t + 4 b + i][16 NT + 4 gi + j] with i = l >> 4, b = (l & 15) >> 2, j = l & 3.  (For the partial tile t = NT the
// blocks b > gi lie below the diagonal or past p: stored into the slab partial all the same, never read by the reduce.)
template <int NT, int G>
__device__ __forceinline__ void narrow_meet_tails(double* lds, int wave, int lane, double* __restrict__ P, int PP) {
    constexpr int NTRI = NT * (NT + 1) / 2, NTAIL = (NT + 1) * G;
    if constexpr (NTAIL > 0) {
        if (wave != 0)
            narrow_for_tails<NTRI, NTAIL, 0>([&](int k, double v) { lds[((wave - 1) * NTAIL + k) * 64 + lane] = v; });
        __syncthreads();
        if (wave == 0)
            narrow_for_tails<NTRI, NTAIL, 0>([&](int k, double v) {
                const double* s1 = lds + k * 64 + lane;
                const double sum = ((v + s1[0]) + s1[NTAIL * 64]) + s1[2 * NTAIL * 64];
                const int gi = k / (NT + 1), t = k - gi * (NT + 1);
                const int row = 16 * t + 4 * ((lane & 15) >> 2) + (lane >> 4), col = 16 * NT + 4 * gi + (lane & 3);
                P[(int64_t)row * PP + col] = sum;
            });
        __syncthreads();
    }
}

// ---- DIAG4 layout (tools/gen_gram_narrow_d4_asm.py): off-diagonal tiles o = tj (tj - 1) / 2 + ti in a[8 o : 8 o + 7], then PAIRS:
// diagonal tile t under rotation s (pair 3 t + s), then the tail accumulators (pair 3 NT + k).
template <int T, int TEND, typename F>
__device__ __forceinline__ void narrowd_for_tiles(F&& fn) {
    if constexpr (T < TEND) {
        double v[4];
        narrowd_tile_read<T>(v);
        fn(T, v);
        narrowd_for_tiles<T + 1, TEND>(fn);
    }
}
template <int NOFF, int KEND, int K, typename F>
__device__ __forceinline__ void narrowd_for_pairs(F&& fn) {
    if constexpr (K < KEND) {
        fn(K, narrowd_pair_read<8 * NOFF + 2 * K>());
        narrowd_for_pairs<NOFF, KEND, K + 1>(fn);
    }
}
template <int T0, int T1, int MEETN>
__device__ __forceinline__ void narrowd_meet(double* lds, int wave, int lane, double* __restrict__ P, int PP) {
    if constexpr (T0 < T1) {
        if (wave != 0)
            narrowd_for_tiles<T0, T1>([&](int t, double (&v)[4]) {
                double* d = lds + ((wave - 1) * MEETN + (t - T0)) * 256 + lane * 4;
                d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3];
            });
        __syncthreads();
        if (wave == 0)
            narrowd_for_tiles<T0, T1>([&](int t, double (&v)[4]) {
                int tj = 1;
                while ((tj + 1) * tj / 2 <= t) ++tj;
                const int ti = t - tj * (tj - 1) / 2;
                const double* s1 = lds + (t - T0) * 256 + lane * 4;
#pragma unroll
                for (int r = 0; r < 4; ++r) {     // C/D register r of lane l = C[4r + (l >> 4)][l & 15]
                    const double sum = ((v[r] + s1[r]) + s1[MEETN * 256 + r]) + s1[2 * MEETN * 256 + r];
                    P[(int64_t)(ti * 16 + 4 * r + (lane >> 4)) * PP + tj * 16 + (lane & 15)] = sum;
                }
            });
        __syncthreads();
    }
}
// Pair d of lane l (i = l >> 4, b = (l & 15) >> 2, j = l & 3).  d = 3 t + s < 3 NT: H[16 t + 4 b + i][16 t + 4 ((b + s) & 3) + j] -- block
// (b, (b + s) & 3) of diagonal tile t; a block below the diagonal (b + s >= 4) is the mirror image of one the reduce needs above it
// (s = 1: (3, 0) -> (0, 3)) or of one another lane group already holds (s = 2: (2, 0), (3, 1)), so it is stored transposed / dropped.
// d = 3 NT + k, k = gi (NT + 1) + t: H[16 t + 4 b + i][16 NT + 4 gi + j] as narrow_meet_tails.
// (pairs [K0, K1): as many per meeting pass as three parked copies fit the ring)
template <int NT, int G, int K0, int K1>
__device__ __forceinline__ void narrowd_meet_pairs(double* lds, int wave, int lane, double* __restrict__ P, int PP) {
    constexpr int NOFF = NT * (NT - 1) / 2, NP = K1 - K0;
    if constexpr (NP <= 0) return;
    if (wave != 0)
        narrowd_for_pairs<NOFF, K1, K0>([&](int k, double v) { lds[((wave - 1) * NP + (k - K0)) * 64 + lane] = v; });
    __syncthreads();
    if (wave == 0)
        narrowd_for_pairs<NOFF, K1, K0>([&](int k, double v) {
            const double* s1 = lds + (k - K0) * 64 + lane;
            const double sum = ((v + s1[0]) + s1[NP * 64]) + s1[2 * NP * 64];
            const int b = (lane & 15) >> 2, i = lane >> 4, j = lane & 3;
            if (k < 3 * NT) {
                const int t = k / 3, s = k - 3 * t, bc = (b + s) & 3;
                const int row = 16 * t + 4 * b + i, col = 16 * t + 4 * bc + j;
                if (b + s < 4) P[(int64_t)row * PP + col] = sum;
                else if (s == 1) P[(int64_t)col * PP + row] = sum;
            } else {
                const int kt = k - 3 * NT, gi = kt / (NT + 1), t = kt - gi * (NT + 1);
                P[(int64_t)(16 * t + 4 * b + i) * PP + 16 * NT + 4 * gi + j] = sum;
            }
        });
    __syncthreads();
}

// Accumulator registers of a shape, and workgroups per CU: two fit when accumulators + ~72 fragment VGPRs stay within the 256
// registers two waves per SIMD can have each (up to 6 tiles + one tail group: 182 + 72 = 254) and the two rings fit the
// LDS (16-row chunks where 32-row ones would not: narrow_kc); one workgroup's LDS waits, barriers and DMA issue then hide
// under the other's MFMAs (per 1e7 rows: p = 50 1.22 -> 1.06 ms, p = 100 2.32 -> 2.24 ms, p = 96 2.17 -> 2.08 ms).  Wider
// shapes take a CU alone.  (Splitting the 28 tiles of 112 columns over wave PAIRS -- eight waves, 112 AGPRs each, two per
// SIMD -- was built and measured in round 1: 2.53 vs 2.55 ms, no gain.)
constexpr int narrow_nreg_full(int nt, int g) { return 8 * (nt * (nt + 1) / 2) + 2 * (nt + 1) * g; }
// (DIAG4: 6 instead of 8 accumulator registers per diagonal tile, and two more fragments -- the rotated ones -- per tile column)
constexpr int narrow_nreg(int nt, int g) { return DLSA_NARROW_DIAG4 ? 8 * (nt * (nt - 1) / 2) + 2 * (3 * nt + (nt + 1) * g) : narrow_nreg_full(nt, g); }
#ifndef DLSA_NARROW_WGS2_MAXREG
#define DLSA_NARROW_WGS2_MAXREG 184
#endif
constexpr int narrow_wgs_per_cu(int nt, int g) { return narrow_nreg_full(nt, g) <= DLSA_NARROW_WGS2_MAXREG ? 2 : 1; }      // (the same shapes either way)
// Rows per chunk: NARROW_KC (32: two k-steps per wave and barrier) unless two workgroups are to share a CU and their
// rings would not fit the LDS side by side; then 16.
constexpr int narrow_kc(int nt, int g) {
    const int ntc = nt + (g > 0 ? 1 : 0);
    return (narrow_wgs_per_cu(nt, g) == 2 && (size_t)2 * NARROW_STAGES * narrow_buf_elems(ntc, NARROW_KC) * 8 > (size_t)kLdsBytes) ? 16 : NARROW_KC;
}
constexpr int narrow_meetn(int nt, int ntc, int kc) {      // tiles per meeting pass: three parked copies must fit in the ring
    const int ntri = DLSA_NARROW_DIAG4 ? nt * (nt - 1) / 2 : nt * (nt + 1) / 2, fit = (int)((size_t)NARROW_STAGES * narrow_buf_elems(ntc, kc) * 8 / (3 * 2048));
    return fit < ntri ? fit : ntri;
}
constexpr size_t narrow_lds_bytes(int ntc, int kc) { return (size_t)NARROW_STAGES * narrow_buf_elems(ntc, kc) * 8; }

// NT full 16-column tiles + G four-column tail groups: C = 16 NT + 4 G >= p columns (see gram_narrow_shape).
template <bool HASW, int NT, int G>
__global__ __launch_bounds__(256, narrow_wgs_per_cu(NT, G)) void gram_narrow_kernel(NarrowArgs a) {
    constexpr int KC = narrow_kc(NT, G);
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    constexpr int NWAVES = 4, THREADS = 64 * NWAVES;
    constexpr int NTC = NT + (G > 0 ? 1 : 0);            // tile columns staged and read as fragments (the last one partial)
    constexpr int LDP = narrow_pitch(NTC), BUF = narrow_buf_elems(NTC, KC), NTRI = NT * (NT + 1) / 2;
    constexpr int NTAIL = (NT + 1) * G, GA = G > 0 ? G : 1;
    constexpr int NOFF = NT * (NT - 1) / 2, NPAIR = 3 * NT + NTAIL;      // DIAG4: off-diagonal tiles, pair accumulators
    constexpr int MEETN = narrow_meetn(NT, NTC, KC);
    constexpr int DMA_PER_CHUNK = KC / NWAVES + (HASW ? 1 : 0);        // instructions per wave and chunk
    static_assert(KC % 16 == 0 && 4 * MEETN >= (DLSA_NARROW_DIAG4 ? NOFF : NTRI), "chunk / meeting shape");
    constexpr int MEETP = (NARROW_STAGES * BUF) / (3 * 64);              // DIAG4: pairs per meeting pass
    static_assert(DLSA_NARROW_DIAG4 ? 2 * MEETP >= NPAIR : (size_t)3 * NTAIL * 64 <= (size_t)NARROW_STAGES * BUF, "tail meeting fits the ring");
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool probe = blockIdx.x == 0 && wave == 0;     // wave-uniform
    const unsigned long long t_begin = probe ? __builtin_readcyclecounter() : 0ull;
    const int slab = blockIdx.x;
    const int64_t rbeg = (int64_t)slab * a.rows_per_slab;
    const int64_t rend = min(rbeg + a.rows_per_slab, a.n);
    const int64_t nrows = rend > rbeg ? rend - rbeg : 0;
    const int nchunks = (int)((nrows + KC - 1) / KC);

    // rows past the slab end read as zeros through the buffer descriptors; columns p .. 16 NTC - 1 are never written
    // by the DMA (lanes masked), so the ring is zeroed once
    const unsigned xbytes = nrows > 0 ? (unsigned)(((nrows - 1) * a.ldx + a.p) * 8) : 0u;
    __amdgpu_buffer_rsrc_t rsrcX = __builtin_amdgcn_make_buffer_rsrc((void*)(a.X + rbeg * a.ldx), 0, (int)xbytes, 0x00020000);
    __amdgpu_buffer_rsrc_t rsrcW =
        __builtin_amdgcn_make_buffer_rsrc((void*)(HASW ? a.w + rbeg : a.X), 0, HASW ? (int)(nrows * 8) : 0, 0x00020000);
    for (int e = tid; e < NARROW_STAGES * BUF; e += THREADS) lds[e] = 0.0;
    __syncthreads();

    const bool col_in = 2 * lane < a.p;                 // a.p is even: both columns of the lane's 16 bytes are loaded
    // Row ps (of KC / 4) this wave stages.  PRIVATE: the rows of its own k-steps (k-step wave + 4 kk = rows 4 (wave + 4 kk) .. + 3), so
    // a wave's MFMAs read only what its own DMA wrote: its own counted s_waitcnt vmcnt orders everything, no barrier in the loop,
    // and the four waves drift freely (their non-MFMA stretches stop coinciding).
    auto own_row = [&](int ps) { return DLSA_NARROW_PRIVATE ? 4 * (wave + 4 * (ps >> 2)) + (ps & 3) : wave + NWAVES * ps; };
    // the chunk's weights.  PRIVATE: this wave's KC / 4 values only, into its own corner of the w slot ([wave][kk][4]): lane L
    // fetches the two weights 2 (L & 1) .. + 1 of k-step L >> 1.  Else every wave fetches all of them (same bytes to the same place).
    auto stage_w = [&](int chunk, int buf) {
        if constexpr (HASW) {
            double* wdst = lds + buf * BUF + KC * LDP;
#if DLSA_NARROW_PRIVATE
            if (lane < KC / 8)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcW, (lds_ptr_t)(wdst + (KC / 4) * wave), 16,
                                                         (4 * (wave + 4 * (lane >> 1)) + 2 * (lane & 1)) * 8, chunk * KC * 8, 0, 0);
#else
            if (lane < KC / 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcW, (lds_ptr_t)wdst, 16, lane * 16, chunk * KC * 8, 0, 0);
#endif
        }
    };
    auto stage = [&](int chunk, int buf) {
        double* base = lds + buf * BUF;
#pragma unroll
        for (int ps = 0; ps < KC / NWAVES; ++ps) {
            const int row = own_row(ps);
            const int soff = (int)(((int64_t)chunk * KC + row) * a.ldx * 8);
            if (col_in) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcX, (lds_ptr_t)(base + row * LDP), 16, lane * 16, soff, 0, DLSA_STREAM_AUX);
        }
        stage_w(chunk, buf);
    };

    // the same DMA in five parts: rows wave + 4 ps for ps in [q RQ, (q + 1) RQ), q < 4; then w
    constexpr int RQ = KC / NWAVES / 4;
    static_assert(RQ * 4 * NWAVES == KC, "rows per wave and chunk in four groups");
    auto stage_rows = [&](int chunk, int buf, int q) {
        double* base = lds + buf * BUF;
#pragma unroll
        for (int ps = q * RQ; ps < (q + 1) * RQ; ++ps) {
            const int row = own_row(ps);
            const int soff = (int)(((int64_t)chunk * KC + row) * a.ldx * 8);
            if (col_in) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcX, (lds_ptr_t)(base + row * LDP), 16, lane * 16, soff, 0, DLSA_STREAM_AUX);
        }
    };

#if DLSA_NARROW_DIAG4
    narrowd_acc_zero<narrow_nreg(NT, G)>();
#else
    narrow_acc_zero<narrow_nreg(NT, G)>();
#endif

#pragma unroll
    for (int ch = 0; ch < NARROW_AHEAD; ++ch) stage(ch, ch);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NARROW_AHEAD - 1) * DMA_PER_CHUNK) : "memory");
#if !DLSA_NARROW_PRIVATE
    asm volatile("s_barrier" ::: "memory");
#endif

    const int frag_off = (lane >> 4) * LDP + (lane & 15);
    const int tail_off = (lane >> 4) * LDP + 16 * NT + (lane & 3);      // the 4 tail columns, broadcast to the 4 blocks
#if DLSA_NARROW_DIAG4
    // the fragment's columns rotated by 4 and by 8 inside each group of 16 lanes: the B operands of the diagonal tiles' second and third
    // small MFMA (the same LDS rows, the same 128-byte segments: conflict-free like frag_off)
    const int rot1_off = (lane >> 4) * LDP + ((lane + 4) & 15), rot2_off = (lane >> 4) * LDP + ((lane + 8) & 15);
#endif
    int cur = 0, nxt2 = NARROW_AHEAD % NARROW_STAGES;    // ring positions of chunk c and of the chunk the DMA fetches (c + NARROW_AHEAD)
    for (int c = 0; c < nchunks; ++c) {
#if !DLSA_NARROW_SPREAD
        if (!DLSA_DBG_WRONG(a.dbg, 1)) stage(c + NARROW_AHEAD, nxt2);            // past the slab end: bounds-checked zeros, no traffic
#endif
        const double* base = lds + cur * BUF;
        // all of this wave's fragments of the chunk are requested up front: only the first k-step waits for LDS
        double f[KC / 16][NTC], wv[KC / 16], bt[KC / 16][GA];
#if DLSA_NARROW_DIAG4
        double r1[KC / 16][NT], r2[KC / 16][NT];
#endif
#pragma unroll
        for (int kk = 0; kk < KC / 16; ++kk) {
            const int ks = wave + 4 * kk;
            const double* kb = base + ks * 4 * LDP;
#pragma unroll
            for (int t = 0; t < NTC; ++t) f[kk][t] = kb[frag_off + t * 16];
#if DLSA_NARROW_DIAG4
#pragma unroll
            for (int t = 0; t < NT; ++t) { r1[kk][t] = kb[rot1_off + t * 16]; r2[kk][t] = kb[rot2_off + t * 16]; }
#endif
#pragma unroll
            for (int gi = 0; gi < G; ++gi) bt[kk][gi] = kb[tail_off + 4 * gi];
            if (HASW) wv[kk] = base[KC * LDP + (DLSA_NARROW_PRIVATE ? (KC / 4) * wave + 4 * kk : ks * 4) + (lane >> 4)];
        }
#pragma unroll
        for (int kk = 0; kk < KC / 16; ++kk) {
            double g[NT], btw[GA];
#pragma unroll
            for (int t = 0; t < NT; ++t) g[t] = HASW ? f[kk][t] * wv[kk] : f[kk][t];
#pragma unroll
            for (int gi = 0; gi < GA; ++gi) btw[gi] = (G > 0) ? (HASW ? bt[kk][gi] * wv[kk] : bt[kk][gi]) : 0.0;
            // tile (ti, tj) += f[ti] (x) g[tj] for all ti <= tj < NT;  tail (t, gi) += blockdiag(f[t] (x) btw[gi]) for t <= NT
#if DLSA_NARROW_SPREAD
            // the DMA of chunk c + 2 (past the slab end: bounds-checked zeros, no traffic) goes out in five parts BEHIND the five
            // segments of the chunk's first k-step (four row groups, then w): as a burst in front of the block it kept the wave
            // out of MFMAs for its whole issue time (gram_plan_kernel.inc)
            if (kk == 0 && !DLSA_DBG_WRONG(a.dbg, 128)) {
                auto between = [&](int q) {
                    if (DLSA_DBG_WRONG(a.dbg, 1)) return;
                    if (q < 4) stage_rows(c + NARROW_AHEAD, nxt2, q); else stage_w(c + NARROW_AHEAD, nxt2);
                };
#if DLSA_NARROW_DIAG4
                narrowd_kstep_spread<NT, G, 0>(f[kk], g, btw, r1[kk], r2[kk], between);
#else
                narrow_kstep_spread<NT, G, 0>(f[kk], g, btw, between);
#endif
                continue;
            }
#endif
#if DLSA_NARROW_DIAG4
            if (!DLSA_DBG_WRONG(a.dbg, 128)) narrowd_kstep<NT, G>(f[kk], g, btw, r1[kk], r2[kk]);
#else
            if (!DLSA_DBG_WRONG(a.dbg, 128)) narrow_kstep<NT, G>(f[kk], g, btw);
#endif
            else asm volatile("" ::"v"(g[0]), "v"(g[NT - 1]), "v"(btw[0]));
        }
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NARROW_AHEAD - 1) * DMA_PER_CHUNK) : "memory");      // chunk c + 1 has landed
#if !DLSA_NARROW_PRIVATE
        asm volatile("s_barrier" ::: "memory");
#endif
        cur = (cur == NARROW_STAGES - 1) ? 0 : cur + 1;
        nxt2 = (nxt2 == NARROW_STAGES - 1) ? 0 : nxt2 + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the zero-fill DMA of the chunks past the end
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // the last MFMAs retire before the accumulators are read (the
    __syncthreads();                                     // compiler cannot see the hazard of an inline-asm MFMA)

    // the four waves' triangles meet in LDS, MEETN tiles at a time; wave 0 stores the slab's partial
    double* __restrict__ P = a.partial + (int64_t)slab * a.PP * a.PP;
#if DLSA_NARROW_DIAG4
    narrowd_meet<0, (MEETN < NOFF ? MEETN : NOFF), MEETN>(lds, wave, lane, P, a.PP);
    narrowd_meet<MEETN, (2 * MEETN < NOFF ? 2 * MEETN : NOFF), MEETN>(lds, wave, lane, P, a.PP);
    narrowd_meet<2 * MEETN, (3 * MEETN < NOFF ? 3 * MEETN : NOFF), MEETN>(lds, wave, lane, P, a.PP);
    narrowd_meet<3 * MEETN, NOFF, MEETN>(lds, wave, lane, P, a.PP);
    narrowd_meet_pairs<NT, G, 0, (MEETP < NPAIR ? MEETP : NPAIR)>(lds, wave, lane, P, a.PP);
    narrowd_meet_pairs<NT, G, (MEETP < NPAIR ? MEETP : NPAIR), NPAIR>(lds, wave, lane, P, a.PP);
#else
    narrow_meet<0, (MEETN < NTRI ? MEETN : NTRI), MEETN>(lds, wave, lane, P, a.PP);
    narrow_meet<MEETN, (2 * MEETN < NTRI ? 2 * MEETN : NTRI), MEETN>(lds, wave, lane, P, a.PP);
    narrow_meet<2 * MEETN, (3 * MEETN < NTRI ? 3 * MEETN : NTRI), MEETN>(lds, wave, lane, P, a.PP);
    narrow_meet<3 * MEETN, NTRI, MEETN>(lds, wave, lane, P, a.PP);
    narrow_meet_tails<NT, G>(lds, wave, lane, P, a.PP);
#endif
    if (probe && lane == 0) *a.clk = __builtin_readcyclecounter() - t_begin;
}

// p columns = NT full tiles + G tail groups of 4 (G <= 3; a fourth group makes a full tile)
static void gram_narrow_shape(int p, int& nt, int& g) {
    nt = p / 16;
    g = (p - 16 * nt + 3) / 4;
    if (g == 4) { ++nt; g = 0; }
}

static int narrow_slabs(int64_t n, int p, int64_t& rows_per_slab) {
    int nt, g;
    gram_narrow_shape(p + (p & 1), nt, g);
    int64_t ns = std::min<int64_t>((int64_t)kNumCU * narrow_wgs_per_cu(nt, g), std::max<int64_t>(1, n / 1024));
    rows_per_slab = ((n + ns - 1) / ns + NARROW_KC - 1) / NARROW_KC * NARROW_KC;
    return (int)((n + rows_per_slab - 1) / rows_per_slab);
}

bool gram_narrow_shape_ok(int64_t n, int p) { return p >= NARROW_MIN_P && p <= NARROW_MAX_P && n >= NARROW_MIN_ROWS; }

bool gram_narrow_eligible(const double* X, int64_t ldx, const double* w, int64_t n, int p) {
    if (!gram_narrow_shape_ok(n, p)) return false;
    if (gram_dbg_env() & 64) return false;      // 64: keep the list plan (A/B runs)
    int64_t rps;
    narrow_slabs(n, p, rps);
    return ldx % 2 == 0 && ((uintptr_t)X % 16) == 0 && (!w || ((uintptr_t)w % 16) == 0) &&
           (double)(rps + 4 * NARROW_KC) * (double)ldx * 8.0 < 2.0e9;                 // 32-bit DMA offsets
}

size_t gram_narrow_ws_bytes(int64_t n, int p) {
    // the first estimate of narrow_slabs: an upper bound of its result that is monotone in n (a workspace sized for the largest
    // partition serves every smaller row count; the exact count is not monotone, see slab_bound in gram.hip)
    int nt, g;
    gram_narrow_shape(p + (p & 1), nt, g);
    const int64_t ns = std::min<int64_t>((int64_t)kNumCU * narrow_wgs_per_cu(nt, g), std::max<int64_t>(1, n / 1024));
    const size_t PP = ((size_t)(p + 15) / 16 * 16 + 63) / 64 * 64;
    return align_up((size_t)ns * PP * PP * 8, 256) + kGramProbeBytes;
}

int gram_narrow_f64(const double* X, int64_t ldx, const double* w, int64_t n, int p, double* H, int64_t ldh,
                    int accumulate, void* ws, size_t ws_bytes, hipStream_t stream) {
    NarrowArgs a;
    // odd p (in an even row pitch, checked by gram_narrow_eligible): load p + 1 columns; the pad column only reaches row
    // and column p of the tile grid, which nobody reads (see gram_impl)
    a.X = X; a.w = w; a.partial = (double*)ws; a.ldx = ldx; a.n = n; a.p = p + (p & 1);
    a.dbg = gram_dbg_env();
    int nt, g;
    gram_narrow_shape(a.p, nt, g);
    a.PP = ((p + 15) / 16 * 16 + 63) / 64 * 64;
    const int nslab = narrow_slabs(n, p, a.rows_per_slab);
    const size_t need = align_up((size_t)nslab * a.PP * a.PP * 8, 256) + kGramProbeBytes;
    a.clk = (unsigned long long*)((char*)ws + need - kGramProbeBytes);
    if (!ws || ws_bytes < need || ((uintptr_t)ws & 255)) {
        set_error("gram: workspace %zu bytes needed (256-aligned), got %zu", need, ws_bytes);
        return DLSA_ERR_WORKSPACE;
    }
#define DLSA_LAUNCH_NARROW(HW, NTV, GV) do { \
        const size_t shm = narrow_lds_bytes(NTV + (GV > 0 ? 1 : 0), narrow_kc(NTV, GV)); \
        if (shm > 48 * 1024) DLSA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(gram_narrow_kernel<HW, NTV, GV>), \
                                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm)); \
        hipLaunchKernelGGL((gram_narrow_kernel<HW, NTV, GV>), dim3(nslab), dim3(256), shm, stream, a); } while (0)
#define DLSA_LAUNCH_NARROW_G(HW, NTV) do { switch (g) { \
        case 0: DLSA_LAUNCH_NARROW(HW, NTV, 0); break; case 1: DLSA_LAUNCH_NARROW(HW, NTV, 1); break; \
        case 2: DLSA_LAUNCH_NARROW(HW, NTV, 2); break; default: DLSA_LAUNCH_NARROW(HW, NTV, 3); break; } } while (0)
#define DLSA_LAUNCH_NARROW_NT(HW) do { switch (nt) { \
        case 3: DLSA_LAUNCH_NARROW_G(HW, 3); break; case 4: DLSA_LAUNCH_NARROW_G(HW, 4); break; \
        case 5: DLSA_LAUNCH_NARROW_G(HW, 5); break; case 6: DLSA_LAUNCH_NARROW_G(HW, 6); break; \
        default: switch (g) { case 0: DLSA_LAUNCH_NARROW(HW, 7, 0); break; case 1: DLSA_LAUNCH_NARROW(HW, 7, 1); break; \
                              default: DLSA_LAUNCH_NARROW(HW, 7, 2); break; } break; } } while (0)
    if (w) DLSA_LAUNCH_NARROW_NT(true);
    else DLSA_LAUNCH_NARROW_NT(false);
    note_gram_kernel(a.clk, stream, "gram_narrow_kernel<%s,%d,%d>", w ? "true" : "false", nt > 6 ? 7 : (nt < 3 ? 3 : nt), nt > 6 ? (g > 2 ? 2 : g) : g);
#undef DLSA_LAUNCH_NARROW_G
#undef DLSA_LAUNCH_NARROW_NT
#undef DLSA_LAUNCH_NARROW
    DLSA_HIP_CHECK(hipGetLastError());
    gram_reduce_launch<double>((const double*)ws, nslab, a.PP, p, H, ldh, accumulate, stream);
    DLSA_HIP_CHECK(hipGetLastError());
    return DLSA_OK;
}

}  // namespace dlsa
