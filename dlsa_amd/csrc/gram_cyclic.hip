// Weighted Gram H = X' diag(w) X for the p = 500 class (481 <= p <= 508, fp64, aligned rows): BASELINE.json's metric width.
// Reference call site: dlsa/models.py:130.
//
// gram.hip's plan for p = 500 paces 576 tile slots for the 528 tiles of the padded 32 x 32 tile triangle (the triangular
// waves of the diagonal blocks idle while their workgroup's full waves finish) and pads 500 columns to 512.  Four layouts
// that rebalance it were measured in round 1 and lost more elsewhere.  This kernel removes both
// losses by construction:
//   * p = 31 full 16-column tiles (496 columns) + G <= 3 four-column tail groups.  The 31 x 31 tile triangle is covered by
//     the CYCLIC classes: tile row i owns the 16 tiles (i, (i + d) mod 31), d = 0..15 -- every unordered pair {i, j} exactly
//     once because 31 is odd (a wrapped tile is the transpose of the one the triangle wants; the epilogue stores it so).
//     So all tile rows carry the same load: 31 x 16 = 496 tiles, no diagonal special case, no triangular wave.
//   * a GROUP of 4 workgroups (one per CU, same XCD) shares a slab of rows.  Its 32 waves are the 16 row blocks (2 tile rows)
//     x 2 distance halves: 16 tiles = 128 AGPRs per wave, two waves per SIMD (the two halves of one row block), every wave
//     the same instruction stream.  Per 4-row k-step a wave reads 2 A + 9 B fragments for 16 v_mfma_f64_16x16x4_f64; the
//     weight goes on the A side (2 multiplications).  The d < 8 wave of a row block also issues the 2 G
//     v_mfma_f64_4x4x4_4b_f64 of its rows' tail columns (gram_narrow.hip explains the instruction).
//   * every workgroup streams the slab's full rows through LDS: 8-row chunks, four stages, LDS-DMA issued three chunks
//     ahead, two pieces behind each tile row of a chunk's second k-step, no per-lane masks; fragment addresses are
//     per-lane registers + compile-time immediates (the chunk loop is unrolled over the stages).  The four workgroups of
//     a group do equal work, stay within microseconds of each other, and three of the four fetches of a row are L2 hits
//     (TCC hit rate 73 %, fabric traffic = 1.0x the algorithmic bytes at p = 496).
// 64 groups = 256 CUs, one launch wave, no tail; the 64 slab partials are summed by gram.hip's reduce kernel.
// Same-box timings per 2.5e7 rows (gram.hip's panel kernel / this one): p = 496 99.9 / 90.8 ms (68.0 TF), p = 500 100.0 /
// 94.9 ms (66.1 TF = 84 % of the 78.6 TF fp64 MFMA peak), p = 504 100.0 / 96.7 ms.
#include "common.h"
#ifndef DLSA_CYC_SPLIT
#define DLSA_CYC_SPLIT 1
#endif
#include <algorithm>
#ifndef DLSA_CYC_FLAGS
#define DLSA_CYC_FLAGS 0                  // experiment: per-stage LANDED / DONE flags in LDS instead of s_barrier (a SIMD that is ahead keeps issuing)
#endif
#ifndef DLSA_CYC_PAIR
#define DLSA_CYC_PAIR 0                   // 1: one barrier per PAIR of chunks (16 rows), the next pair's DMA issued during this pair's first two k-steps
#endif
#ifndef DLSA_CYC_PRIO
#define DLSA_CYC_PRIO 0                   // experiment: 1 = the second-dispatched waves (4..7) run at s_setprio 1, 2 = the first four
#endif

namespace dlsa {

template <typename T>
void gram_reduce_launch(const T* partial, int nslab, int PP, int p, T* H, int64_t ldh, int accumulate, hipStream_t stream);   // gram.hip

constexpr int CYC_NT = 31;                    // full tile columns
constexpr int CYC_C0 = 16 * CYC_NT;           // first tail column (496)
constexpr int CYC_KC = 8;                     // rows per chunk = two 4-row k-steps
constexpr int CYC_NST = 4;                    // LDS stages: the DMA runs three chunks ahead of the MFMAs
constexpr int CYC_LDP = 528;                  // LDS row pitch in doubles: >= 512, = 16 mod 32 (two rows per ds_read_b64 lane group)
constexpr int CYC_BUF = CYC_KC * CYC_LDP + CYC_KC;      // a chunk + its w
constexpr int CYC_GROUP = 4;                  // workgroups per slab
constexpr int CYC_PP = 512;
constexpr int CYC_MIN_P = 481, CYC_MAX_P = 508;
constexpr int64_t CYC_MIN_ROWS = 65536;

struct CycArgs {
    const double* X;
    const double* w;
    double* partial;      // [nslab][PP][PP]
    int64_t ldx, n, rows_per_slab;
    int p;                // columns loaded (even)
    int nslab;
    unsigned long long* clk;   // clock probe: wave 0 of workgroup 0 stores its s_memtime delta (dlsa_gram_last_kernel)
};

#include "gram_cyclic_asm.inc"

template <int T, int TEND, typename F>
__device__ __forceinline__ void cyc_for_tiles(F&& fn) {
    if constexpr (T < TEND) {
        double v[4];
        cyc_tile_read<T>(v);
        fn(T, v);
        cyc_for_tiles<T + 1, TEND>(fn);
    }
}

template <int K, int KEND, typename F>
__device__ __forceinline__ void cyc_for_tails(F&& fn) {
    if constexpr (K < KEND) {
        fn(K, cyc_tail_read<K>());
        cyc_for_tails<K + 1, KEND>(fn);
    }
}

// A wave owns RW = 2 tile rows x 8 distances = 16 tiles (128 AGPRs): 8 waves per workgroup, two per SIMD, so that one wave's
// LDS waits and DMA issue hide under the other's MFMAs.  (RW = 4 -- 32 tiles, one wave per SIMD, 15 instead of 22 fragment reads
// per SIMD and k-step -- was built and measured: 101 vs 97 ms at p = 496; with nobody to cover a wave's waits the pipe idles.)
template <bool HASW, int G>
__global__ __launch_bounds__(512, 2) void gram_cyclic_kernel(CycArgs a) {
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    constexpr int KC = CYC_KC, LDP = CYC_LDP, BUF = CYC_BUF, GA = G > 0 ? G : 1;
    constexpr int RW = 2, NW = 8, NB = RW + 7;
    constexpr int PIECES = KC * 4 / NW;                  // DMA pieces per wave and chunk (4): two behind each tile row of (c, 1)
    constexpr int PPR = PIECES / RW;
    constexpr int DMA_PER_CHUNK = PIECES + (HASW ? 1 : 0);
    static_assert(KC == 8 && PPR * RW == PIECES && CYC_NST == 4, "pipeline shape");
    static_assert((BUF + 4 * LDP) * 8 + 128 < 65536, "stage parity + k-step offsets must fit a ds_read immediate");
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool probe = blockIdx.x == 0 && wave == 0;     // wave-uniform
    const unsigned long long t_begin = probe ? __builtin_readcyclecounter() : 0ull;
    // block -> (slab, member): the four workgroups of a slab sit on one XCD (blocks b, b + 8, b + 16, b + 24 of a 32-block round)
    const int b = blockIdx.x, xcd = b % kNumXCD, jb = b / kNumXCD;
    const int slab = (jb / CYC_GROUP) * kNumXCD + xcd, member = jb % CYC_GROUP;
    // waves w and w + 4 share a SIMD: they take the two distance halves of the same row block
    const int rb = member * 4 + (wave & 3), db = wave >> 2;      // row block 0..15, distance half
    const int i0 = RW * rb, d0 = 8 * db;
    const bool two_rows = rb != 15;                      // 31 tile rows: the last block holds tile row 30 only
    const bool tails = G > 0 && db == 0;                 // the d < 8 wave of a row block also takes its rows' tail columns

    const int64_t rbeg = (int64_t)slab * a.rows_per_slab;
    const int64_t rend = min(rbeg + a.rows_per_slab, a.n);
    const int64_t nrows = rend > rbeg ? rend - rbeg : 0;
    const int nchunks = (int)((nrows + KC - 1) / KC);

    // The DMA copies 512 columns of every row, whatever p: columns p .. 511 of the LDS rows then hold whatever follows the
    // row in memory (the next row, or zeros past the end of the slab through the descriptor's bounds check).  That is
    // harmless: an MFMA output element depends on ONE column of A and ONE column of B, so those columns only reach rows /
    // columns >= p of the tile grid, which nobody reads -- and it keeps every piece free of per-lane masks.
    const unsigned xbytes = nrows > 0 ? (unsigned)(((nrows - 1) * a.ldx + a.p) * 8) : 0u;
    __amdgpu_buffer_rsrc_t rsrcX = __builtin_amdgcn_make_buffer_rsrc((void*)(a.X + rbeg * a.ldx), 0, (int)xbytes, 0x00020000);
    __amdgpu_buffer_rsrc_t rsrcW =
        __builtin_amdgcn_make_buffer_rsrc((void*)(HASW ? a.w + rbeg : a.X), 0, HASW ? (int)(nrows * 8) : 0, 0x00020000);

    // piece pc of this wave: global piece id = pc NW + wave;  row = id >> 2, column quarter = id & 3 (128 columns, 16 B per lane)
    auto dma_piece = [&](int chunk, int buf, int pc) {
        const int id = pc * NW + wave, row = id >> 2, q = id & 3;
        const int soff = (int)((((int64_t)chunk * KC + row) * a.ldx + 128 * q) * 8);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcX, (lds_ptr_t)(lds + buf * BUF + row * LDP + 128 * q), 16, lane * 16, soff, 0, 0);
    };
    auto dma_w = [&](int chunk, int buf) {       // every wave fetches the chunk's w: same in-order count in all waves
        if (HASW && lane < KC / 2)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcW, (lds_ptr_t)(lds + buf * BUF + KC * LDP), 16, lane * 16, chunk * KC * 8, 0, 0);
    };

    cyc_acc_zero<128>();
    if constexpr (G > 0) cyc_tail_zero<2 * RW * G>();

    // Pipeline (chunk c = k-steps (c, 0), (c, 1); stage = c mod 4):
    //   (c, 0): MFMAs;  then s_waitcnt for chunk c + 1 (issued 1.5 chunks ago) + s_barrier: chunk c + 1 is visible to every wave,
    //           and every wave has left chunk c - 1, whose stage the next DMA will overwrite;
    //   (c, 1): fragments of (c + 1, 0) are requested; MFMAs, with the DMA of chunk c + 3 issued piecewise behind the tile rows.
    // Chunks past the end of the slab are fetched (and computed: the chunk loop runs in rounds of four stages) all the same:
    // zeros through the descriptor's bounds check, no traffic, and the in-order vmcnt bookkeeping stays a constant.
#pragma unroll
    for (int ch = 0; ch < (DLSA_CYC_PAIR ? 2 : 3); ++ch) {
#pragma unroll
        for (int pc = 0; pc < PIECES; ++pc) dma_piece(ch, ch, pc);
        dma_w(ch, ch);
    }
#if DLSA_CYC_FLAGS
    // Per-stage hand-off flags instead of the workgroup barrier: byte w of LANDED[s] = tag of the chunk whose pieces wave w has
    // seen land in stage s; byte w of DONE[s] = tag of the chunk wave w has finished reading there.  tag(c) = (c / 4 + 1) & 255.
    // A wave reads a chunk once all eight LANDED bytes carry its tag, and overwrites a stage once all eight DONE bytes do.
    unsigned long long* const flagL = reinterpret_cast<unsigned long long*>(lds + CYC_NST * BUF);      // [4]
    unsigned long long* const flagD = flagL + 4;                                                        // [4]
    if (tid < 8) flagL[tid] = 0ull;
    __syncthreads();
    auto tag_of = [](int c) { return (unsigned)(((c >> 2) + 1) & 255); };
    auto post = [&](unsigned long long* f, int stage, int c) {
        if (lane == 0) reinterpret_cast<volatile unsigned char*>(f + stage)[wave] = (unsigned char)tag_of(c);
    };
    auto await = [&](unsigned long long* f, int stage, int c) {
        const unsigned long long want = 0x0101010101010101ull * tag_of(c);
        while (__builtin_amdgcn_readfirstlane((int)(*reinterpret_cast<volatile unsigned long long*>(f + stage) != want)))
            __builtin_amdgcn_s_sleep(1);
    };
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * DMA_PER_CHUNK) : "memory");
    post(flagL, 0, 0);
    await(flagL, 0, 0);
#else
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DLSA_CYC_PAIR ? 0 : 2 * DMA_PER_CHUNK) : "memory");
    asm volatile("s_barrier" ::: "memory");
#endif

    // Per-lane BYTE addresses of the fragments inside stage pair sp (stages 2 sp, 2 sp + 1): row (lane >> 4) of a k-step,
    // column 16 tile + (lane & 15).  Stage parity and k-step are compile-time immediates of the ds_read (the chunk loop is
    // unrolled over the four stages), so a fragment read costs no address arithmetic in the loop.
    const int lane_part = ((lane >> 4) * LDP + (lane & 15)) * 8;
    int adA[2][RW], adB[2][NB], adT[2], adW[2];
#pragma unroll
    for (int sp = 0; sp < 2; ++sp) {
        const int sb = sp * 2 * BUF * 8;
        adA[sp][0] = sb + lane_part + 128 * i0;
        adA[sp][1] = sb + lane_part + 128 * (i0 + 1);                    // row block 15: tile 31 = the corner (columns 496 ..), tail only
#pragma unroll
        for (int j = 0; j < NB; ++j) adB[sp][j] = sb + lane_part + 128 * ((i0 + d0 + j) % CYC_NT);
        adT[sp] = sb + ((lane >> 4) * LDP + CYC_C0 + (lane & 3)) * 8;      // tail columns, broadcast to the 4 blocks
        adW[sp] = sb + (KC * LDP + (lane >> 4)) * 8;
    }
    const char* ldsb = (const char*)lds;
    struct Frag { double fa[RW], fb[NB], bt[GA], wv; };
    auto ld = [&](int addr, int imm) { return *(const double*)(ldsb + addr + imm); };
    // fragments of k-step ks of the chunk in stage ST (both compile-time after unrolling)
    auto load_frags = [&](int ST, int ks, Frag& f) {
        const int sp = ST >> 1, imm = ((ST & 1) * BUF + ks * 4 * LDP) * 8;
#pragma unroll
        for (int r = 0; r < RW; ++r) f.fa[r] = ld(adA[sp][r], imm);
#pragma unroll
        for (int j = 0; j < NB; ++j) f.fb[j] = ld(adB[sp][j], imm);
        if (tails) {
#pragma unroll
            for (int g = 0; g < G; ++g) f.bt[g] = ld(adT[sp], imm + 32 * g);
        }
        f.wv = HASW ? ld(adW[sp], ((ST & 1) * BUF + ks * 4) * 8) : 1.0;
    };
    auto kstep = [&](const Frag& f, int chunk_dma, int buf_dma, bool issue) {
        // the weight goes on the A side: 2 multiplications per k-step instead of 9 (and the tail rows come scaled for free)
        double aw[RW];
#pragma unroll
        for (int r = 0; r < RW; ++r) aw[r] = HASW ? f.fa[r] * f.wv : f.fa[r];
        cyc_row<0>(aw[0], f.fb[0], f.fb[1], f.fb[2], f.fb[3], f.fb[4], f.fb[5], f.fb[6], f.fb[7]);
        if (issue) {
#pragma unroll
            for (int k = 0; k < PPR; ++k) dma_piece(chunk_dma, buf_dma, k);
        }
        // Row block 15 has no second tile row.  With tail columns (G > 0) its waves run the 8 MFMAs all the same, on the corner
        // fragment, and drop the result: with one light SIMD that workgroup ran ~1 % faster than its three partners, ended
        // 400 us ahead of them (bench/cyc_drift.py) and took its rows out of their L2 window -- fabric traffic 1.8x, 97.7
        // instead of 94.9 ms at p = 500.  Without tail columns the four stay within 3 us of each other either way and the
        // dummy MFMAs only cost power (92.8 vs 90.8 ms at p = 496), so there the row is skipped.  (An explicit meeting of
        // the four workgroups every 64 chunks -- arrival counter + bounded poll -- was measured too: no gain over this.)
        if (G > 0 || two_rows)
            cyc_row<1>(aw[1], f.fb[1], f.fb[2], f.fb[3], f.fb[4], f.fb[5], f.fb[6], f.fb[7], f.fb[8]);
        if (issue) {
#pragma unroll
            for (int k = 0; k < PPR; ++k) dma_piece(chunk_dma, buf_dma, PPR + k);
            dma_w(chunk_dma, buf_dma);
        }
        // The tail columns: 2 G small MFMAs on the A fragments the wave has just scaled.  (Measured same-box at p = 500: issuing
        // them in batches of 2 / 4 k-steps, or one tile row per wave of the SIMD pair instead of both on the d < 8 wave: 0.5-2.5 %
        // slower each; G = 1 and G = 2 cost the same.)
        if constexpr (G > 0) {
            if (tails) cyc_tail_a<RW, G>(aw, f.bt);
        }
    };

    Frag fr0, fr1;
    load_frags(0, 0, fr0);
    if (DLSA_CYC_PRIO == 1 && db) __builtin_amdgcn_s_setprio(1);
    if (DLSA_CYC_PRIO == 2 && !db) __builtin_amdgcn_s_setprio(1);
#if DLSA_CYC_FLAGS
    for (int c4 = 0; c4 < nchunks; c4 += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {                        // chunk c = c4 + u sits in stage u
            const int c = c4 + u;
            load_frags(u, 1, fr1);                               // (c, 1), while (c, 0) computes
            if (c > 0) post(flagD, (u + 3) & 3, c - 1);          // this wave's reads of chunk c - 1 were consumed by the MFMAs it has issued
            __builtin_amdgcn_sched_barrier(0);
            {
                double aw[RW];
#pragma unroll
                for (int r = 0; r < RW; ++r) aw[r] = HASW ? fr0.fa[r] * fr0.wv : fr0.fa[r];
                cyc_row<0>(aw[0], fr0.fb[0], fr0.fb[1], fr0.fb[2], fr0.fb[3], fr0.fb[4], fr0.fb[5], fr0.fb[6], fr0.fb[7]);
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DMA_PER_CHUNK) : "memory");      // this wave's pieces of chunk c + 1 have landed
                post(flagL, (u + 1) & 3, c + 1);
                __builtin_amdgcn_sched_barrier(0);
                if (G > 0 || two_rows)
                    cyc_row<1>(aw[1], fr0.fb[1], fr0.fb[2], fr0.fb[3], fr0.fb[4], fr0.fb[5], fr0.fb[6], fr0.fb[7], fr0.fb[8]);
                if constexpr (G > 0) {
                    if (tails) cyc_tail_a<RW, G>(aw, fr0.bt);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            await(flagL, (u + 1) & 3, c + 1);                    // every wave's pieces of chunk c + 1 are there
            load_frags((u + 1) & 3, 0, fr0);                     // (c + 1, 0), while (c, 1) computes
            __builtin_amdgcn_sched_barrier(0);
            if (c > 0) await(flagD, (u + 3) & 3, c - 1);         // every wave has left chunk c - 1: its stage takes chunk c + 3
            __builtin_amdgcn_sched_barrier(0);
            kstep(fr1, c + 3, (u + 3) & 3, true);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#elif DLSA_CYC_PAIR
    // One barrier per PAIR of chunks: pair (c, c + 1) sits in stages (u, u + 1), the DMA of the next pair goes into the other two
    // stages behind the tile rows of this pair's first two k-steps and is waited for (vmcnt 0) at the pair's end.  Half the
    // barriers: the skew between the four SIMDs of a workgroup is paid once per 16 rows instead of once per 8.
    for (int c4 = 0; c4 < nchunks; c4 += 4) {
#pragma unroll
        for (int u = 0; u < 4; u += 2) {
            const int c = c4 + u;
            load_frags(u, 1, fr1);
            __builtin_amdgcn_sched_barrier(0);
            kstep(fr0, c + 2, (u + 2) & 3, true);
            __builtin_amdgcn_sched_barrier(0);
            load_frags(u + 1, 0, fr0);
            __builtin_amdgcn_sched_barrier(0);
            kstep(fr1, c + 3, (u + 3) & 3, true);
            __builtin_amdgcn_sched_barrier(0);
            load_frags(u + 1, 1, fr1);
            __builtin_amdgcn_sched_barrier(0);
            kstep(fr0, 0, 0, false);
            __builtin_amdgcn_sched_barrier(0);
            {
                double aw[RW];
#pragma unroll
                for (int r = 0; r < RW; ++r) aw[r] = HASW ? fr1.fa[r] * fr1.wv : fr1.fa[r];
                cyc_row<0>(aw[0], fr1.fb[0], fr1.fb[1], fr1.fb[2], fr1.fb[3], fr1.fb[4], fr1.fb[5], fr1.fb[6], fr1.fb[7]);
                __builtin_amdgcn_sched_barrier(0);
                if (db) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    asm volatile("s_barrier" ::: "memory");
                }
                if (G > 0 || two_rows)
                    cyc_row<1>(aw[1], fr1.fb[1], fr1.fb[2], fr1.fb[3], fr1.fb[4], fr1.fb[5], fr1.fb[6], fr1.fb[7], fr1.fb[8]);
                if constexpr (G > 0) {
                    if (tails) cyc_tail_a<RW, G>(aw, fr1.bt);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (!db) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    asm volatile("s_barrier" ::: "memory");
                }
            }
            load_frags((u + 2) & 3, 0, fr0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#else
    for (int c4 = 0; c4 < nchunks; c4 += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {                        // chunk c4 + u sits in stage u
            load_frags(u, 1, fr1);                               // (c, 1), while (c, 0) computes
            __builtin_amdgcn_sched_barrier(0);
#if DLSA_CYC_SPLIT
            // The two waves of a SIMD (db = 0 / 1) meet the chunk barrier half an MFMA block apart: the db = 1 wave after its
            // first tile row, the db = 0 wave after the whole k-step.  At the same program point both would run their
            // post-barrier reads and DMA issue with the matrix pipe idle; offset, each covers the other (gram_plan_kernel.inc).
            {
                double aw[RW];
#pragma unroll
                for (int r = 0; r < RW; ++r) aw[r] = HASW ? fr0.fa[r] * fr0.wv : fr0.fa[r];
                cyc_row<0>(aw[0], fr0.fb[0], fr0.fb[1], fr0.fb[2], fr0.fb[3], fr0.fb[4], fr0.fb[5], fr0.fb[6], fr0.fb[7]);
                __builtin_amdgcn_sched_barrier(0);
                if (db) {
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DMA_PER_CHUNK) : "memory");
                    asm volatile("s_barrier" ::: "memory");
                }
                if (G > 0 || two_rows)
                    cyc_row<1>(aw[1], fr0.fb[1], fr0.fb[2], fr0.fb[3], fr0.fb[4], fr0.fb[5], fr0.fb[6], fr0.fb[7], fr0.fb[8]);
                if constexpr (G > 0) {
                    if (tails) cyc_tail_a<RW, G>(aw, fr0.bt);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (!db) {
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DMA_PER_CHUNK) : "memory");  // chunk c + 1 has landed (c + 2 may be in flight)
                    asm volatile("s_barrier" ::: "memory");
                }
            }
#else
            kstep(fr0, 0, 0, false);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DMA_PER_CHUNK) : "memory");      // chunk c + 1 has landed (c + 2 may be in flight)
            asm volatile("s_barrier" ::: "memory");
#endif
            load_frags((u + 1) & 3, 0, fr0);                     // (c + 1, 0), while (c, 1) computes
            __builtin_amdgcn_sched_barrier(0);
            kstep(fr1, c4 + u + 3, (u + 3) & 3, true);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the zero-fill DMA of the chunks past the end
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");       // the last MFMAs retire before the accumulators are read

    // epilogue: every tile of the slab's triangle belongs to exactly one wave of the group -- store straight to the partial
    double* __restrict__ P = a.partial + (int64_t)slab * CYC_PP * CYC_PP;
    const int nt_rows = two_rows ? 2 : 1;
    cyc_for_tiles<0, 8 * RW>([&](int k, double (&v)[4]) {
        const int r = k >> 3, cdist = k & 7;
        if (r < nt_rows) {
            const int it = i0 + r, jt = (it + d0 + cdist) % CYC_NT;
#pragma unroll
            for (int q = 0; q < 4; ++q) {                    // C/D register q of lane l = C[4q + (l >> 4)][l & 15]
                const int rr = 16 * it + 4 * q + (lane >> 4), cc = 16 * jt + (lane & 15);
                if (jt >= it) P[(int64_t)rr * CYC_PP + cc] = v[q];
                else P[(int64_t)cc * CYC_PP + rr] = v[q];   // wrapped distance: the tile is the transpose of (jt, it)
            }
        }
    });
    if constexpr (G > 0) {
        // tail accumulator (s, g), lane l: H[16 (i0 + s) + 4 bk + i][496 + 4 g + j], i = l >> 4, bk = (l & 15) >> 2, j = l & 3
        if (tails) cyc_for_tails<0, RW * G>([&](int idx, double v) {
            const int s = idx / G, g = idx - s * G;
            const int row = 16 * (i0 + s) + 4 * ((lane & 15) >> 2) + (lane >> 4), col = CYC_C0 + 4 * g + (lane & 3);
            P[(int64_t)row * CYC_PP + col] = v;
        });
    }
    if (probe && lane == 0) *a.clk = __builtin_readcyclecounter() - t_begin;
}

static int cyc_slabs(int64_t n, int64_t& rows_per_slab) {
    int64_t ns = kNumCU / CYC_GROUP;                         // 64 groups fill the chip once
    while (ns > kNumXCD && n / ns < 4 * CYC_KC) ns -= kNumXCD;
    rows_per_slab = ((n + ns - 1) / ns + CYC_KC - 1) / CYC_KC * CYC_KC;
    return (int)ns;                                          // trailing slabs may be empty: they store zeros
}

bool gram_cyclic_shape_ok(int64_t n, int p) { return p >= CYC_MIN_P && p <= CYC_MAX_P && n >= CYC_MIN_ROWS; }

bool gram_cyclic_eligible(const double* X, int64_t ldx, const double* w, int64_t n, int p) {
    if (!gram_cyclic_shape_ok(n, p)) return false;
    if (gram_dbg_env() & 8) return false;                    // DLSA_GRAM_DBG 8: keep the panel kernel (valid results, A/B runs)
    int64_t rps;
    cyc_slabs(n, rps);
    return ldx % 2 == 0 && ((uintptr_t)X % 16) == 0 && (!w || ((uintptr_t)w % 16) == 0) &&
           (double)(rps + 8 * CYC_KC) * (double)ldx * 8.0 < 2.0e9;            // 32-bit DMA offsets
}

size_t gram_cyclic_ws_bytes(int64_t n, int p) {
    int64_t rps;
    return align_up((size_t)cyc_slabs(n, rps) * CYC_PP * CYC_PP * 8, 256) + kGramProbeBytes;
}

int gram_cyclic_f64(const double* X, int64_t ldx, const double* w, int64_t n, int p, double* H, int64_t ldh,
                    int accumulate, void* ws, size_t ws_bytes, hipStream_t stream) {
    CycArgs a;
    a.X = X; a.w = w; a.partial = (double*)ws; a.ldx = ldx; a.n = n;
    a.p = p + (p & 1);       // odd p in an even row pitch: the pad column only reaches row / column p of H, which nobody reads
    a.nslab = cyc_slabs(n, a.rows_per_slab);
    const size_t need = align_up((size_t)a.nslab * CYC_PP * CYC_PP * 8, 256) + kGramProbeBytes;
    a.clk = (unsigned long long*)((char*)ws + need - kGramProbeBytes);
    if (!ws || ws_bytes < need || ((uintptr_t)ws & 255)) {
        set_error("gram: workspace %zu bytes needed (256-aligned), got %zu", need, ws_bytes);
        return DLSA_ERR_WORKSPACE;
    }
    const int g = a.p <= CYC_C0 ? 0 : (a.p - CYC_C0 + 3) / 4;
    const size_t shm = (size_t)CYC_NST * CYC_BUF * 8 + (DLSA_CYC_FLAGS ? 64 : 0);
    const int blocks = a.nslab * CYC_GROUP;
#define DLSA_LAUNCH_CYC(HW, GV) do { \
        DLSA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(gram_cyclic_kernel<HW, GV>), \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm)); \
        hipLaunchKernelGGL((gram_cyclic_kernel<HW, GV>), dim3(blocks), dim3(512), shm, stream, a); } while (0)
#define DLSA_LAUNCH_CYC_G(HW) do { switch (g) { \
        case 0: DLSA_LAUNCH_CYC(HW, 0); break; case 1: DLSA_LAUNCH_CYC(HW, 1); break; \
        case 2: DLSA_LAUNCH_CYC(HW, 2); break; default: DLSA_LAUNCH_CYC(HW, 3); break; } } while (0)
    if (w) DLSA_LAUNCH_CYC_G(true);
    else DLSA_LAUNCH_CYC_G(false);
    note_gram_kernel(a.clk, stream, "gram_cyclic_kernel<%s,%d>", w ? "true" : "false", g > 3 ? 3 : g);
#undef DLSA_LAUNCH_CYC_G
#undef DLSA_LAUNCH_CYC
    DLSA_HIP_CHECK(hipGetLastError());
    gram_reduce_launch<double>((const double*)ws, a.nslab, CYC_PP, p, H, ldh, accumulate, stream);
    DLSA_HIP_CHECK(hipGetLastError());
    return DLSA_OK;
}

}  // namespace dlsa
