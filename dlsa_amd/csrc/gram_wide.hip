// Wide-p fp32 Gram  H = X' diag(w) X  (the linear-model map step of BASELINE config 5: p = 2000 fp32;
// reference call site dlsa/models.py:130 with w = 1 / README.md:6 for the linear case).
//
// Why a second shape: v_mfma_f32_16x16x4_f32 runs at twice the fp64 rate on elements half the size, so the
// 128x128 output tile of gram.hip has an arithmetic intensity of 32 flop/B against what it stages -- at the
// 157 TF fp32 peak that is 4.9 TB/s of panel traffic, and with 16 panels the slab's working set no longer
// lives in one XCD's L2 (measured: 18 % hit rate, 15x the algorithmic bytes, MFMA pipe 69 % busy).  Here
//   * the columns are cut into GROUPS of 64 (+ one PLAIN tile of the last <= 16 columns when p mod 64 <= 16: p = 2000 is
//     31 groups + 16 columns).  A UNIT is the 64 x 64 block of H of a group pair (ga <= gb): 16 MFMA tiles = 64 accumulator
//     VGPRs.  The tiles of a group are INTERLEAVED: lane m of tile e holds column 4 m + e, so ONE ds_read_b128 of the
//     natural row layout is the fragment of all four tiles;
//   * a WORKGROUP ITEM stages up to 8 groups (two virtual panels of 4 x 64 columns: 64 flop per staged byte) for one slab
//     of rows and its 8 waves hold up to TWO units each.  Round 4: ANY 8 groups and ANY units among them -- the lanes of
//     the row DMA carry per-group global offsets -- instead of two aligned 256-column panels and a 4 x 8-tile wave block.
//     The planner (build_wide_items) covers the group-pair triangle exactly: panel pairs as before (16 units), and the
//     within-panel and leftover-group units packed 8 or 16 to an item by a seeded randomised greedy.  p = 2000: 31.0 item
//     equivalents instead of 34 (no padded tile column, 10 instead of 12 tile slots per diagonal 4 x 4 -- the diagonal units
//     still compute their lower halves: 2.3 %); the plain tile's 31 four-tile column strips ride as a 4-MFMA QUARTER unit
//     on the waves of four items (+ 0.5 item equivalents);
//   * staging is global->LDS DMA (buffer_load_dwordx4 ... lds, one 1 KiB panel row per wave instruction), FOUR 16-row
//     stages (135 KB LDS, one workgroup per CU).  One raw s_barrier per chunk certifies chunk c + 2 (every wave has waited for
//     its own pieces) and frees chunk c's stage: chunk c + 1 was certified one barrier earlier, so its first fragments are
//     fetched BEFORE the barrier and the MFMAs restart at once behind it; the DMA of chunk c + 3 is issued one row at a
//     time behind the k-steps instead of as a burst behind the barrier.
// Needs 16-byte aligned rows (ldx % 4 == 0, p % 4 == 0); everything else stays on gram.hip's kernels.
#include "common.h"
#include <vector>
#include <algorithm>
#include <mutex>
#include <map>
#include <stdlib.h>

namespace dlsa {

constexpr int WGRP = 64;               // columns per group = 4 interleaved tiles
constexpr int WKC = 16;                // rows per stage
constexpr int WLDP = 256;              // LDS row pitch in floats: ds_read_b128 lane groups mix two rows -> 256-byte multiples, no padding
#ifndef DLSA_WIDE_STAGES
#define DLSA_WIDE_STAGES 4
#endif
constexpr int WSTAGES = DLSA_WIDE_STAGES;
constexpr int WWAVES = 8;
constexpr int WTHREADS = 64 * WWAVES;
constexpr int WPANEL_ELEMS = WKC * WLDP;
constexpr int WQ_OFF = 2 * WPANEL_ELEMS;               // the plain tile's 16 x 16 floats
constexpr int WW_OFF = WQ_OFF + WKC * 16;              // the w chunk (16 floats)
constexpr int WBUF_ELEMS = WW_OFF + WKC;
static_assert(WLDP == 4 * WGRP && WSTAGES >= 3 && (size_t)WSTAGES * WBUF_ELEMS * 4 <= (size_t)kLdsBytes, "stages fit the LDS");

struct WideWave {
    int offA[2], offB[2];              // unit u: LDS float offsets (inside a stage) of its A and B groups
    int row0[2], col0[2];              // its block of H (first row / column); row0 < 0: no unit in this slot
    int qoffA, qrow0;                  // quarter unit: A group's offset and first row of H; qrow0 < 0: none
    int qq;                            // this wave stores the plain tile's own 16 x 16 block
    int pad;
};
struct WideItem {
    int gcol[8];                       // first global column of the staged groups (slot s = lane >> 4 of panel s >> 2); -1: empty slot
    int qcol;                          // first column of the plain tile if the item stages it, else -1
    int nfull;                         // units per wave (1 or 2): what the k-step loop runs
    int cost;                          // MFMAs per wave and k-step
    int pad;
    WideWave w[WWAVES];
};

struct WideArgs {
    const float* X;
    const float* w;
    float* partial;                    // [nslab][PP][PP], PP = p rounded up to 64
    const WideItem* items;
    int64_t ldx, n, rows_per_slab;
    int p, PP, nitems, nslab, xcd_map;
    int dbg;                           // DLSA_GRAM_DBG (timing experiments only): 1 = no DMA after the prologue, 16 = no barrier
};

typedef float wacc_t __attribute__((ext_vector_type(4)));

template <bool HASW>
__global__ __launch_bounds__(WTHREADS, 1) void gram_wide_f32_kernel(WideArgs a) {
    __shared__ __attribute__((aligned(16))) float lds[WSTAGES * WBUF_ELEMS];
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    int item_id, slab;
    {
        const int b = blockIdx.x;
        if (a.xcd_map) {
            const int xcd = b % kNumXCD, j = b / kNumXCD;
            item_id = j % a.nitems;
            slab = (j / a.nitems) * kNumXCD + xcd;
        } else {
            item_id = b % a.nitems;
            slab = b / a.nitems;
        }
    }
    const WideItem* __restrict__ it = a.items + item_id;
    const WideWave ww = it->w[wave];
    const int nfull = it->nfull;                                       // workgroup-uniform
    const bool hasq = it->qcol >= 0;
    const int lane_off4 = (lane >> 4) * WLDP + (lane & 15) * 4;        // a lane reads 4 consecutive columns per fragment read

    const int64_t rbeg = (int64_t)slab * a.rows_per_slab;
    const int64_t rend = min(rbeg + a.rows_per_slab, a.n);
    const int64_t nrows = rend > rbeg ? rend - rbeg : 0;
    const int nchunks = (int)((nrows + WKC - 1) / WKC);

    const unsigned xbytes = nrows > 0 ? (unsigned)(((nrows - 1) * a.ldx + a.p) * (int64_t)sizeof(float)) : 0u;
    __amdgpu_buffer_rsrc_t rsrcX = __builtin_amdgcn_make_buffer_rsrc((void*)(a.X + rbeg * a.ldx), 0, (int)xbytes, 0x00020000);
    __amdgpu_buffer_rsrc_t rsrcW = __builtin_amdgcn_make_buffer_rsrc((void*)(HASW ? a.w + rbeg : a.X), 0,
                                                                     HASW ? (int)(nrows * sizeof(float)) : 0, 0x00020000);
    // lanes whose 4 columns lie past p (or in an empty slot) never write: zero all stages once
    for (int e = tid; e < WSTAGES * WBUF_ELEMS; e += WTHREADS) lds[e] = 0.f;
    // the lane's global byte offsets inside a row: its group's first column + its 4 columns (masked-off lanes: an offset beyond
    // the buffer -> the DMA returns zeros for them; p % 4 == 0: a lane is all in or all out)
    constexpr int OOB = 0x7ffffff0;
    int voff[2];
#pragma unroll
    for (int pn = 0; pn < 2; ++pn) {
        const int gc = it->gcol[4 * pn + (lane >> 4)];
        const int col = gc + (lane & 15) * 4;
        voff[pn] = (gc >= 0 && col + 3 < a.p) ? col * 4 : OOB;
    }
    const int qc = it->qcol + lane * 4;
    const int voffq = (hasq && lane < 4 && qc + 3 < a.p) ? qc * 4 : OOB;

    // every wave stages WKC / 8 = 2 rows of a chunk: per row one DMA per virtual panel (+ the plain tile's 64 bytes); wave 0 the weights
    constexpr int RPW = WKC / WWAVES;
    auto stage_row = [&](int chunk, int stage, int r2) {
        float* base = lds + stage * WBUF_ELEMS;
        const int row = wave * RPW + r2;
        const int soff = (int)(((int64_t)chunk * WKC + row) * a.ldx * (int64_t)sizeof(float));
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcX, (lds_ptr_t)(base + row * WLDP), 16, voff[0], soff, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcX, (lds_ptr_t)(base + WPANEL_ELEMS + row * WLDP), 16, voff[1], soff, 0, 0);
        if (hasq && lane < 4) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcX, (lds_ptr_t)(base + WQ_OFF + row * 16), 16, voffq, soff, 0, 0);
    };
    auto stage_w = [&](int chunk, int stage) {
        if (HASW && wave == 0 && lane < WKC / 4)       // exec-masked: the other lanes must not write past the 16 floats
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcW, (lds_ptr_t)(lds + stage * WBUF_ELEMS + WW_OFF), 16, lane * 16,
                                                     chunk * WKC * (int)sizeof(float), 0, 0);
    };
    // wait until only this wave's pieces of the newest `keep` chunks may still be in flight (pieces per chunk: 2 rows x (2 | 3), + w)
    auto wait_keep = [&](auto kc) {
        constexpr int keep = decltype(kc)::value;
        if (hasq) {
            if (HASW && wave == 0) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(keep * (3 * RPW + 1)) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(keep * 3 * RPW) : "memory");
        } else {
            if (HASW && wave == 0) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(keep * (2 * RPW + 1)) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(keep * 2 * RPW) : "memory");
        }
    };

    __syncthreads();                                   // zero fill done before the first DMA lands
    // chunks 0 .. WSTAGES - 2 in flight (past the slab's end: bounds-checked zeros, no traffic -- the counts stay uniform)
#pragma unroll
    for (int c0 = 0; c0 < WSTAGES - 1; ++c0) {
#pragma unroll
        for (int r2 = 0; r2 < RPW; ++r2) stage_row(c0, c0, r2);
        stage_w(c0, c0);
    }
    wait_keep(std::integral_constant<int, WSTAGES - 3>{});      // chunks 0 and 1 have landed (this wave's pieces) ...
    asm volatile("s_barrier" ::: "memory");                     // ... every wave's

    // NU units per wave (the item's), Q: the item carries quarter units
    auto run = [&](auto nuc, auto qcst) {
        constexpr int NU = decltype(nuc)::value;
        constexpr bool Q = decltype(qcst)::value;
        wacc_t acc[NU][4][4];
        wacc_t accq[Q ? 4 : 1], accqq = wacc_t{0, 0, 0, 0};
#pragma unroll
        for (int u = 0; u < NU; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[u][i][e] = wacc_t{0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < (Q ? 4 : 1); ++i) accq[i] = wacc_t{0, 0, 0, 0};

        wacc_t a4[2][NU], b4[2][NU], aq = wacc_t{0, 0, 0, 0};
        float bq = 0.f, wv[2] = {1.f, 1.f};
        auto fetch = [&](int stage, int ks, int slot) {
            const float* base = lds + stage * WBUF_ELEMS;
            const float* kb = base + ks * 4 * WLDP + lane_off4;
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                a4[slot][u] = *reinterpret_cast<const wacc_t*>(kb + ww.offA[u]);
                b4[slot][u] = *reinterpret_cast<const wacc_t*>(kb + ww.offB[u]);
            }
            if (HASW) wv[slot] = base[WW_OFF + ks * 4 + (lane >> 4)];
        };
        // the quarter unit's fragments are asked for at the top of their own k-step: the two units' 32 MFMAs run before they are used
        auto fetch_q = [&](int stage, int ks) {
            const float* base = lds + stage * WBUF_ELEMS;
            aq = *reinterpret_cast<const wacc_t*>(base + ks * 4 * WLDP + lane_off4 + ww.qoffA);
            bq = base[WQ_OFF + ks * 64 + lane];                         // 4 rows x 16 columns: lane (k, n)
        };
        fetch(0, 0, 0);
        int st = 0;                                                    // stage of chunk c
        for (int c = 0; c < nchunks; ++c) {
            const int st_next = st + 1 == WSTAGES ? 0 : st + 1, st_dma = st == 0 ? WSTAGES - 1 : st - 1;     // chunk c + WSTAGES - 1 -> the stage chunk c - 1 has left
#pragma unroll
            for (int ks = 0; ks < WKC / 4; ++ks) {
                const int cur = ks & 1;
                // the next k-step's fragments -- the next CHUNK's first ones at the last k-step: that chunk was certified a barrier ago
                if (ks + 1 < WKC / 4) fetch(st, ks + 1, cur ^ 1);
                else fetch(st_next, 0, cur ^ 1);
                if constexpr (Q) fetch_q(st, ks);
                __builtin_amdgcn_sched_barrier(0);     // keep the prefetch ahead of the MFMA block (the scheduler sinks it otherwise)
                if (HASW) {
#pragma unroll
                    for (int u = 0; u < NU; ++u) a4[cur][u] *= wv[cur];
                }
#pragma unroll
                for (int u = 0; u < NU; ++u) {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            acc[u][i][e] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[cur][u][i], b4[cur][u][e], acc[u][i][e], 0, 0, 0);
                    if (u == 0) {
                        // this wave's share of the DMA of chunk c + WSTAGES - 1: one row behind the first unit of k-steps 0 and 1, w behind k-step 2's
                        __builtin_amdgcn_sched_barrier(0);
                        if (!DLSA_DBG_WRONG(a.dbg, 1)) {
                            if (ks < RPW) stage_row(c + WSTAGES - 1, st_dma, ks);
                            else if (ks == RPW) stage_w(c + WSTAGES - 1, st_dma);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                if constexpr (Q) {
                    const float bqw = HASW ? bq * wv[cur] : bq;        // the weights on the B side here: one multiplication instead of five
#pragma unroll
                    for (int i = 0; i < 4; ++i) accq[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[i], bqw, accq[i], 0, 0, 0);
                    accqq = __builtin_amdgcn_mfma_f32_16x16x4f32(bq, bqw, accqq, 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            // chunk c + 2 has landed (this wave's pieces; c + 3 may be in flight), then for every wave; and every wave has left chunk c
            wait_keep(std::integral_constant<int, WSTAGES - 3>{});
            if (!DLSA_DBG_WRONG(a.dbg, 16)) asm volatile("s_barrier" ::: "memory");
            st = st_next;
        }

        // epilogue: each unit's 64 x 64 block -> this slab's partial buffer (leading dimension PP >= every staged column, so no bounds
        // checks; entries below the diagonal or past p are never read by the reduce kernel).
        // acc[u][i][e][r] is H[row0 + 4 (4 (lane >> 4) + r) + i][col0 + 4 (lane & 15) + e]  (fp32 C/D layout: tile row = 4 (lane >> 4) + r,
        // tile column = lane & 15; tile i of the A group holds the rows 4 m + i, tile e of the B group the columns 4 m + e)
        float* __restrict__ P = a.partial + (int64_t)slab * a.PP * a.PP;
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            if (ww.row0[u] < 0) continue;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float* dst = P + (int64_t)(ww.row0[u] + 4 * (4 * (lane >> 4) + r) + i) * a.PP + ww.col0[u] + 4 * (lane & 15);
                    const wacc_t v = {acc[u][i][0][r], acc[u][i][1][r], acc[u][i][2][r], acc[u][i][3][r]};
                    *reinterpret_cast<wacc_t*>(dst) = v;
                }
        }
        if constexpr (Q) {
            const int qcol = it->qcol;
            if (ww.qrow0 >= 0) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        P[(int64_t)(ww.qrow0 + 4 * (4 * (lane >> 4) + r) + i) * a.PP + qcol + (lane & 15)] = accq[i][r];
            }
            if (ww.qq) {
#pragma unroll
                for (int r = 0; r < 4; ++r) P[(int64_t)(qcol + 4 * (lane >> 4) + r) * a.PP + qcol + (lane & 15)] = accqq[r];
            }
        }
    };
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    if (nfull == 2) { if (hasq) run(I2{}, std::true_type{}); else run(I2{}, std::false_type{}); }
    else { if (hasq) run(I1{}, std::true_type{}); else run(I1{}, std::false_type{}); }
}

template <typename T>
void gram_reduce_launch(const T* partial, int nslab, int PP, int p, T* H, int64_t ldh, int accumulate, hipStream_t stream);   // gram.hip
void gram_reduce_launch_f32_to_f64(const float* partial, int nslab, int PP, int p, double* H, int64_t ldh, int accumulate, hipStream_t stream);

// -------------------------------------------------------------------------------------------------
// host: plan
// -------------------------------------------------------------------------------------------------
struct WideShape { int G, plain_col; };           // groups (the last one may be partial), first column of the plain tile or -1
static WideShape wide_shape(int p) {
    WideShape s;
    s.G = p / WGRP;
    const int rem = p % WGRP;
    s.plain_col = -1;
    if (rem > 16) ++s.G;                           // a partial group (its lanes past p are masked)
    else if (rem > 0) s.plain_col = s.G * WGRP;    // <= 16 columns: the plain tile
    return s;
}

struct HostUnit { int ga, gb; };
struct HostItem { std::vector<int> groups; std::vector<HostUnit> units; std::vector<int> qgroups; bool qq = false; };

// units of one item onto its 8 waves, LDS slots for its groups
static WideItem finish_item(const HostItem& h, const WideShape& sh) {
    WideItem g{};
    auto slot_off = [](int s) { return (s >> 2) * WPANEL_ELEMS + (s & 3) * WGRP; };
    auto slot_of = [&](int grp) { for (size_t s = 0; s < h.groups.size(); ++s) if (h.groups[s] == grp) return (int)s; return -1; };
    for (int s = 0; s < 8; ++s) g.gcol[s] = s < (int)h.groups.size() ? h.groups[s] * WGRP : -1;
    g.qcol = h.qgroups.empty() && !h.qq ? -1 : sh.plain_col;
    g.nfull = h.units.size() > WWAVES ? 2 : 1;
    g.cost = 16 * g.nfull + (g.qcol >= 0 ? 5 : 0);
    for (int wv = 0; wv < WWAVES; ++wv) {
        WideWave& w = g.w[wv];
        for (int u = 0; u < 2; ++u) {
            const size_t k = (size_t)wv + (size_t)u * WWAVES;      // round-robin: waves 0 .. get the first eight units, then the second eight
            if (k < h.units.size()) {
                w.offA[u] = slot_off(slot_of(h.units[k].ga)); w.offB[u] = slot_off(slot_of(h.units[k].gb));
                w.row0[u] = h.units[k].ga * WGRP; w.col0[u] = h.units[k].gb * WGRP;
            } else {
                w.offA[u] = w.offB[u] = 0; w.row0[u] = w.col0[u] = -1;                        // computes garbage, stores nothing
            }
        }
        if (wv < (int)h.qgroups.size()) { w.qoffA = slot_off(slot_of(h.qgroups[wv])); w.qrow0 = h.qgroups[wv] * WGRP; }
        else { w.qoffA = 0; w.qrow0 = -1; }
        w.qq = (h.qq && wv == 0) ? 1 : 0;
    }
    return g;
}

static void build_wide_items(int p, std::vector<WideItem>& items) {
    const WideShape sh = wide_shape(p);
    const int G = sh.G, P = G / 4;
    std::vector<HostItem> host;
    // full panel pairs: 4 x 4 groups = 16 units; wave wv: A group wv >> 1, B groups 2 (wv & 1), + 1 (round-robin order of finish_item)
    for (int x = 0; x < P; ++x)
        for (int y = x + 1; y < P; ++y) {
            HostItem h;
            for (int s = 0; s < 4; ++s) h.groups.push_back(4 * x + s);
            for (int s = 0; s < 4; ++s) h.groups.push_back(4 * y + s);
            for (int u = 0; u < 2; ++u)
                for (int wv = 0; wv < WWAVES; ++wv) h.units.push_back(HostUnit{4 * x + (wv >> 1), 4 * y + 2 * (wv & 1) + u});
            host.push_back(h);
        }
    // loose units: inside a panel, and every pair with a leftover group -- packed 16 (or 8) to an item of <= 8 groups by a
    // randomised greedy (seeded, so the plan is a function of p); cost = sum over items of the units per wave
    std::vector<HostUnit> loose0;
    for (int x = 0; x < P; ++x)
        for (int a = 4 * x; a < 4 * x + 4; ++a)
            for (int b = a; b < 4 * x + 4; ++b) loose0.push_back(HostUnit{a, b});
    for (int l = 4 * P; l < G; ++l)
        for (int g = 0; g < G; ++g)
            if (g < 4 * P || g >= l) loose0.push_back(HostUnit{std::min(l, g), std::max(l, g)});
    unsigned long long rng = 0x9E3779B97F4A7C15ull ^ (unsigned long long)p;
    auto rnd = [&]() { rng = rng * 6364136223846793005ull + 1442695040888963407ull; return (unsigned)(rng >> 33); };
    std::vector<HostItem> best;
    int best_cost = 1 << 30;
    for (int restart = 0; restart < 64 && !loose0.empty(); ++restart) {
        std::vector<HostUnit> loose = loose0;
        std::vector<HostItem> cur;
        int cost = 0;
        while (!loose.empty()) {
            std::vector<int> deg(G, 0);
            for (auto& u : loose) { ++deg[u.ga]; if (u.gb != u.ga) ++deg[u.gb]; }
            // the item's groups: seed = a group with the most loose units, then whichever adds the most units inside the set
            std::vector<int> S;
            {
                int mx = 0;
                for (int g = 0; g < G; ++g) mx = std::max(mx, deg[g]);
                std::vector<int> c;
                for (int g = 0; g < G; ++g) if (deg[g] == mx) c.push_back(g);
                S.push_back(restart ? c[rnd() % c.size()] : c[0]);
            }
            while (S.size() < 8) {
                std::vector<int> gain(G, 0);
                std::vector<char> in(G, 0);
                for (int g : S) in[g] = 1;
                for (auto& u : loose) {
                    if (u.ga == u.gb) { if (!in[u.ga]) ++gain[u.ga]; }
                    else if (in[u.ga] && !in[u.gb]) ++gain[u.gb];
                    else if (in[u.gb] && !in[u.ga]) ++gain[u.ga];
                }
                int mx = 0;
                for (int g = 0; g < G; ++g) mx = std::max(mx, gain[g]);
                if (mx == 0) break;
                std::vector<int> c;
                for (int g = 0; g < G; ++g) if (gain[g] == mx) c.push_back(g);
                S.push_back(restart ? c[rnd() % c.size()] : c[0]);
            }
            std::vector<char> in(G, 0);
            for (int g : S) in[g] = 1;
            std::vector<int> avail;                     // indices into loose
            for (size_t k = 0; k < loose.size(); ++k) if (in[loose[k].ga] && in[loose[k].gb]) avail.push_back((int)k);
            const int target = avail.size() >= 16 ? 16 : (avail.size() >= 8 ? 8 : (int)avail.size());
            std::vector<char> taken(loose.size(), 0);
            int ntaken = 0;
            while (ntaken < target) {
                // finish a whole group if one fits (it never has to be staged again) -- the largest such; else units of the busiest groups
                const int cap = target - ntaken;
                int pick = -1, pick_n = 0;
                for (int g : S) {
                    int mine = 0, tot = 0;
                    for (size_t k = 0; k < loose.size(); ++k) {
                        if (taken[k] || (loose[k].ga != g && loose[k].gb != g)) continue;
                        ++tot;
                        if (in[loose[k].ga] && in[loose[k].gb]) ++mine;
                    }
                    if (mine > 0 && mine == tot && mine <= cap && (mine > pick_n || (mine == pick_n && restart && (rnd() & 1)))) { pick = g; pick_n = mine; }
                }
                if (pick >= 0) {
                    for (size_t k = 0; k < loose.size(); ++k)
                        if (!taken[k] && (loose[k].ga == pick || loose[k].gb == pick)) { taken[k] = 1; ++ntaken; }
                    continue;
                }
                int bk = -1, bscore = -1;
                for (int k : avail) {
                    if (taken[k]) continue;
                    const int score = (deg[loose[k].ga] + deg[loose[k].gb]) * 4 + (restart ? (int)(rnd() & 3) : 0);
                    if (score > bscore) { bscore = score; bk = k; }
                }
                taken[bk] = 1; ++ntaken;
            }
            HostItem h;
            h.groups = S;
            std::vector<HostUnit> rest;
            for (size_t k = 0; k < loose.size(); ++k) (taken[k] ? h.units : rest).push_back(loose[k]);
            loose.swap(rest);
            cost += h.units.size() > WWAVES ? 2 : 1;
            cur.push_back(h);
        }
        if (cost < best_cost) { best_cost = cost; best.swap(cur); }
        if (best_cost * 8 <= (int)loose0.size() + 7) break;          // the lower bound
    }
    for (auto& h : best) host.push_back(h);
    // the plain tile: one quarter unit per group (its column strip against the group's 64 rows), eight to an item -- every wave of such
    // an item runs the four extra MFMAs, so they go to as few items as cover all the groups (greedy set cover, two-unit items first);
    // the plain tile's own 16 x 16 block rides with the first of them
    if (sh.plain_col >= 0) {
        std::vector<char> done(G, 0);
        int left = G;
        bool qq_placed = false;
        while (left > 0) {
            int bi = -1, bn = 0;
            for (size_t k = 0; k < host.size(); ++k) {
                if (!host[k].qgroups.empty()) continue;
                int nn = 0;
                for (int g : host[k].groups) if (!done[g]) ++nn;
                if (nn > bn || (nn == bn && nn > 0 && bi >= 0 && host[k].units.size() > host[bi].units.size())) { bn = nn; bi = (int)k; }
            }
            for (int g : host[bi].groups) if (!done[g]) { host[bi].qgroups.push_back(g); done[g] = 1; --left; }
            if (!qq_placed) { host[bi].qq = true; qq_placed = true; }
        }
    }
    items.clear();
    for (auto& h : host) items.push_back(finish_item(h, sh));
    // expensive items first: the cheap ones fill the last round of workgroups
    std::stable_sort(items.begin(), items.end(), [](const WideItem& x, const WideItem& y) { return x.cost > y.cost; });
}

struct WidePlan { int nitems = 0, PP = 0, cost = 0; WideItem* d_items = nullptr; };
static std::mutex g_wide_mu;
static std::map<std::pair<int, int>, WidePlan> g_wide_plans;
static std::map<int, std::pair<int, int>> g_wide_counts;      // p -> (items, summed cost): what the slab choice needs, without a device

static std::pair<int, int> wide_counts(int p) {
    std::lock_guard<std::mutex> lk(g_wide_mu);
    auto f = g_wide_counts.find(p);
    if (f != g_wide_counts.end()) return f->second;
    std::vector<WideItem> items;
    build_wide_items(p, items);
    int cost = 0;
    for (auto& g : items) cost += g.cost;
    return g_wide_counts[p] = std::make_pair((int)items.size(), cost);
}

static int wide_pp(int p) { return (p + WGRP - 1) / WGRP * WGRP; }

static int get_wide_plan(int p, WidePlan& out) {
    int dev = 0;
    DLSA_HIP_CHECK(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(g_wide_mu);
    auto key = std::make_pair(dev, p);
    auto f = g_wide_plans.find(key);
    if (f != g_wide_plans.end()) { out = f->second; return DLSA_OK; }
    std::vector<WideItem> items;
    build_wide_items(p, items);
    WidePlan pl;
    pl.nitems = (int)items.size();
    pl.PP = wide_pp(p);
    for (auto& g : items) pl.cost += g.cost;
    DLSA_HIP_CHECK(hipMalloc((void**)&pl.d_items, items.size() * sizeof(WideItem)));
    DLSA_HIP_CHECK(hipMemcpy(pl.d_items, items.data(), items.size() * sizeof(WideItem), hipMemcpyHostToDevice));
    g_wide_plans[key] = pl;
    out = pl;
    return DLSA_OK;
}

// One workgroup per CU is resident.  Choose the number of slabs (a multiple of 8 for the XCD mapping) so that the summed cost of
// nitems * nslab workgroups fills whole rounds of 256 full-cost workgroups, slabs stay >= 2048 rows, and a slab's bytes fit the
// 32-bit buffer descriptor.
static void choose_wide_slabs(int64_t n, int p, int cost, int& nslab, int64_t& rows_per_slab) {
    const int64_t max_rows = std::max<int64_t>(WKC, (int64_t)(1.9e9 / ((double)p * sizeof(float))) / WKC * WKC);
    int64_t ns_min = std::max<int64_t>(1, (n + max_rows - 1) / max_rows);
    ns_min = (ns_min + kNumXCD - 1) / kNumXCD * kNumXCD;
    const int64_t ns_max = std::max<int64_t>(ns_min, std::min<int64_t>(1024, n / 2048 / kNumXCD * kNumXCD));
    int64_t best = ns_min;
    double best_score = -1.0;
    const int64_t round_cost = (int64_t)kNumCU * 32;                    // a round of two-unit workgroups
    // ... and enough rounds that one workgroup is a small part of a CU's share (a slab count that gives fewer than twelve rounds
    // is marked down in proportion: with four rounds the last one's idle CUs and the items' unequal costs were ~3 %)
    for (int64_t ns = ns_min; ns <= ns_max; ns += kNumXCD) {
        const int64_t work = ns * cost;
        const int64_t rounds = (work + round_cost - 1) / round_cost;
        const double eff = (double)work / (double)(rounds * round_cost);
        const double score = eff * std::min(1.0, 0.88 + 0.01 * (double)rounds);
        if (score > best_score + 0.004) { best_score = score; best = ns; }       // prefer fewer, larger slabs
        if ((rounds >= 12 && eff >= 0.975) || rounds >= 20) break;               // (every slab costs a PP x PP partial: written, then read by the reduce)
    }
    rows_per_slab = ((n + best - 1) / best + WKC - 1) / WKC * WKC;
    if (rows_per_slab < WKC) rows_per_slab = WKC;
    nslab = (int)best;                                                  // trailing slabs may be empty: they write zeros
}

bool gram_wide_f32_shape_ok(int64_t n, int p) {
    return p >= 768 && (p % 4 == 0) && n >= 16384;
}

bool gram_wide_f32_eligible(const float* X, int64_t ldx, const float* w, int64_t n, int p) {
    { const char* e = kernel_knob("DLSA_GRAM_WIDE_F32"); if (e && atoi(e) == 0) return false; }      // dlsa_kernel_options.gram_wide_f32 = 0: the panel kernel (A/B runs, tests)
    if (!gram_wide_f32_shape_ok(n, p) || (ldx % 4) || ((uintptr_t)X & 15) || (w && ((uintptr_t)w & 15))) return false;
    int nslab; int64_t rps;
    choose_wide_slabs(n, p, wide_counts(p).second, nslab, rps);
    return (double)rps * (double)ldx * sizeof(float) < 2.1e9;     // a slab must fit the 32-bit buffer descriptor
}

size_t gram_wide_f32_ws_bytes(int64_t n, int p) {
    int nslab; int64_t rps;
    choose_wide_slabs(n, p, wide_counts(p).second, nslab, rps);
    const size_t PP = (size_t)wide_pp(p);
    return align_up((size_t)nslab * PP * PP * sizeof(float), 256);
}

int gram_wide_f32(const float* X, int64_t ldx, const float* w, int64_t n, int p, float* H, int64_t ldh,
                  int accumulate, void* ws, size_t ws_bytes, hipStream_t stream, double* H64) {
    WidePlan pl;
    int rc = get_wide_plan(p, pl);
    if (rc) return rc;
    int nslab; int64_t rps;
    choose_wide_slabs(n, p, pl.cost, nslab, rps);
    const size_t need = (size_t)nslab * pl.PP * pl.PP * sizeof(float);
    if (!ws || ws_bytes < need || ((uintptr_t)ws & 255)) {
        set_error("gram(f32, wide): workspace %zu bytes needed (256-aligned), got %zu", need, ws_bytes);
        return DLSA_ERR_WORKSPACE;
    }
    WideArgs a;
    a.X = X; a.w = w; a.partial = (float*)ws; a.items = pl.d_items; a.ldx = ldx; a.n = n; a.rows_per_slab = rps;
    a.p = p; a.PP = pl.PP; a.nitems = pl.nitems; a.nslab = nslab; a.xcd_map = (nslab % kNumXCD == 0) ? 1 : 0;
    a.dbg = gram_dbg_env();
    if (a.dbg & 2) a.xcd_map = 0;
    const int blocks = pl.nitems * nslab;
    if (w) hipLaunchKernelGGL((gram_wide_f32_kernel<true>), dim3(blocks), dim3(WTHREADS), 0, stream, a);
    else hipLaunchKernelGGL((gram_wide_f32_kernel<false>), dim3(blocks), dim3(WTHREADS), 0, stream, a);
    note_gram_kernel(nullptr, stream, "gram_wide_f32_kernel<%s>", w ? "true" : "false");
    DLSA_HIP_CHECK(hipGetLastError());
    if (H64) gram_reduce_launch_f32_to_f64((const float*)ws, nslab, pl.PP, p, H64, ldh, accumulate, stream);
    else gram_reduce_launch<float>((const float*)ws, nslab, pl.PP, p, H, ldh, accumulate, stream);
    DLSA_HIP_CHECK(hipGetLastError());
    return DLSA_OK;
}

// host-only self check of the wide plan (CPU test-suite): every 16 x 16 tile on/above the diagonal stored exactly once (a diagonal
// unit also stores its lower half: counted as slots, not as tiles), every unit's groups staged by its item.
// nslots = MFMAs per k-step summed over the waves that run them (what the workgroups execute), ntiles = tiles on/above the diagonal.
int gram_wide_plan_check(int p, int* nitems, int* nslots, int* ntiles) {
    std::vector<WideItem> items;
    build_wide_items(p, items);
    const WideShape sh = wide_shape(p);
    const int ntile = (p + 15) / 16;
    std::vector<int> seen((size_t)ntile * ntile, 0);
    int slots = 0;
    auto lds_slot_col = [&](const WideItem& g, int off) {      // the global column a stage offset stands for, or -1
        const int s = (off / WPANEL_ELEMS) * 4 + (off % WPANEL_ELEMS) / WGRP;
        return (off % WGRP) == 0 && s >= 0 && s < 8 ? g.gcol[s] : -1;
    };
    auto mark = [&](int row0, int col0, int nr, int nc) {      // the 16 x 16 tiles of the block [row0, +nr) x [col0, +nc) in INTERLEAVED order cover it entirely
        for (int ti = row0 / 16; ti < (row0 + nr + 15) / 16; ++ti)
            for (int tj = col0 / 16; tj < (col0 + nc + 15) / 16; ++tj)
                if (ti <= tj && ti < ntile && tj < ntile) ++seen[(size_t)ti * ntile + tj];
    };
    for (auto& g : items) {
        if (g.nfull != 1 && g.nfull != 2) return -6;
        for (int wv = 0; wv < WWAVES; ++wv) {
            const WideWave& w = g.w[wv];
            slots += 16 * g.nfull + (g.qcol >= 0 ? 5 : 0);
            for (int u = 0; u < 2; ++u) {
                if (w.row0[u] < 0) continue;
                if (u >= g.nfull) return -7;
                if (lds_slot_col(g, w.offA[u]) != w.row0[u] || lds_slot_col(g, w.offB[u]) != w.col0[u]) return -4;
                if (w.row0[u] > w.col0[u]) return -1;
                mark(w.row0[u], w.col0[u], WGRP, WGRP);
            }
            if (w.qrow0 >= 0) {
                if (g.qcol != sh.plain_col || g.qcol < 0 || lds_slot_col(g, w.qoffA) != w.qrow0) return -8;
                mark(w.qrow0, g.qcol, WGRP, 16);
            }
            if (w.qq) { if (g.qcol < 0) return -9; mark(g.qcol, g.qcol, 16, 16); }
        }
    }
    int count = 0;
    for (int ti = 0; ti < ntile; ++ti)
        for (int tj = ti; tj < ntile; ++tj) {
            if (seen[(size_t)ti * ntile + tj] != 1) return seen[(size_t)ti * ntile + tj] ? -2 : -3;
            ++count;
        }
    if (nitems) *nitems = (int)items.size();
    if (nslots) *nslots = slots;
    if (ntiles) *ntiles = count;
    return 0;
}

}  // namespace dlsa

extern "C" int dlsa_gram_wide_plan_check(int p, int* nitems, int* nslots, int* ntiles) {
    return dlsa::gram_wide_plan_check(p, nitems, nslots, ntiles);
}
