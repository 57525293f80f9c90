// Wide-p fp32 Gram  H = X' diag(w) X  (the linear-model map step of BASELINE config 5: p = 2000 fp32;
// reference call site dlsa/models.py:130 with w = 1 / README.md:6 for the linear case).
//
// Why a second shape: v_mfma_f32_16x16x4_f32 runs at twice the fp64 rate on elements half the size, so the
// 128x128 output tile of gram.hip has an arithmetic intensity of 32 flop/B against what it stages -- at the
// 157 TF fp32 peak that is 4.9 TB/s of panel traffic, and with 16 panels the slab's working set no longer
// lives in one XCD's L2 (measured: 18 % hit rate, 15x the algorithmic bytes, MFMA pipe 69 % busy).  Here
//   * a PANEL is 256 columns; a WORKGROUP of 8 waves owns a 256x256 block of H for one slab of rows
//     (64 flop per staged byte -> 2.5 TB/s at peak, HBM can feed it even without L2 hits);
//   * a WAVE owns 64 rows x 128 columns of H as 4x8 MFMA tiles (128 accumulator VGPRs).  The tiles are
//     INTERLEAVED: lane m of tile e of a 64-column group holds column 4m+e, so one ds_read_b128 of the
//     natural row layout feeds four MFMAs -- 3 LDS reads (+ w) and 4 v_mul per k-step of 32 MFMAs (with
//     contiguous 16-column tiles it was 6 ds_read2_b32); the epilogue undoes the interleave with 16-byte stores;
//   * staging is global->LDS DMA (buffer_load_dwordx4 ... lds, one 1 KiB panel row per wave instruction),
//     three 16-row stages (105 KB LDS, one workgroup per CU): chunk c+2 is in flight while c is consumed,
//     one barrier per chunk;
//   * diagonal panels are cut into their six 4x8 blocks that touch the upper triangle and packed eight
//     to a workgroup (two panels per workgroup), so p=2000 runs 28 + 6 workgroup items per slab.
// Needs 16-byte aligned rows (ldx % 4 == 0, p % 4 == 0); everything else stays on gram.hip's kernels.
#include "common.h"
#include <vector>
#include <algorithm>
#include <mutex>
#include <map>
#include <stdlib.h>

namespace dlsa {

constexpr int WTILE = 16;
constexpr int WPANEL = 256;            // columns per panel = 16 tiles
#ifndef DLSA_WIDE_KC
#define DLSA_WIDE_KC 16
#endif
#ifndef DLSA_WIDE_STAGES
#define DLSA_WIDE_STAGES 3
#endif
constexpr int WKC = DLSA_WIDE_KC;      // rows per stage
constexpr int WLDP = 256;              // LDS row pitch in floats: ds_read_b128 lane groups mix two rows -> 256-byte multiples, no padding
constexpr int WSTAGES = DLSA_WIDE_STAGES;
constexpr int WAHEAD = WSTAGES - 1;    // chunks in flight ahead of the one being consumed
constexpr int WWAVES = 8;
constexpr int WTHREADS = 64 * WWAVES;
constexpr int WMR = 4, WNR = 8;        // tiles per wave block
constexpr int WPANEL_ELEMS = WKC * WLDP;
constexpr int WBUF_ELEMS = 2 * WPANEL_ELEMS + WKC;     // two panels + the w chunk (16 floats)

struct WideBlock {
    unsigned char selA, ta0;           // A tiles: panel select (0 = panA, 1 = panB), first local tile (0, 4, 8, 12)
    unsigned char selB, tb0;           // B tiles: panel select, first local tile (0 or 8)
    unsigned int mask;                 // bit i*8+j: the 16x16 cell (ta0+i, tb0+j) holds entries on/above the diagonal inside p;
                                       // 0 = idle wave (the kernel computes and stores the whole 64 x 128 block otherwise)
};
struct WideItem {
    int panA, panB;
    WideBlock wb[WWAVES];
};

struct WideArgs {
    const float* X;
    const float* w;
    float* partial;                    // [nslab][PP][PP], PP = panels x 256
    const WideItem* items;
    int64_t ldx, n, rows_per_slab;
    int p, PP, nitems, nslab, xcd_map;
    int dbg;                           // DLSA_GRAM_DBG (timing experiments only): 1 = no DMA after the prologue, 16 = no barrier
};

typedef float wacc_t __attribute__((ext_vector_type(4)));

template <bool HASW>
__global__ __launch_bounds__(WTHREADS, 1) void gram_wide_f32_kernel(WideArgs a) {
    __shared__ __attribute__((aligned(16))) float lds[WSTAGES * WBUF_ELEMS];
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
    const int panA = it->panA, panB = it->panB;
    const WideBlock wb = it->wb[wave];
    const bool active = wb.mask != 0;                                  // wave-uniform
    const int offA = wb.selA * WPANEL_ELEMS + wb.ta0 * WTILE;
    const int offB = wb.selB * WPANEL_ELEMS + wb.tb0 * WTILE;
    const int lane_off4 = (lane >> 4) * WLDP + (lane & 15) * 4;      // a lane reads 4 consecutive columns per fragment read

    const int64_t rbeg = (int64_t)slab * a.rows_per_slab;
    const int64_t rend = min(rbeg + a.rows_per_slab, a.n);
    const int64_t nrows = rend > rbeg ? rend - rbeg : 0;
    const int nchunks = (int)((nrows + WKC - 1) / WKC);

    wacc_t acc[WMR][WNR];
#pragma unroll
    for (int i = 0; i < WMR; ++i)
#pragma unroll
        for (int j = 0; j < WNR; ++j) acc[i][j] = wacc_t{0, 0, 0, 0};

    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    const unsigned xbytes = nrows > 0 ? (unsigned)(((nrows - 1) * a.ldx + a.p) * (int64_t)sizeof(float)) : 0u;
    __amdgpu_buffer_rsrc_t rsrcX = __builtin_amdgcn_make_buffer_rsrc((void*)(a.X + rbeg * a.ldx), 0, (int)xbytes, 0x00020000);
    __amdgpu_buffer_rsrc_t rsrcW = __builtin_amdgcn_make_buffer_rsrc((void*)(HASW ? a.w + rbeg : a.X), 0,
                                                                     HASW ? (int)(nrows * sizeof(float)) : 0, 0x00020000);
    // lanes whose 4 columns lie past p never write: zero all stages once
    for (int e = tid; e < WSTAGES * WBUF_ELEMS; e += WTHREADS) lds[e] = 0.f;
    const bool inA = panA * WPANEL + lane * 4 + 3 < a.p;               // p % 4 == 0: a lane is all in or all out
    const bool inB = panB * WPANEL + lane * 4 + 3 < a.p;
    const int lane_boff = lane * 16;

    // every wave issues exactly WKC/4 row DMAs per chunk (WKC/8 rows x 2 panels; wave 0 one more for w)
    constexpr int RPW = WKC / WWAVES;                  // rows per wave per chunk
    constexpr int LPC = 2 * RPW;                       // loads per wave per chunk
    auto stage_dma = [&](int chunk, int stage) {
        float* base = lds + stage * WBUF_ELEMS;
#pragma unroll
        for (int r2 = 0; r2 < RPW; ++r2) {
            const int row = wave * RPW + r2;
            const int64_t rowoff = ((int64_t)chunk * WKC + row) * a.ldx;
            const int soffA = (int)((rowoff + panA * WPANEL) * (int64_t)sizeof(float));
            const int soffB = (int)((rowoff + panB * WPANEL) * (int64_t)sizeof(float));
            // masked-off lanes: voffset beyond the buffer -> the DMA returns zeros for them
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcX, (lds_ptr_t)(base + row * WLDP), 16,
                                                     inA ? lane_boff : 0x7ffffff0, soffA, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcX, (lds_ptr_t)(base + WPANEL_ELEMS + row * WLDP), 16,
                                                     inB ? lane_boff : 0x7ffffff0, soffB, 0, 0);
        }
        if (HASW && wave == 0 && lane < WKC / 4)       // exec-masked: the other lanes must not write past the 16 floats
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcW, (lds_ptr_t)(base + 2 * WPANEL_ELEMS), 16, lane * 16,
                                                     chunk * WKC * (int)sizeof(float), 0, 0);
    };
    // wait until only the WAHEAD-1 newest chunks' DMAs of this wave may still be in flight
    auto wait_prev = [&]() {
        if (HASW && wave == 0) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((WAHEAD - 1) * (LPC + 1)) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" :: "n"((WAHEAD - 1) * LPC) : "memory");
    };

    __syncthreads();                                   // zero fill done before the first DMA lands
#pragma unroll
    for (int c0 = 0; c0 < WAHEAD; ++c0)
        if (c0 < nchunks) stage_dma(c0, c0);

    for (int c = 0; c < nchunks; ++c) {
        if (c + WAHEAD - 1 < nchunks) wait_prev();     // chunk c landed (the newer ones may be in flight)
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // raw s_barrier: __syncthreads() would add a fence that drains vmcnt to 0, i.e. wait for chunk c+1 as well
        if (!DLSA_DBG_WRONG(a.dbg, 16)) asm volatile("s_barrier" ::: "memory");   // chunk c landed for every wave; stage (c-1)%S is free again
        if (c + WAHEAD < nchunks && !DLSA_DBG_WRONG(a.dbg, 1)) stage_dma(c + WAHEAD, (c + WAHEAD) % WSTAGES);
        if (active) {
            const float* base = lds + (c % WSTAGES) * WBUF_ELEMS;
            // fragments of k-step ks+1 are fetched while the 32 MFMAs of k-step ks issue (register double buffer)
            wacc_t a4[2], b4[2][2];
            float wv[2];
            auto fetch = [&](int ks, int slot) {
                const float* kb = base + ks * 4 * WLDP + lane_off4;
                a4[slot] = *reinterpret_cast<const wacc_t*>(kb + offA);
                b4[slot][0] = *reinterpret_cast<const wacc_t*>(kb + offB);
                b4[slot][1] = *reinterpret_cast<const wacc_t*>(kb + offB + 64);
                if (HASW) wv[slot] = base[2 * WPANEL_ELEMS + ks * 4 + (lane >> 4)];
            };
            fetch(0, 0);
#pragma unroll
            for (int ks = 0; ks < WKC / 4; ++ks) {
                const int cur = ks & 1;
                if (ks + 1 < WKC / 4) fetch(ks + 1, cur ^ 1);
                __builtin_amdgcn_sched_barrier(0);     // keep the prefetch ahead of the MFMA block (the scheduler sinks it otherwise)
                if (HASW) a4[cur] *= wv[cur];
#pragma unroll
                for (int i = 0; i < WMR; ++i)
#pragma unroll
                    for (int j = 0; j < WNR; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[cur][i], b4[cur][j >> 2][j & 3], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }

    // epilogue: the whole 64 x 128 block -> this slab's partial buffer (leading dimension PP = panels x 256, so no
    // bounds checks; entries below the diagonal or past p are never read by the reduce kernel).
    // acc[i][4g+e][r] is H[row0 + 4*(4*(lane>>4) + r) + i][col0 + 64g + 4*(lane&15) + e]  (fp32 C/D layout:
    // tile row = 4*(lane>>4) + r, tile column = lane&15; tile e of a group holds the columns 4m+e)
    float* __restrict__ P = a.partial + (int64_t)slab * a.PP * a.PP;
    if (active) {
        const int row0 = (wb.selA ? panB : panA) * WPANEL + wb.ta0 * WTILE;
        const int col0 = (wb.selB ? panB : panA) * WPANEL + wb.tb0 * WTILE;
#pragma unroll
        for (int i = 0; i < WMR; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float* dst = P + (int64_t)(row0 + 4 * (4 * (lane >> 4) + r) + i) * a.PP + col0 + 4 * (lane & 15);
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    wacc_t v = {acc[i][4 * g + 0][r], acc[i][4 * g + 1][r], acc[i][4 * g + 2][r], acc[i][4 * g + 3][r]};
                    *reinterpret_cast<wacc_t*>(dst + 64 * g) = v;
                }
            }
    }
}

template <typename T>
void gram_reduce_launch(const T* partial, int nslab, int PP, int p, T* H, int64_t ldh, int accumulate, hipStream_t stream);   // gram.hip
void gram_reduce_launch_f32_to_f64(const float* partial, int nslab, int PP, int p, double* H, int64_t ldh, int accumulate, hipStream_t stream);

// -------------------------------------------------------------------------------------------------
// host: plan
// -------------------------------------------------------------------------------------------------
static void build_wide_items(int p, std::vector<WideItem>& items) {
    const int ntile = (p + WTILE - 1) / WTILE;
    const int npan = (p + WPANEL - 1) / WPANEL;
    auto block_mask = [&](int pr, int ta0, int pc, int tb0) {
        unsigned m = 0;
        for (int i = 0; i < WMR; ++i)
            for (int j = 0; j < WNR; ++j) {
                const int ti = pr * 16 + ta0 + i, tj = pc * 16 + tb0 + j;
                if (ti < ntile && tj < ntile && ti <= tj) m |= 1u << (i * WNR + j);
            }
        return m;
    };
    items.clear();
    // off-diagonal panel pairs: 8 blocks = one workgroup
    for (int x = 0; x < npan; ++x)
        for (int y = x + 1; y < npan; ++y) {
            WideItem g{};
            g.panA = x; g.panB = y;
            for (int wv = 0; wv < WWAVES; ++wv) {
                WideBlock& b = g.wb[wv];
                b.selA = 0; b.ta0 = (unsigned char)(4 * (wv >> 1));
                b.selB = 1; b.tb0 = (unsigned char)(8 * (wv & 1));
                b.mask = block_mask(x, b.ta0, y, b.tb0);
            }
            items.push_back(g);
        }
    // diagonal panels: blocks with at least one tile on/above the diagonal, packed 8 to a workgroup,
    // at most two panels per workgroup
    struct DB { int pan, ta0, tb0; unsigned mask; };
    std::vector<DB> pool;
    for (int d = 0; d < npan; ++d)
        for (int r = 0; r < 4; ++r)
            for (int c = 0; c < 2; ++c) {
                const unsigned m = block_mask(d, 4 * r, d, 8 * c);
                if (m) pool.push_back(DB{d, 4 * r, 8 * c, m});
            }
    size_t pos = 0;
    while (pos < pool.size()) {
        WideItem g{};
        g.panA = pool[pos].pan; g.panB = g.panA;
        int wv = 0;
        while (pos < pool.size() && wv < WWAVES) {
            const DB& d = pool[pos];
            if (d.pan != g.panA && d.pan != g.panB) {
                if (g.panB != g.panA) break;            // a third panel: next workgroup
                g.panB = d.pan;
            }
            WideBlock& b = g.wb[wv++];
            b.selA = b.selB = (d.pan == g.panA) ? 0 : 1;
            b.ta0 = (unsigned char)d.ta0; b.tb0 = (unsigned char)d.tb0; b.mask = d.mask;
            ++pos;
        }
        items.push_back(g);
    }
}

struct WidePlan { int nitems = 0, PP = 0; WideItem* d_items = nullptr; };
static std::mutex g_wide_mu;
static std::map<std::pair<int, int>, WidePlan> g_wide_plans;

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
    pl.PP = (p + WPANEL - 1) / WPANEL * WPANEL;
    DLSA_HIP_CHECK(hipMalloc((void**)&pl.d_items, items.size() * sizeof(WideItem)));
    DLSA_HIP_CHECK(hipMemcpy(pl.d_items, items.data(), items.size() * sizeof(WideItem), hipMemcpyHostToDevice));
    g_wide_plans[key] = pl;
    out = pl;
    return DLSA_OK;
}

// One workgroup per CU is resident.  Choose the number of slabs (a multiple of 8 for the XCD mapping) so that
// nitems*nslab fills whole rounds of 256 workgroups, slabs stay >= 2048 rows, and a slab's bytes fit the
// 32-bit buffer descriptor.
static void choose_wide_slabs(int64_t n, int p, int nitems, int& nslab, int64_t& rows_per_slab) {
    const int64_t max_rows = std::max<int64_t>(WKC, (int64_t)(1.9e9 / ((double)p * sizeof(float))) / WKC * WKC);
    int64_t ns_min = std::max<int64_t>(1, (n + max_rows - 1) / max_rows);
    ns_min = (ns_min + kNumXCD - 1) / kNumXCD * kNumXCD;
    const int64_t ns_max = std::max<int64_t>(ns_min, std::min<int64_t>(1024, n / 2048 / kNumXCD * kNumXCD));
    int64_t best = ns_min;
    double best_eff = -1.0;
    for (int64_t ns = ns_min; ns <= ns_max; ns += kNumXCD) {
        const int64_t wg = ns * nitems;
        const double eff = (double)wg / (double)((wg + kNumCU - 1) / kNumCU * kNumCU);
        if (eff > best_eff + 0.02) { best_eff = eff; best = ns; }       // prefer fewer, larger slabs
        if (eff >= 0.995) break;
    }
    rows_per_slab = ((n + best - 1) / best + WKC - 1) / WKC * WKC;
    if (rows_per_slab < WKC) rows_per_slab = WKC;
    nslab = (int)best;                                                  // trailing slabs may be empty: they write zeros
}

static int wide_nitems(int p) {       // closed form of build_wide_items' count is not worth it: build and count
    std::vector<WideItem> items;
    build_wide_items(p, items);
    return (int)items.size();
}

bool gram_wide_f32_shape_ok(int64_t n, int p) {
    return p >= 768 && (p % 4 == 0) && n >= 16384;
}

bool gram_wide_f32_eligible(const float* X, int64_t ldx, const float* w, int64_t n, int p) {
    if (getenv("DLSA_GRAM_NOWIDE")) return false;
    if (!gram_wide_f32_shape_ok(n, p) || (ldx % 4) || ((uintptr_t)X & 15) || (w && ((uintptr_t)w & 15))) return false;
    int nslab; int64_t rps;
    choose_wide_slabs(n, p, wide_nitems(p), nslab, rps);
    return (double)rps * (double)ldx * sizeof(float) < 2.1e9;     // a slab must fit the 32-bit buffer descriptor
}

size_t gram_wide_f32_ws_bytes(int64_t n, int p) {
    std::vector<WideItem> items;
    build_wide_items(p, items);
    int nslab; int64_t rps;
    choose_wide_slabs(n, p, (int)items.size(), nslab, rps);
    const size_t PP = (size_t)(p + WPANEL - 1) / WPANEL * WPANEL;
    return align_up((size_t)nslab * PP * PP * sizeof(float), 256);
}

int gram_wide_f32(const float* X, int64_t ldx, const float* w, int64_t n, int p, float* H, int64_t ldh,
                  int accumulate, void* ws, size_t ws_bytes, hipStream_t stream, double* H64) {
    WidePlan pl;
    int rc = get_wide_plan(p, pl);
    if (rc) return rc;
    int nslab; int64_t rps;
    choose_wide_slabs(n, p, pl.nitems, nslab, rps);
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

// host-only self check of the wide plan (CPU test-suite): every tile on/above the diagonal exactly once
int gram_wide_plan_check(int p, int* nitems, int* nslots, int* ntiles) {
    std::vector<WideItem> items;
    build_wide_items(p, items);
    const int ntile = (p + WTILE - 1) / WTILE;
    std::vector<int> seen((size_t)ntile * ntile, 0);
    int count = 0, slots = 0;
    for (auto& g : items) {
        if (g.panA > g.panB) return -5;
        for (int wv = 0; wv < WWAVES; ++wv) {
            const WideBlock& b = g.wb[wv];
            if (!b.mask) continue;
            slots += WMR * WNR;
            for (int i = 0; i < WMR; ++i)
                for (int j = 0; j < WNR; ++j) {
                    if (!((b.mask >> (i * WNR + j)) & 1)) continue;
                    const int ti = (b.selA ? g.panB : g.panA) * 16 + b.ta0 + i;
                    const int tj = (b.selB ? g.panB : g.panA) * 16 + b.tb0 + j;
                    if (ti > tj || tj >= ntile) return -1;
                    if (seen[(size_t)ti * ntile + tj]++) return -2;
                    ++count;
                }
        }
    }
    if (count != ntile * (ntile + 1) / 2) return -3;
    if (nitems) *nitems = (int)items.size();
    if (nslots) *nslots = slots;
    if (ntiles) *ntiles = count;
    return 0;
}

}  // namespace dlsa

extern "C" int dlsa_gram_wide_plan_check(int p, int* nitems, int* nslots, int* ntiles) {
    return dlsa::gram_wide_plan_check(p, nitems, nslots, ntiles);
}
