// Tall-skinny weighted Gram  H = X' diag(w) X  on gfx950 MFMA  (reference call site:
// dlsa/models.py:130  Sig_inv = x_train.T.dot(np.multiply((prob*(1-prob))[:,None], x_train))).
//
// Shape of the problem: M = N = p (<= a few thousand), K = n rows (10^7..10^8).  It is a
// split-K GEMM whose output is tiny and symmetric, so
//   * only tiles on or above the diagonal of H are stored, the reduce kernel mirrors them;
//   * the p columns are cut into 128-column PANELS (8 tiles of 16).  A WAVE owns a 4x4 block of
//     16x16 output tiles (64x64 of H, 128 accumulator VGPRs in fp64): per 4-row k-step it reads
//     4 A fragments + 4 B fragments from LDS and issues 16 MFMAs -- the register blocking is
//     what lets v_mfma_f64_16x16x4_f64 run at its 64-cycle issue rate (measured: 70 TF with
//     4x4 blocking from LDS vs 56 TF with one fragment pair per MFMA; bench/ubench_gram_inner.hip);
//   * a WORKGROUP is 4 such waves whose blocks need at most two panels (an off-diagonal panel
//     pair = 4 blocks; the three blocks of a diagonal panel ride together with a neighbour's),
//     x one SLAB of rows.  It streams the slab's two panels through LDS in 16-row chunks
//     (double buffered, 74 KB -> two workgroups per CU so one's staging hides under the other's
//     MFMAs) and finally writes its tiles to a per-slab partial buffer;
//   * workgroups that share a slab are dispatched to the same XCD back to back, so the slab's
//     panels come from HBM once and are re-read from that XCD's L2;
//   * a second kernel sums the slab partials in a fixed order (deterministic) into H.
//
// MFMA operand layout (v_mfma_f64_16x16x4_f64): A[i=lane&15][k=lane>>4], B[k=lane>>4][j=lane&15],
// C/D reg r of lane l = C[4r + (l>>4)][l&15]  (the fp64 C/D map differs from every other dtype).
// With A[i][k] = X[r0+k][ca+i] and B[k][j] = w[r0+k] X[r0+k][cb+j] both fragments are the SAME
// LDS read pattern from row 4ks + (l>>4) of the chunk (row pitch 144 elements).  In the blocked mode the
// MFMA tiles of a 64-column block are INTERLEAVED: tile e of 32-column group g holds the columns
// 32g + 2m + e (m = lane&15), so ONE 16-byte LDS read of the natural row layout feeds two tiles:
// 4 ds_read_b128 + the w read per 16 MFMAs instead of 8 ds_read_b64.  On this chip every LDS / VALU
// instruction issued next to fp64 MFMAs costs MFMA cycles, so halving them matters;
// the next k-step's fragments are also fetched before the current MFMA block (register double buffer).
// The epilogue undoes the interleave.  fp32 uses v_mfma_f32_16x16x4_f32 with the same scheme.
#include "common.h"
#include <vector>
#include <algorithm>
#include <mutex>
#include <map>
#include <stdlib.h>

namespace dlsa {

constexpr int TILE = 16;
constexpr int PANEL = 128;        // columns per panel = 8 tiles
#ifndef DLSA_GRAM_KC
#define DLSA_GRAM_KC 16
#endif
#ifndef DLSA_GRAM_PREFETCH
#define DLSA_GRAM_PREFETCH 1
#endif
#ifndef DLSA_GRAM_OCC
#define DLSA_GRAM_OCC 2
#endif
constexpr int KC = DLSA_GRAM_KC;  // rows per staged chunk = KC/4 MFMA k-steps
// LDS row pitch in elements, chosen per fragment-read instruction (MI355X_MICROARCH.md, LDS lane groups):
//   list mode, one element per read: rows of a lane group must sit 16 elements apart modulo 32 -> 144;
//   blocked fp64, ds_read_b128 (groups of 16 lanes mixing two rows): 256-byte multiples -> 128, no padding;
//   blocked fp32, ds_read_b64 (groups of 32 lanes = two rows): 128 bytes apart modulo 256 -> 160.
constexpr int gram_ldp(int elem_bytes, bool list) { return list ? 144 : (elem_bytes == 8 ? 128 : 160); }
constexpr int GRAM_WAVES = 4;
constexpr int GRAM_THREADS = 64 * GRAM_WAVES;
constexpr int MR = 4, NR = 4;     // tiles per wave block

struct WaveBlock {
    unsigned char a[MR];          // sel<<3 | local tile index (0..7) of the A (row) tiles
    unsigned char b[NR];          // same for the B (column) tiles
    unsigned short mask;          // bit i*NR+j: tile (a[i], b[j]) is stored (on/above the diagonal, inside p)
    unsigned short tri;           // 1: diagonal block (a == b): only tiles j >= i are computed
};
constexpr int LIST_CAP = 12;      // list mode: at most 12 tiles per wave (16 makes hipcc spill in fp64)
struct GramItem {
    int panA, panB;               // panel indices (panB == panA: single-panel item)
    WaveBlock wb[GRAM_WAVES];     // blocked mode: one 4x4 tile block per wave
    // list mode (ragged / small p): an explicit, balanced tile list per wave, padded with dummies;
    // code = dummy<<8 | selA<<7 | tiA<<4 | selB<<3 | tjB   (ti, tj local 0..7)
    unsigned short list[GRAM_WAVES][LIST_CAP];
    int lnt[GRAM_WAVES];          // real tiles per wave
};

template <typename T>
struct GramArgs {
    const T* X;
    const T* w;
    T* partial;          // [nslab][PP][PP]
    const GramItem* items;
    int64_t ldx;
    int64_t n;
    int64_t rows_per_slab;
    int p;
    int PP;              // padded dimension: ntile*16 rounded up to a multiple of 64
    int nitems;
    int nslab;
    int xcd_map;         // 1: XCD-aware block->(item,slab) mapping (needs nslab % 8 == 0)
    int dbg;             // DLSA_GRAM_DBG (profiling experiments only): 1 = no global loads after chunk 0, 2 = no XCD map, 4 = no LDS-DMA, 16 = no barrier (timing only), 32 = never use the list plan
};

template <typename T> struct Mfma;
template <> struct Mfma<double> {
    typedef double acc_t __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ acc_t run(double a, double b, acc_t c) {
        return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ int crow(int lane, int r) { return 4 * r + (lane >> 4); }
};
template <> struct Mfma<float> {
    typedef float acc_t __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ acc_t run(float a, float b, acc_t c) {
        return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ int crow(int lane, int r) { return 4 * (lane >> 4) + r; }
};

template <typename T> struct Vec2;
template <> struct Vec2<double> { typedef double2 type; };
template <> struct Vec2<float> { typedef float2 type; };

// One chunk of one panel is KC rows x 128 columns.  Each thread stages 2 adjacent columns
// of one row per pass; a wave covers a whole 128-column row (1 KiB coalesced in fp64).
template <typename T, bool VEC>
__device__ __forceinline__ typename Vec2<T>::type load_pair(const T* __restrict__ X, int64_t ldx,
                                                            int64_t grow, int64_t rend, int gcol, int p) {
    typename Vec2<T>::type v;
    v.x = T(0); v.y = T(0);
    if (grow < rend) {
        const T* src = X + grow * ldx + gcol;
        if (gcol + 1 < p) {
            if (VEC) {
                v = *reinterpret_cast<const typename Vec2<T>::type*>(src);
            } else {
                v.x = src[0]; v.y = src[1];
            }
        } else if (gcol < p) {
            v.x = src[0];
        }
    }
    return v;
}

// MODE 0: scalar global loads -> registers -> LDS (any alignment);  MODE 1: 16-byte loads -> registers -> LDS;
// MODE 2 (fp64): direct global->LDS DMA (buffer_load_dwordx4 ... lds): no staging VGPRs, no ds_write pass,
// addresses are SGPR offsets, rows past the slab end read as zeros through the buffer descriptor.
// (A fifth "loader" wave issuing all the DMA was tried and measured slower: 109 vs 101.7 ms.)
// NT == 0: blocked mode (one 4x4 tile block per wave, 8 fragment reads per 16 MFMAs -- the fast path for
// p that fills the blocks).  NT > 0: list mode -- every wave walks its own list of NT tiles with one fragment
// pair per MFMA (~20 % slower per tile, bench/ubench_gram_inner.hip mode 5) but no wasted tile slot and
// perfectly balanced waves: the better plan for small or ragged p (p=100: 7 tiles per wave instead of 16 slots).
template <typename T, bool HASW, int MODE, int NT>
__global__ __launch_bounds__(GRAM_THREADS, DLSA_GRAM_OCC) void gram_kernel(GramArgs<T> a) {
    constexpr bool VEC = MODE >= 1;
    constexpr bool DMA = MODE == 2;
    constexpr bool LIST = NT > 0;
    constexpr int NTL = LIST ? NT : 1;
    typedef typename Mfma<T>::acc_t acc_t;
    typedef typename Vec2<T>::type vec2_t;
    constexpr int PASSES = KC / GRAM_WAVES;              // staging passes per panel
    constexpr int LDP = gram_ldp((int)sizeof(T), LIST);
    constexpr int PANEL_ELEMS = KC * LDP;
    constexpr int BUF_ELEMS = 2 * PANEL_ELEMS + KC;      // two panels + the w chunk
    __shared__ __attribute__((aligned(16))) T lds[2 * BUF_ELEMS];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // workgroup -> (item, slab)
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
    const GramItem* __restrict__ it = a.items + item_id;
    const int panA = it->panA, panB = it->panB;
    const int npanels = (panA == panB) ? 1 : 2;
    const WaveBlock wb = it->wb[wave];
    const bool active = LIST ? (it->lnt[wave] > 0) : (wb.mask != 0);   // wave-uniform
    const bool tri = wb.tri != 0;                        // wave-uniform
    int loffA[NTL], loffB[NTL];
    if constexpr (LIST) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int code = it->list[wave][t];
            loffA[t] = ((code >> 7) & 1) * (KC * LDP) + ((code >> 4) & 7) * TILE;
            loffB[t] = ((code >> 3) & 1) * (KC * LDP) + (code & 7) * TILE;
        }
    }

    // LDS element offsets of the block's 64-column A and B origins (wave-uniform); a[0] / b[0] are the
    // first tiles of 4-aligned tile groups, i.e. column 0 or 64 of their panel
    const int offA0 = ((wb.a[0] >> 3) & 1) * PANEL_ELEMS + (wb.a[0] & 7) * TILE;
    const int offB0 = ((wb.b[0] >> 3) & 1) * PANEL_ELEMS + (wb.b[0] & 7) * TILE;
    const int lane_off = (lane >> 4) * LDP + (lane & 15);          // list mode: one element per fragment read
    const int lane_off2 = (lane >> 4) * LDP + (lane & 15) * 2;     // blocked mode: two adjacent columns per read

    const int64_t rbeg = (int64_t)slab * a.rows_per_slab;
    const int64_t rend = min(rbeg + a.rows_per_slab, a.n);
    const int nchunks = rbeg < rend ? (int)((rend - rbeg + KC - 1) / KC) : 0;

    acc_t acc[MR][NR];
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int j = 0; j < NR; ++j) acc[i][j] = acc_t{0, 0, 0, 0};
    acc_t accl[NTL];
#pragma unroll
    for (int t = 0; t < NTL; ++t) accl[t] = acc_t{0, 0, 0, 0};

    vec2_t st[2][PASSES];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int ps = 0; ps < PASSES; ++ps) { st[s][ps].x = T(0); st[s][ps].y = T(0); }
    T wreg = T(0);
    const int srow = wave;          // + GRAM_WAVES*pass
    const int scol = lane * 2;
    // per-thread column class of each panel: 2 = both columns inside p, 1 = boundary, 0 = outside
    int colk[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int gcol = (s == 0 ? panA : panB) * PANEL + scol;
        colk[s] = (s < npanels) ? ((gcol + 1 < a.p) ? 2 : (gcol < a.p ? 1 : 0)) : 0;
    }
    const int lane_boff = scol * (int)sizeof(T);

    // Steady state: the whole chunk is inside the slab, so a load is "wave-uniform row pointer
    // (SALU) + constant per-lane byte offset" with no per-element bounds checks; the generic
    // load_pair path only runs for the slab's last chunk, unaligned inputs and boundary columns.
    auto stage_load = [&](int chunk) {
        const int64_t r0 = rbeg + (int64_t)chunk * KC;
        const bool fast = VEC && (r0 + KC <= rend);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int pan = (s == 0 ? panA : panB);
            if (fast && colk[s] == 2) {
#pragma unroll
                for (int ps = 0; ps < PASSES; ++ps) {
                    const T* rowp = a.X + (r0 + srow + GRAM_WAVES * ps) * a.ldx + pan * PANEL;   // wave-uniform
                    st[s][ps] = *reinterpret_cast<const vec2_t*>(reinterpret_cast<const char*>(rowp) + lane_boff);
                }
            } else if (colk[s] != 0) {
#pragma unroll
                for (int ps = 0; ps < PASSES; ++ps)
                    st[s][ps] = load_pair<T, VEC>(a.X, a.ldx, r0 + srow + GRAM_WAVES * ps, rend, pan * PANEL + scol, a.p);
            }
        }
        if (HASW && tid < KC) {
            const int64_t r = r0 + tid;
            wreg = (r < rend) ? a.w[r] : T(0);
        }
    };
    auto stage_write = [&](int buf) {
        T* base = lds + buf * BUF_ELEMS;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            if (s < npanels) {
#pragma unroll
                for (int ps = 0; ps < PASSES; ++ps)
                    *reinterpret_cast<vec2_t*>(base + s * PANEL_ELEMS + (srow + GRAM_WAVES * ps) * LDP + scol) = st[s][ps];
            }
        }
        if (HASW && tid < KC) base[2 * PANEL_ELEMS + tid] = wreg;
    };

    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    __amdgpu_buffer_rsrc_t rsrcX, rsrcW;
    if constexpr (DMA) {
        const int64_t nrows = rend > rbeg ? rend - rbeg : 0;
        const unsigned xbytes = nrows > 0 ? (unsigned)(((nrows - 1) * a.ldx + a.p) * (int64_t)sizeof(T)) : 0u;
        rsrcX = __builtin_amdgcn_make_buffer_rsrc((void*)(a.X + rbeg * a.ldx), 0, (int)xbytes, 0x00020000);
        rsrcW = __builtin_amdgcn_make_buffer_rsrc((void*)(HASW ? a.w + rbeg : a.X), 0, HASW ? (int)(nrows * sizeof(T)) : 0, 0x00020000);
        // columns past p are never written by the DMA (lanes masked): zero the buffers once
        for (int e = tid; e < 2 * BUF_ELEMS; e += GRAM_THREADS) lds[e] = T(0);
    }
    auto stage_dma = [&](int chunk, int buf) {
        if constexpr (DMA) {
            T* base = lds + buf * BUF_ELEMS;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                if (colk[s] == 2) {                      // per-lane: both columns inside p
                    const int pan = (s == 0 ? panA : panB);
#pragma unroll
                    for (int ps = 0; ps < PASSES; ++ps) {
                        const int row = srow + GRAM_WAVES * ps;
                        const int soff = (int)((((int64_t)chunk * KC + row) * a.ldx + pan * PANEL) * (int64_t)sizeof(T));
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcX, (lds_ptr_t)(base + s * PANEL_ELEMS + row * LDP), 16,
                                                                 lane_boff, soff, 0, 0);
                    }
                }
            }
            if (HASW && wave == 0 && lane < KC / 2)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcW, (lds_ptr_t)(base + 2 * PANEL_ELEMS), 16, lane * 16,
                                                         chunk * KC * (int)sizeof(T), 0, 0);
        }
    };

    if constexpr (DMA) {
        __syncthreads();                                 // zero fill done before the first DMA lands
        if (nchunks > 0) stage_dma(0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        if (nchunks > 0) {
            stage_load(0);
            stage_write(0);
        }
    }
    __syncthreads();

    for (int c = 0; c < nchunks; ++c) {
        if (c + 1 < nchunks && !DLSA_DBG_WRONG(a.dbg, 1)) {
            if constexpr (DMA) stage_dma(c + 1, (c + 1) & 1);
            else stage_load(c + 1);
        }
        if (active) {
            const T* base = lds + (c & 1) * BUF_ELEMS;
            if constexpr (LIST) {
#pragma unroll
                for (int ks = 0; ks < KC / 4; ++ks) {
                    const T* kb = base + ks * 4 * LDP + lane_off;
                    T wv = T(1);
                    if (HASW) wv = base[2 * PANEL_ELEMS + ks * 4 + (lane >> 4)];
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        const T av = kb[loffA[t]];
                        T bv = kb[loffB[t]];
                        if (HASW) bv *= wv;
                        accl[t] = Mfma<T>::run(av, bv, accl[t]);
                    }
                }
            } else {
                vec2_t a2[2][2], b2[2][2];
                T wv[2];
                auto fetch = [&](int ks, int slot) {
                    const T* kb = base + ks * 4 * LDP + lane_off2;
                    // written as scalar loads on purpose: hipcc merges each pair into one 16-byte ds_read, whereas an
                    // explicit vector load here makes it put "s_waitcnt vmcnt(0)" in front of the first read, i.e.
                    // wait for the LDS-DMA of the NEXT chunk that was issued just above (serialising load and compute)
                    a2[slot][0].x = kb[offA0]; a2[slot][0].y = kb[offA0 + 1];
                    a2[slot][1].x = kb[offA0 + 32]; a2[slot][1].y = kb[offA0 + 33];
                    b2[slot][0].x = kb[offB0]; b2[slot][0].y = kb[offB0 + 1];
                    b2[slot][1].x = kb[offB0 + 32]; b2[slot][1].y = kb[offB0 + 33];
                    if (HASW) wv[slot] = base[2 * PANEL_ELEMS + ks * 4 + (lane >> 4)];
                };
#if DLSA_GRAM_PREFETCH
                fetch(0, 0);
#endif
#pragma unroll
                for (int ks = 0; ks < KC / 4; ++ks) {
                    const int cur = ks & 1;
#if DLSA_GRAM_PREFETCH
                    if (ks + 1 < KC / 4) fetch(ks + 1, cur ^ 1);
                    __builtin_amdgcn_sched_barrier(0);      // keep the prefetch ahead of the MFMA block
#else
                    fetch(ks, cur);
#endif
                    T av[MR], bv[NR];
#pragma unroll
                    for (int g = 0; g < 2; ++g) {
                        av[2 * g] = a2[cur][g].x; av[2 * g + 1] = a2[cur][g].y;
                        bv[2 * g] = b2[cur][g].x; bv[2 * g + 1] = b2[cur][g].y;
                    }
                    if (HASW) {
#pragma unroll
                        for (int j = 0; j < NR; ++j) bv[j] *= wv[cur];
                    }
#pragma unroll
                    for (int i = 0; i < MR; ++i)
#pragma unroll
                        for (int j = i; j < NR; ++j) acc[i][j] = Mfma<T>::run(av[i], bv[j], acc[i][j]);
                    if (!tri) {     // wave-uniform: a diagonal block skips its 6 below-diagonal tiles
#pragma unroll
                        for (int i = 1; i < MR; ++i)
#pragma unroll
                            for (int j = 0; j < i; ++j) acc[i][j] = Mfma<T>::run(av[i], bv[j], acc[i][j]);
                    }
#if DLSA_GRAM_PREFETCH
                    __builtin_amdgcn_sched_barrier(0);
#endif
                }
            }
        }
        if constexpr (DMA) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // chunk c+1 has landed in the other buffer
        } else {
            if (c + 1 < nchunks) stage_write((c + 1) & 1);
        }
        if (!DLSA_DBG_WRONG(a.dbg, 16)) __syncthreads();                    // dbg 16: timing experiment only (wrong results)
    }

    // epilogue: stored tiles -> this slab's partial buffer
    T* __restrict__ P = a.partial + (int64_t)slab * a.PP * a.PP;
    if constexpr (LIST) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int code = it->list[wave][t];
            if (!((code >> 8) & 1)) {
                const int r0 = ((((code >> 7) & 1) ? panB : panA) * 8 + ((code >> 4) & 7)) * TILE;
                const int c0 = ((((code >> 3) & 1) ? panB : panA) * 8 + (code & 7)) * TILE;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    P[(int64_t)(r0 + Mfma<T>::crow(lane, r)) * a.PP + c0 + (lane & 15)] = accl[t][r];
            }
        }
    } else if (active) {
        // Interleaved tiles: acc[i][j][r] = H[row0 + 32(i>>1) + 2*tr + (i&1)][col0 + 32(j>>1) + 2*tc + (j&1)] with
        // tr = crow(lane, r), tc = lane&15.  PP is a multiple of 64, so the whole 64x64 block lies inside the
        // partial buffer; entries below the diagonal or past p are never read by the reduce kernel.
        const int row0 = ((((wb.a[0] >> 3) & 1) ? panB : panA) * 8 + (wb.a[0] & 7)) * TILE;
        const int col0 = ((((wb.b[0] >> 3) & 1) ? panB : panA) * 8 + (wb.b[0] & 7)) * TILE;
        const int tc = lane & 15;
        if (!tri) {
#pragma unroll
            for (int i = 0; i < MR; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int R = row0 + 32 * (i >> 1) + 2 * Mfma<T>::crow(lane, r) + (i & 1);
                    T* dst = P + (int64_t)R * a.PP + col0 + 2 * tc;
#pragma unroll
                    for (int g = 0; g < 2; ++g) {
                        vec2_t v;
                        v.x = acc[i][2 * g][r]; v.y = acc[i][2 * g + 1][r];
                        *reinterpret_cast<vec2_t*>(dst + 32 * g) = v;
                    }
                }
        } else {
            // diagonal block: only tiles j >= i were computed.  Inside a diagonal 32x32 group the skipped tile
            // (2g+1, 2g) is the transpose of tile (2g, 2g+1), whose lower half is therefore stored mirrored;
            // every entry on/above the diagonal is written exactly once (deterministic).
#pragma unroll
            for (int i = 0; i < MR; ++i)
#pragma unroll
                for (int j = i; j < NR; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int tr = Mfma<T>::crow(lane, r);
                        const int R = row0 + 32 * (i >> 1) + 2 * tr + (i & 1);
                        const int C = col0 + 32 * (j >> 1) + 2 * tc + (j & 1);
                        const T v = acc[i][j][r];
                        if ((i >> 1) != (j >> 1)) P[(int64_t)R * a.PP + C] = v;
                        else if (tr <= tc) P[(int64_t)R * a.PP + C] = v;
                        else if (i != j) P[(int64_t)C * a.PP + R] = v;
                    }
        }
    }
}

// H[i][j] = H[j][i] = (accumulate ? H[i][j] : 0) + sum_s partial[s][i][j]  for i <= j  (fixed order).  Only the upper
// triangle is read -- coalesced along j -- and each sum is written to both triangles (the mirrored write is the
// uncoalesced one, but it is p^2/2 elements once instead of nslab strided reads per element).
// H (+)= sum over slabs of the partial upper triangles, mirrored.  One element = 16 threads (a quarter wave's rows of the
// block): thread kg sums slabs kg, kg + 16, ... in two chains, the 16 sums meet in LDS and are added in a FIXED order
// (bit-reproducible).  With one thread per element the 512 slabs of the narrow kernel were 128 dependent loads deep: 62 us
// at p = 100, 3 % of the whole Gram of 1e7 rows and a third of a 1e6-row partition's.
constexpr int RED_J = 16, RED_K = 16;
// TO = the output / summation type: the slab partials' own type, or double for fp32 partials whose sum goes on into an
// fp64 accumulator (dlsa_gram_f32_acc64: a streaming map step adds chunk after chunk without fp32 cancellation across chunks)
template <typename TP, typename T>
__global__ __launch_bounds__(RED_J * RED_K) void gram_reduce_kernel(const TP* __restrict__ partial, int nslab, int PP, int p,
                                                                      T* __restrict__ H, int64_t ldh, int accumulate) {
    __shared__ T part[RED_K][RED_J + 1];
    const int jl = threadIdx.x % RED_J, kg = threadIdx.x / RED_J;
    const int j = blockIdx.x * RED_J + jl;
    const int i = blockIdx.y;
    if (blockIdx.x * RED_J + RED_J - 1 < i) return;           // the whole block lies below the diagonal
    const bool live = j < p && j >= i;
    T s0 = T(0), s1 = T(0);
    if (live) {
        const TP* src = partial + (int64_t)i * PP + j;
        const int64_t stride = (int64_t)PP * PP;
        int k = kg;
        for (; k + RED_K < nslab; k += 2 * RED_K) {
            s0 += (T)src[k * stride];
            s1 += (T)src[(k + RED_K) * stride];
        }
        if (k < nslab) s0 += (T)src[k * stride];
    }
    part[kg][jl] = s0 + s1;
    __syncthreads();
    if (kg == 0 && live) {
        T s = part[0][jl];
#pragma unroll
        for (int g = 1; g < RED_K; ++g) s += part[g][jl];
        T* dst = H + (int64_t)i * ldh + j;
        *dst = accumulate ? (*dst + s) : s;
        if (i != j) {
            T* mir = H + (int64_t)j * ldh + i;
            *mir = accumulate ? (*mir + s) : s;
        }
    }
}

template <typename T>
void gram_reduce_launch(const T* partial, int nslab, int PP, int p, T* H, int64_t ldh, int accumulate, hipStream_t stream) {
    dim3 rg((p + RED_J - 1) / RED_J, p);
    hipLaunchKernelGGL((gram_reduce_kernel<T, T>), rg, dim3(RED_J * RED_K), 0, stream, partial, nslab, PP, p, H, ldh, accumulate);
}
void gram_reduce_launch_f32_to_f64(const float* partial, int nslab, int PP, int p, double* H, int64_t ldh, int accumulate, hipStream_t stream) {
    dim3 rg((p + RED_J - 1) / RED_J, p);
    hipLaunchKernelGGL((gram_reduce_kernel<float, double>), rg, dim3(RED_J * RED_K), 0, stream, partial, nslab, PP, p, H, ldh, accumulate);
}
template void gram_reduce_launch<double>(const double*, int, int, int, double*, int64_t, int, hipStream_t);
template void gram_reduce_launch<float>(const float*, int, int, int, float*, int64_t, int, hipStream_t);

// ---------------------------------------------------------------------------------------
// Host side: wave blocks -> workgroup items, cached per p on the device.
// ---------------------------------------------------------------------------------------
struct GramPlan {
    int p = 0, ntile = 0, npan = 0, PP = 0;
    int nitems = 0;               // blocked plan
    GramItem* d_items = nullptr;
    int nt_list = 0, nitems_list = 0;   // list plan (nt_list == 0: not worth it for this p)
    GramItem* d_items_list = nullptr;
};

// A wave block in global tile coordinates: tile rows ra[0..3], tile cols cb[0..3] (-1 = unused)
struct HostBlock {
    int ra[MR], cb[NR];
    int pr, pc;      // panel of the rows / of the columns
};

static void build_items(int p, std::vector<GramItem>& items) {
    const int ntile = (p + TILE - 1) / TILE;
    // all 4x4 tile blocks that contain at least one tile on/above the diagonal
    std::vector<HostBlock> blocks;
    const int nb4 = (ntile + 3) / 4;                     // blocks of 4 tiles per dimension
    for (int bi = 0; bi < nb4; ++bi)
        for (int bj = bi; bj < nb4; ++bj) {
            HostBlock hb;
            for (int i = 0; i < MR; ++i) hb.ra[i] = (bi * 4 + i < ntile) ? bi * 4 + i : -1;
            for (int j = 0; j < NR; ++j) hb.cb[j] = (bj * 4 + j < ntile) ? bj * 4 + j : -1;
            hb.pr = (bi * 4) / 8;
            hb.pc = (bj * 4) / 8;
            blocks.push_back(hb);
        }
    // pack blocks into workgroups of GRAM_WAVES blocks touching at most two panels:
    // first the off-diagonal panel pairs (exactly 4 blocks when both panels are full), then the
    // blocks that live inside one panel, greedily appended to a workgroup that already stages it.
    struct Group { int pa, pb; std::vector<HostBlock> bl; };
    std::vector<Group> groups;
    auto find_or_new = [&](int pa, int pb) -> Group& {
        for (auto& g : groups)
            if ((int)g.bl.size() < GRAM_WAVES && g.pa == pa && g.pb == pb) return g;
        groups.push_back(Group{pa, pb, {}});
        return groups.back();
    };
    for (auto& hb : blocks)
        if (hb.pr != hb.pc) find_or_new(hb.pr, hb.pc).bl.push_back(hb);
    for (auto& hb : blocks) {
        if (hb.pr != hb.pc) continue;
        Group* best = nullptr;
        for (auto& g : groups) {
            if ((int)g.bl.size() >= GRAM_WAVES) continue;
            const bool has = (g.pa == hb.pr || g.pb == hb.pr);
            const bool room = (g.pa == g.pb);            // single-panel group can adopt a second panel
            if (has) { best = &g; break; }
            if (room && !best) best = &g;
        }
        if (!best) { groups.push_back(Group{hb.pr, hb.pr, {}}); best = &groups.back(); }
        if (best->pa != hb.pr && best->pb != hb.pr) best->pb = hb.pr;   // adopt as second panel
        best->bl.push_back(hb);
    }
    items.clear();
    for (auto& g : groups) {
        GramItem it{};
        it.panA = std::min(g.pa, g.pb);
        it.panB = std::max(g.pa, g.pb);
        for (size_t wv = 0; wv < g.bl.size(); ++wv) {
            const HostBlock& hb = g.bl[wv];
            WaveBlock& w = it.wb[wv];
            const int selr = (hb.pr == it.panA) ? 0 : 1, selc = (hb.pc == it.panA) ? 0 : 1;
            int first_r = -1, first_c = -1;
            for (int i = 0; i < MR; ++i) if (hb.ra[i] >= 0 && first_r < 0) first_r = hb.ra[i];
            for (int j = 0; j < NR; ++j) if (hb.cb[j] >= 0 && first_c < 0) first_c = hb.cb[j];
            unsigned short mask = 0;
            for (int i = 0; i < MR; ++i) {
                const int tr = hb.ra[i] >= 0 ? hb.ra[i] : first_r;      // unused slots alias a valid tile
                w.a[i] = (unsigned char)((selr << 3) | (tr & 7));
            }
            for (int j = 0; j < NR; ++j) {
                const int tc = hb.cb[j] >= 0 ? hb.cb[j] : first_c;
                w.b[j] = (unsigned char)((selc << 3) | (tc & 7));
            }
            for (int i = 0; i < MR; ++i)
                for (int j = 0; j < NR; ++j)
                    if (hb.ra[i] >= 0 && hb.cb[j] >= 0 && hb.ra[i] <= hb.cb[j]) mask |= (unsigned short)(1u << (i * NR + j));
            w.mask = mask;
            w.tri = (hb.ra[0] == hb.cb[0] && hb.pr == hb.pc) ? 1 : 0;
        }
        items.push_back(it);
    }
}

// List-mode plan: every tile on/above the diagonal goes to exactly one (item, wave) list; items are panel
// pairs (split into sub-items when a pair holds more than 4*LIST_CAP tiles); lists are padded with dummy
// tiles up to nt_max.  Returns false if no split up to 64 sub-items fits.
static bool build_list_items(int p, std::vector<GramItem>& items, int& nt_max) {
    const int ntile = (p + TILE - 1) / TILE;
    const int npan = (p + PANEL - 1) / PANEL;
    struct Tile { int ti, tj; };
    auto pan_of = [](int t) { return t / 8; };
    std::vector<std::pair<int, int>> pairs;
    if (npan == 1) pairs.push_back({0, 0});
    else for (int x = 0; x < npan; ++x) for (int y = x + 1; y < npan; ++y) pairs.push_back({x, y});
    for (int sub = 1; sub <= 64; ++sub) {
        std::vector<std::vector<Tile>> lists(pairs.size() * sub);
        std::vector<std::pair<int, int>> item_pair;
        for (auto& pr : pairs) for (int s2 = 0; s2 < sub; ++s2) item_pair.push_back(pr);
        std::vector<int> rr(pairs.size(), 0);
        for (int ti = 0; ti < ntile; ++ti)
            for (int tj = ti; tj < ntile; ++tj) {
                if (pan_of(ti) == pan_of(tj)) continue;
                size_t k = 0;
                for (; k < pairs.size(); ++k) if (pairs[k].first == pan_of(ti) && pairs[k].second == pan_of(tj)) break;
                lists[k * sub + (rr[k]++ % sub)].push_back({ti, tj});
            }
        // tiles inside one panel: round-robin over the panels, each to the least loaded item staging that panel
        std::vector<std::vector<Tile>> diag(npan);
        for (int ti = 0; ti < ntile; ++ti)
            for (int tj = ti; tj < ntile; ++tj)
                if (pan_of(ti) == pan_of(tj)) diag[pan_of(ti)].push_back({ti, tj});
        bool any = true;
        for (size_t turn = 0; any; ++turn) {
            any = false;
            for (int pn = 0; pn < npan; ++pn) {
                if (turn >= diag[pn].size()) continue;
                any = true;
                int best = -1;
                for (size_t k = 0; k < lists.size(); ++k) {
                    if (item_pair[k].first != pn && item_pair[k].second != pn) continue;
                    if (best < 0 || lists[k].size() < lists[best].size()) best = (int)k;
                }
                lists[best].push_back(diag[pn][turn]);
            }
        }
        bool ok = true;
        nt_max = 0;
        items.clear();
        for (size_t k = 0; k < lists.size() && ok; ++k) {
            if (lists[k].empty()) continue;
            GramItem g{};
            g.panA = item_pair[k].first;
            g.panB = item_pair[k].second;
            std::sort(lists[k].begin(), lists[k].end(), [](const Tile& x, const Tile& y) {
                return x.ti != y.ti ? x.ti < y.ti : x.tj < y.tj; });
            const int cnt = (int)lists[k].size();
            int pos = 0;
            for (int wv = 0; wv < GRAM_WAVES; ++wv) {
                const int take = cnt / GRAM_WAVES + (wv < cnt % GRAM_WAVES ? 1 : 0);
                if (take > LIST_CAP) { ok = false; break; }
                for (int n = 0; n < take; ++n, ++pos) {
                    const Tile& t = lists[k][pos];
                    const int selA = (pan_of(t.ti) == g.panA) ? 0 : 1, selB = (pan_of(t.tj) == g.panA) ? 0 : 1;
                    g.list[wv][n] = (unsigned short)((selA << 7) | ((t.ti & 7) << 4) | (selB << 3) | (t.tj & 7));
                }
                g.lnt[wv] = take;
                nt_max = std::max(nt_max, take);
            }
            if (ok) {
                for (int wv = 0; wv < GRAM_WAVES; ++wv)
                    for (int n = g.lnt[wv]; n < LIST_CAP; ++n) g.list[wv][n] = (unsigned short)(0x100 | (g.list[0][0] & 0xFF));
                items.push_back(g);
            }
        }
        if (ok) return true;
    }
    return false;
}

// Decide between the blocked and the list plan from the measured per-tile-step costs
// (bench/ubench_gram_inner.hip: ~68 cycles with 4x4 blocking, ~86 with one fragment pair per MFMA).
// nt_list = 0 selects the blocked plan; otherwise the list template size (4, 8 or 12).
static void build_plan_items(int p, std::vector<GramItem>& blocked, std::vector<GramItem>& listed, int& nt_list) {
    build_items(p, blocked);
    int nt_max = 0;
    nt_list = 0;
    double cost_blocked = 0.0;
    for (auto& g : blocked) {
        int worst = 0;
        for (int wv = 0; wv < GRAM_WAVES; ++wv) if (g.wb[wv].mask) worst = std::max(worst, g.wb[wv].tri ? 10 : MR * NR);
        cost_blocked += worst * 68.0;
    }
    if (build_list_items(p, listed, nt_max)) {
        const int nt = nt_max <= 4 ? 4 : (nt_max <= 8 ? 8 : 12);
        const double cost_list = (double)listed.size() * nt * 86.0;
        if (cost_list < 0.9 * cost_blocked) nt_list = nt;
    }
    if (!nt_list) listed.clear();
}

static std::mutex g_plan_mu;
static std::map<std::pair<int, int>, GramPlan> g_plans;   // (device, p)

static int get_plan(int p, GramPlan& out) {
    int dev = 0;
    DLSA_HIP_CHECK(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(g_plan_mu);
    auto key = std::make_pair(dev, p);
    auto f = g_plans.find(key);
    if (f != g_plans.end()) { out = f->second; return DLSA_OK; }
    GramPlan pl;
    pl.p = p;
    pl.ntile = (p + TILE - 1) / TILE;
    pl.npan = (p + PANEL - 1) / PANEL;
    pl.PP = (pl.ntile * TILE + 63) / 64 * 64;      // blocked mode stores whole 64x64 blocks
    std::vector<GramItem> items, listed;
    build_plan_items(p, items, listed, pl.nt_list);
    pl.nitems = (int)items.size();
    pl.nitems_list = (int)listed.size();
    // the item tables are a few KB of immutable metadata, created once per (device, p)
    DLSA_HIP_CHECK(hipMalloc((void**)&pl.d_items, items.size() * sizeof(GramItem)));
    DLSA_HIP_CHECK(hipMemcpy(pl.d_items, items.data(), items.size() * sizeof(GramItem), hipMemcpyHostToDevice));
    if (pl.nt_list) {
        DLSA_HIP_CHECK(hipMalloc((void**)&pl.d_items_list, listed.size() * sizeof(GramItem)));
        DLSA_HIP_CHECK(hipMemcpy(pl.d_items_list, listed.data(), listed.size() * sizeof(GramItem), hipMemcpyHostToDevice));
    }
    g_plans[key] = pl;
    out = pl;
    return DLSA_OK;
}

static void choose_slabs(int64_t n, int nitems, int& nslab, int64_t& rows_per_slab) {
    // Workgroups resident at once: OCC per CU.  Pick nslab so that nitems*nslab is (just under) a whole
    // number of rounds of resident workgroups -- no tail -- with about 32k rows per slab (few, large
    // partials: 1 round for a 1e6-row partition, nitems rounds for the 2.5e7-row shard), at least 256
    // rows per slab for small inputs, and a multiple of 8 slabs for the XCD-aware mapping.
    const int64_t resident = (int64_t)DLSA_GRAM_OCC * kNumCU;
    nitems = std::max(1, nitems);
    int64_t rounds = (n * nitems + resident * 16384) / (resident * 32768);
    rounds = std::min<int64_t>(std::max<int64_t>(rounds, 1), nitems);
    int64_t ns = std::max<int64_t>(1, resident * rounds / nitems);
    const int64_t by_rows = std::max<int64_t>(1, (n + 255) / 256);
    ns = std::min(ns, by_rows);
    if (ns >= kNumXCD) ns = ns / kNumXCD * kNumXCD;
    rows_per_slab = ((n + ns - 1) / ns + KC - 1) / KC * KC;
    if (rows_per_slab < KC) rows_per_slab = KC;
    ns = std::max<int64_t>(1, (n + rows_per_slab - 1) / rows_per_slab);
    if (ns >= kNumXCD && ns % kNumXCD) ns = (ns + kNumXCD - 1) / kNumXCD * kNumXCD;  // empty tail slabs write zeros
    nslab = (int)ns;
}

// Upper bound of choose_slabs' slab count that is MONOTONE in n.  A workspace sized for the largest partition of a fit is reused for
// every smaller row count (other partitions, subsample levels), and the exact count is not monotone: the rounding of rows_per_slab to
// whole chunks makes 319489 rows take 504 slabs and 319488 rows 512 (found by bench/fit_fuzz.py: the smaller partition of an
// i % 2 split did not fit the workspace of the larger one).
static int slab_bound(int64_t n, int nitems) {
    const int64_t resident = (int64_t)DLSA_GRAM_OCC * kNumCU;
    nitems = std::max(1, nitems);
    int64_t rounds = (n * nitems + resident * 16384) / (resident * 32768);
    rounds = std::min<int64_t>(std::max<int64_t>(rounds, 1), nitems);
    int64_t ns = std::max<int64_t>(1, resident * rounds / nitems);
    ns = std::min(ns, std::max<int64_t>(1, (n + 255) / 256));
    return (int)((ns + kNumXCD - 1) / kNumXCD * kNumXCD);          // the final count never exceeds the first estimate, rounded up to the XCDs
}

// gram_wide.hip: the 256-column-panel fp32 kernel for wide p
bool gram_wide_f32_shape_ok(int64_t n, int p);
bool gram_wide_f32_eligible(const float* X, int64_t ldx, const float* w, int64_t n, int p);
size_t gram_wide_f32_ws_bytes(int64_t n, int p);
int gram_wide_f32(const float* X, int64_t ldx, const float* w, int64_t n, int p, float* H, int64_t ldh,
                  int accumulate, void* ws, size_t ws_bytes, hipStream_t stream, double* H64 = nullptr);

// gram_narrow.hip: the row-split fp64 kernel for 49 <= p <= 120
bool gram_narrow_shape_ok(int64_t n, int p);
bool gram_narrow_eligible(const double* X, int64_t ldx, const double* w, int64_t n, int p);
size_t gram_narrow_ws_bytes(int64_t n, int p);
int gram_narrow_f64(const double* X, int64_t ldx, const double* w, int64_t n, int p, double* H, int64_t ldh,
                    int accumulate, void* ws, size_t ws_bytes, hipStream_t stream);

// gram_cyclic.hip: the balanced cyclic-tile fp64 kernel for the p = 500 class (481 <= p <= 508)
bool gram_cyclic_shape_ok(int64_t n, int p);
bool gram_cyclic_eligible(const double* X, int64_t ldx, const double* w, int64_t n, int p);
size_t gram_cyclic_ws_bytes(int64_t n, int p);
int gram_cyclic_f64(const double* X, int64_t ldx, const double* w, int64_t n, int p, double* H, int64_t ldh,
                    int accumulate, void* ws, size_t ws_bytes, hipStream_t stream);

// gram_plan.hip: generated per-wave tile plans on 1 / 2 / 4-CU groups, interleaved tile pairs, 125 <= p <= 572 (fp64)
bool gram_plan_shape_ok(int64_t n, int p);
bool gram_plan_eligible(const double* X, int64_t ldx, const double* w, int64_t n, int p);
size_t gram_plan_ws_bytes(int64_t n, int p);
int gram_plan_f64(const double* X, int64_t ldx, const double* w, int64_t n, int p, double* H, int64_t ldh,
                  int accumulate, void* ws, size_t ws_bytes, hipStream_t stream);

static size_t gram_ws_bytes(int64_t n, int p, int elem_bytes) {
    const int ntile = (p + TILE - 1) / TILE;
    std::vector<GramItem> items, listed;
    int nt_list = 0;
    build_plan_items(p, items, listed, nt_list);      // host-only, cheap: the exact item counts
    const int nslab = slab_bound(n, (int)items.size()), nslab2 = nt_list ? slab_bound(n, (int)listed.size()) : 0;
    const size_t PP = ((size_t)ntile * TILE + 63) / 64 * 64;
    size_t bytes = align_up((size_t)std::max(nslab, nslab2) * PP * PP * elem_bytes, 256);
    if (elem_bytes == 4 && gram_wide_f32_shape_ok(n, p)) bytes = std::max(bytes, gram_wide_f32_ws_bytes(n, p));
    if (elem_bytes == 8 && gram_narrow_shape_ok(n, p)) bytes = std::max(bytes, gram_narrow_ws_bytes(n, p));
    if (elem_bytes == 8 && gram_plan_shape_ok(n, p)) bytes = std::max(bytes, gram_plan_ws_bytes(n, p));
    if (elem_bytes == 8 && gram_cyclic_shape_ok(n, p)) bytes = std::max(bytes, gram_cyclic_ws_bytes(n, p));
    return bytes;
}

// H64 (fp32 rows only): the slab partials are summed in fp64 and go to / into this fp64 matrix instead of H
template <typename T>
int gram_impl(const T* X, int64_t ldx, const T* w, int64_t n, int p, T* H, int64_t ldh,
              int accumulate, void* ws, size_t ws_bytes, hipStream_t stream, double* H64 = nullptr) {
    DLSA_REQUIRE((X || n == 0) && (H || H64), "gram: null X or H");      // an empty row block (n == 0) gives H = 0
    DLSA_REQUIRE(p > 0 && n >= 0 && ldx >= p && ldh >= p, "gram: bad shape n=%lld p=%d ldx=%lld ldh=%lld",
                 (long long)n, p, (long long)ldx, (long long)ldh);
    if constexpr (sizeof(T) == 4) {
        if (gram_wide_f32_eligible(X, ldx, w, n, p))
            return gram_wide_f32(X, ldx, w, n, p, H, ldh, accumulate, ws, ws_bytes, stream, H64);
    } else {
        if (gram_narrow_eligible(X, ldx, w, n, p))
            return gram_narrow_f64(X, ldx, w, n, p, H, ldh, accumulate, ws, ws_bytes, stream);
        if (gram_cyclic_eligible(X, ldx, w, n, p))            // the p = 500 class: still ~1.4 % ahead of the plan kernel there
            return gram_cyclic_f64(X, ldx, w, n, p, H, ldh, accumulate, ws, ws_bytes, stream);
        if (gram_plan_eligible(X, ldx, w, n, p))
            return gram_plan_f64(X, ldx, w, n, p, H, ldh, accumulate, ws, ws_bytes, stream);
    }
    GramPlan pl;
    int rc = get_plan(p, pl);
    if (rc) return rc;
    const bool vec = (ldx % 2 == 0) && (((uintptr_t)X % (2 * sizeof(T))) == 0);
    int dbg = 0;
    dbg = gram_dbg_env();
    int mode = vec ? 1 : 0;
    const bool use_list = pl.nt_list != 0 && vec && !(dbg & 32);      // list plan needs the aligned staging paths
    const int nitems = use_list ? pl.nitems_list : pl.nitems;
    const int nt_list = use_list ? pl.nt_list : 0;
    int nslab; int64_t rps;
    choose_slabs(n, nitems, nslab, rps);
    const size_t need = (size_t)nslab * pl.PP * pl.PP * sizeof(T);
    if (!ws || ws_bytes < need || ((uintptr_t)ws & 255)) {
        set_error("gram: workspace %zu bytes needed (256-aligned), got %zu", need, ws_bytes);
        return DLSA_ERR_WORKSPACE;
    }
    GramArgs<T> a;
    a.X = X; a.w = w; a.partial = (T*)ws; a.items = use_list ? pl.d_items_list : pl.d_items; a.ldx = ldx; a.n = n;
    // Odd p in an even row pitch (e.g. an intercept column in front of an even design): the kernel loads p + 1
    // columns, so that rows stay 16-byte units for the vector / LDS-DMA staging.  Whatever sits in the pad column
    // (even NaN) only reaches row and column p of the tile grid, which nobody reads: an MFMA output element depends on
    // one row of A and one column of B only.  ceil(p / 16) does not change, p being odd.
    const int p_load = (vec && (p & 1)) ? p + 1 : p;
    a.rows_per_slab = rps; a.p = p_load; a.PP = pl.PP; a.nitems = nitems; a.nslab = nslab;
    a.xcd_map = (nslab % kNumXCD == 0) ? 1 : 0;
    a.dbg = dbg;
    if (a.dbg & 2) a.xcd_map = 0;
    if (sizeof(T) == 8 && vec && (p_load % 2 == 0) && (!w || ((uintptr_t)w % 16) == 0) &&
        (double)rps * (double)ldx * sizeof(T) < 2.0e9)
        mode = 2;                                        // direct global->LDS DMA
    if (a.dbg & 4) mode = vec ? 1 : 0;
    const int blocks = nitems * nslab;
#define DLSA_LAUNCH_GRAM_NT(HW, MD, NTV) hipLaunchKernelGGL((gram_kernel<T, HW, MD, NTV>), dim3(blocks), dim3(GRAM_THREADS), 0, stream, a)
#define DLSA_LAUNCH_GRAM(HW, MD) do { \
        if (MD == 0 || nt_list == 0) DLSA_LAUNCH_GRAM_NT(HW, MD, 0); \
        else if (nt_list == 4) DLSA_LAUNCH_GRAM_NT(HW, (MD == 0 ? 1 : MD), 4); \
        else if (nt_list == 8) DLSA_LAUNCH_GRAM_NT(HW, (MD == 0 ? 1 : MD), 8); \
        else DLSA_LAUNCH_GRAM_NT(HW, (MD == 0 ? 1 : MD), 12); } while (0)
    if (w) {
        if (mode == 2) { if constexpr (sizeof(T) == 8) DLSA_LAUNCH_GRAM(true, 2); }
        else if (mode == 1) DLSA_LAUNCH_GRAM(true, 1);
        else DLSA_LAUNCH_GRAM(true, 0);
    } else {
        if (mode == 2) { if constexpr (sizeof(T) == 8) DLSA_LAUNCH_GRAM(false, 2); }
        else if (mode == 1) DLSA_LAUNCH_GRAM(false, 1);
        else DLSA_LAUNCH_GRAM(false, 0);
    }
#undef DLSA_LAUNCH_GRAM
#undef DLSA_LAUNCH_GRAM_NT
    DLSA_HIP_CHECK(hipGetLastError());
    note_gram_kernel(nullptr, stream, "gram_kernel<%s,%s,%d,%d> (panel kernel%s)", sizeof(T) == 8 ? "double" : "float", w ? "true" : "false",
                     mode, mode == 0 ? 0 : nt_list, nt_list ? ", tile-list plan" : "");
    if constexpr (sizeof(T) == 4) {
        if (H64) gram_reduce_launch_f32_to_f64((const float*)ws, nslab, pl.PP, p, H64, ldh, accumulate, stream);
        else gram_reduce_launch<T>((const T*)ws, nslab, pl.PP, p, H, ldh, accumulate, stream);
    } else {
        gram_reduce_launch<T>((const T*)ws, nslab, pl.PP, p, H, ldh, accumulate, stream);
    }
    DLSA_HIP_CHECK(hipGetLastError());
    return DLSA_OK;
}

template int gram_impl<double>(const double*, int64_t, const double*, int64_t, int, double*, int64_t, int, void*, size_t, hipStream_t, double*);
template int gram_impl<float>(const float*, int64_t, const float*, int64_t, int, float*, int64_t, int, void*, size_t, hipStream_t, double*);

size_t gram_workspace_bytes_impl(int64_t n, int p, int elem_bytes) { return gram_ws_bytes(n, p, elem_bytes); }

int gram_impl_f64(const double* X, int64_t ldx, const double* w, int64_t n, int p, double* H, int64_t ldh,
                  int accumulate, void* ws, size_t ws_bytes, hipStream_t stream) {
    return gram_impl<double>(X, ldx, w, n, p, H, ldh, accumulate, ws, ws_bytes, stream);
}

// host-only self check of the tile plan: every tile on/above the diagonal is stored exactly once,
// by a wave whose workgroup stages both of its panels.  Used by the CPU test-suite.
// Outputs: number of workgroup items, tile slots computed (incl. waste), tiles stored.
int gram_plan_check(int p, int* nitems, int* nslots, int* ntiles) {
    std::vector<GramItem> items, listed;
    int nt_list = 0;
    build_plan_items(p, items, listed, nt_list);
    const int ntile = (p + TILE - 1) / TILE;
    std::vector<int> seen((size_t)ntile * ntile, 0);
    int count = 0, slots = 0;
    for (auto& g : items) {
        if (g.panA > g.panB) return -5;
        for (int wv = 0; wv < GRAM_WAVES; ++wv) {
            const WaveBlock& w = g.wb[wv];
            if (w.mask) slots += w.tri ? 10 : MR * NR;
            for (int i = 0; i < MR; ++i)
                for (int j = 0; j < NR; ++j) {
                    if (!((w.mask >> (i * NR + j)) & 1)) continue;
                    const int ti = (((w.a[i] >> 3) & 1) ? g.panB : g.panA) * 8 + (w.a[i] & 7);
                    const int tj = (((w.b[j] >> 3) & 1) ? g.panB : g.panA) * 8 + (w.b[j] & 7);
                    if (ti > tj || tj >= ntile) return -1;
                    if (seen[(size_t)ti * ntile + tj]++) return -2;
                    ++count;
                }
        }
    }
    if (count != ntile * (ntile + 1) / 2) return -3;
    if (nt_list) {      // the list plan must cover the same tiles exactly once as well
        std::fill(seen.begin(), seen.end(), 0);
        int lcount = 0;
        for (auto& g : listed) {
            if (g.panA > g.panB) return -15;
            for (int wv = 0; wv < GRAM_WAVES; ++wv) {
                if (g.lnt[wv] > nt_list) return -14;
                for (int t = 0; t < LIST_CAP; ++t) {
                    const int code = g.list[wv][t];
                    if (t >= g.lnt[wv]) { if (!((code >> 8) & 1)) return -16; continue; }
                    if ((code >> 8) & 1) return -17;
                    const int ti = (((code >> 7) & 1) ? g.panB : g.panA) * 8 + ((code >> 4) & 7);
                    const int tj = (((code >> 3) & 1) ? g.panB : g.panA) * 8 + (code & 7);
                    if (ti > tj || tj >= ntile) return -11;
                    if (seen[(size_t)ti * ntile + tj]++) return -12;
                    ++lcount;
                }
            }
        }
        if (lcount != ntile * (ntile + 1) / 2) return -13;
        slots = (int)listed.size() * GRAM_WAVES * nt_list;     // the plan that will run
        if (nitems) *nitems = -(int)listed.size();              // negative = list plan selected
    } else if (nitems) {
        *nitems = (int)items.size();
    }
    if (nslots) *nslots = slots;
    if (ntiles) *ntiles = count;
    return 0;
}

}  // namespace dlsa

extern "C" {

// debugging/test hook (not part of the drop-in surface): validates the tile plan for p
int dlsa_gram_plan_check(int p, int* nitems, int* nslots, int* ntiles) {
    return dlsa::gram_plan_check(p, nitems, nslots, ntiles);
}

size_t dlsa_gram_workspace_bytes(int64_t n, int p, int elem_bytes) {
    if (p <= 0 || n < 0 || (elem_bytes != 4 && elem_bytes != 8)) return 0;
    return dlsa::gram_workspace_bytes_impl(n, p, elem_bytes);
}

int dlsa_gram_f64(const double* X, int64_t ldx, const double* w, int64_t n, int p, double* H,
                  int64_t ldh, int accumulate, void* ws, size_t ws_bytes, void* stream) {
    return dlsa::gram_impl<double>(X, ldx, w, n, p, H, ldh, accumulate, ws, ws_bytes, (hipStream_t)stream);
}

int dlsa_gram_f32(const float* X, int64_t ldx, const float* w, int64_t n, int p, float* H,
                  int64_t ldh, int accumulate, void* ws, size_t ws_bytes, void* stream) {
    return dlsa::gram_impl<float>(X, ldx, w, n, p, H, ldh, accumulate, ws, ws_bytes, (hipStream_t)stream);
}

int dlsa_gram_f32_acc64(const float* X, int64_t ldx, const float* w, int64_t n, int p, double* H64,
                        int64_t ldh, int accumulate, void* ws, size_t ws_bytes, void* stream) {
    DLSA_REQUIRE(H64, "gram_f32_acc64: null H");
    return dlsa::gram_impl<float>(X, ldx, w, n, p, nullptr, ldh, accumulate, ws, ws_bytes, (hipStream_t)stream, H64);
}

}  // extern "C"
