// Tall-skinny weighted Gram  H = X' diag(w) X  on gfx950 MFMA  (reference call site:
// dlsa/models.py:130  Sig_inv = x_train.T.dot(np.multiply((prob*(1-prob))[:,None], x_train))).
//
// Shape of the problem: M = N = p (<= a few thousand), K = n rows (10^7..10^8).  It is a
// split-K GEMM whose output is tiny and symmetric, so
//   * only upper-triangular 16x16 tiles of H are computed, the reduce kernel mirrors them;
//   * the p columns are cut into 128-column PANELS; a work ITEM owns a pair of panels and a
//     balanced list of output tiles whose operands live in those two panels;
//   * a workgroup (8 waves, one per CU) = one item x one SLAB of rows.  It streams the slab
//     through LDS in 32-row chunks (double buffered), every wave keeps its tiles in VGPRs for
//     the whole slab and finally writes them to a per-slab partial buffer;
//   * blocks that share a slab are dispatched to the same XCD back to back so that the slab's
//     panels are fetched from HBM once and re-read from that XCD's L2;
//   * a second kernel sums the slab partials in a fixed order (deterministic) into H.
//
// MFMA operand layout (v_mfma_f64_16x16x4_f64): A[i=lane&15][k=lane>>4], B[k=lane>>4][j=lane&15],
// C/D reg r of lane l = C[4r + (l>>4)][l&15].  With A[i][k] = X[r0+k][ca+i] and
// B[k][j] = w[r0+k] X[r0+k][cb+j] both fragments are the SAME LDS read pattern
// "lane l <- chunk[(4ks + l>>4)][16t + (l&15)]", conflict-free at a row pitch of 144.
#include "common.h"
#include <vector>
#include <algorithm>
#include <mutex>
#include <map>

namespace dlsa {

constexpr int TILE = 16;
constexpr int PANEL = 128;        // columns per panel = 8 tiles
constexpr int KC = 32;            // rows per staged chunk = 8 MFMA k-steps
constexpr int LDP = 144;          // LDS row pitch (elements): 144 mod 32 == 16 -> conflict-free frags
constexpr int GRAM_WAVES = 8;
constexpr int GRAM_THREADS = 64 * GRAM_WAVES;
constexpr int NT_CAP = 11;        // max tiles per wave (11*8 = 88 accumulator VGPRs in fp64)

struct GramItem {
    int panA, panB;               // panel indices (panB == panA: single-panel item)
    int nt[GRAM_WAVES];           // tiles per wave
    unsigned short tile[GRAM_WAVES][NT_CAP];  // selA<<7 | tiA<<4 | selB<<3 | tjB   (ti,tj local 0..7)
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
    int PP;              // padded dimension = ntile*16
    int nitems;
    int nslab;
    int xcd_map;         // 1: XCD-aware block->(item,slab) mapping (needs nslab % 8 == 0)
};

template <typename T> struct Mfma;
template <> struct Mfma<double> {
    typedef double acc_t __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ acc_t run(double a, double b, acc_t c) {
        return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    }
    // C/D register r of lane l is row 4r + (l>>4)   (f64 layout differs from every other dtype)
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

template <typename T, int NT, bool HASW, bool VEC>
__global__ __launch_bounds__(GRAM_THREADS, 2) void gram_kernel(GramArgs<T> a) {
    typedef typename Mfma<T>::acc_t acc_t;
    typedef typename Vec2<T>::type vec2_t;
    constexpr int PASSES = KC / GRAM_WAVES;                     // staging passes per panel
    constexpr int PANEL_ELEMS = KC * LDP;
    constexpr int BUF_ELEMS = 2 * PANEL_ELEMS + KC;      // two panels + the w chunk
    __shared__ __attribute__((aligned(16))) T lds[2 * BUF_ELEMS];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // block -> (item, slab)
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
    const int nt = it->nt[wave];

    // LDS element offsets of every tile's A and B fragment origin (wave-uniform)
    int offA[NT], offB[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int code = (t < nt) ? it->tile[wave][t] : 0;
        offA[t] = ((code >> 7) & 1) * PANEL_ELEMS + ((code >> 4) & 7) * TILE;
        offB[t] = ((code >> 3) & 1) * PANEL_ELEMS + (code & 7) * TILE;
    }
    const int lane_off = (lane >> 4) * LDP + (lane & 15);

    const int64_t rbeg = (int64_t)slab * a.rows_per_slab;
    const int64_t rend = min(rbeg + a.rows_per_slab, a.n);
    const int nchunks = rbeg < rend ? (int)((rend - rbeg + KC - 1) / KC) : 0;

    acc_t acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = acc_t{0, 0, 0, 0};

    vec2_t st[2][PASSES];
    T wreg = T(0);
    const int srow = wave;          // + GRAM_WAVES*pass
    const int scol = lane * 2;

    auto stage_load = [&](int chunk) {
        const int64_t r0 = rbeg + (int64_t)chunk * KC;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            if (s < npanels) {
                const int gcol = (s == 0 ? panA : panB) * PANEL + scol;
#pragma unroll
                for (int ps = 0; ps < PASSES; ++ps)
                    st[s][ps] = load_pair<T, VEC>(a.X, a.ldx, r0 + srow + GRAM_WAVES * ps, rend, gcol, a.p);
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

    if (nchunks > 0) {
        stage_load(0);
        stage_write(0);
    }
    __syncthreads();

    for (int c = 0; c < nchunks; ++c) {
        if (c + 1 < nchunks) stage_load(c + 1);
        const T* base = lds + (c & 1) * BUF_ELEMS;
#pragma unroll
        for (int ks = 0; ks < KC / 4; ++ks) {
            T wv = T(1);
            if (HASW) wv = base[2 * PANEL_ELEMS + ks * 4 + (lane >> 4)];
            const T* kb = base + ks * 4 * LDP + lane_off;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                if (t < nt) {
                    const T av = kb[offA[t]];
                    T bv = kb[offB[t]];
                    if (HASW) bv *= wv;
                    acc[t] = Mfma<T>::run(av, bv, acc[t]);
                }
            }
        }
        if (c + 1 < nchunks) stage_write((c + 1) & 1);
        __syncthreads();
    }

    // epilogue: accumulator tiles -> this slab's partial buffer
    T* __restrict__ P = a.partial + (int64_t)slab * a.PP * a.PP;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        if (t < nt) {
            const int code = it->tile[wave][t];
            const int r0 = ((((code >> 7) & 1) ? panB : panA) * 8 + ((code >> 4) & 7)) * TILE;
            const int c0 = ((((code >> 3) & 1) ? panB : panA) * 8 + (code & 7)) * TILE;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                P[(int64_t)(r0 + Mfma<T>::crow(lane, r)) * a.PP + c0 + (lane & 15)] = acc[t][r];
        }
    }
}

// H[i][j] = (accumulate ? H[i][j] : 0) + sum_s partial[s][min(i,j)][max(i,j)]   (fixed order)
template <typename T>
__global__ void gram_reduce_kernel(const T* __restrict__ partial, int nslab, int PP, int p,
                                   T* __restrict__ H, int64_t ldh, int accumulate) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y;
    if (j >= p) return;
    const int r = min(i, j), c = max(i, j);
    const T* src = partial + (int64_t)r * PP + c;
    T s = T(0);
    const int64_t stride = (int64_t)PP * PP;
    for (int k = 0; k < nslab; ++k) s += src[k * stride];
    T* dst = H + (int64_t)i * ldh + j;
    *dst = accumulate ? (*dst + s) : s;
}

// ---------------------------------------------------------------------------------------
// Host side: tile -> item -> wave assignment, cached per p on the device.
// ---------------------------------------------------------------------------------------
struct GramPlan {
    int p = 0, ntile = 0, npan = 0, PP = 0, nitems = 0, nt_max = 0;
    GramItem* d_items = nullptr;
};

static bool try_build_items(int p, int sub, std::vector<GramItem>& items, int& nt_max) {
    const int ntile = (p + TILE - 1) / TILE;
    const int npan = (p + PANEL - 1) / PANEL;
    struct Tile { int ti, tj; };
    std::vector<std::pair<int, int>> pairs;
    if (npan == 1) pairs.push_back({0, 0});
    else for (int x = 0; x < npan; ++x) for (int y = x + 1; y < npan; ++y) pairs.push_back({x, y});
    std::vector<std::vector<Tile>> lists;
    std::vector<std::pair<int, int>> item_pair;
    for (auto& pr : pairs) for (int s = 0; s < sub; ++s) { item_pair.push_back(pr); lists.emplace_back(); }
    auto pan_of = [](int t) { return t / 8; };
    auto pair_index = [&](int x, int y) {
        for (size_t k = 0; k < pairs.size(); ++k) if (pairs[k].first == x && pairs[k].second == y) return (int)k;
        return -1;
    };
    // tiles whose operands sit in two different panels belong to that panel pair
    std::vector<int> rr(pairs.size(), 0);
    for (int ti = 0; ti < ntile; ++ti)
        for (int tj = ti; tj < ntile; ++tj) {
            if (pan_of(ti) == pan_of(tj)) continue;
            const int k = pair_index(pan_of(ti), pan_of(tj));
            lists[k * sub + (rr[k]++ % sub)].push_back({ti, tj});
        }
    // tiles inside one panel go to the least loaded item that stages that panel; the panels are
    // visited round-robin (one tile each per turn) so that no item fills up early
    {
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
    }
    items.clear();
    nt_max = 0;
    for (size_t k = 0; k < lists.size(); ++k) {
        if (lists[k].empty()) continue;
        if ((int)lists[k].size() > NT_CAP * GRAM_WAVES) return false;
        GramItem g{};
        g.panA = item_pair[k].first;
        g.panB = item_pair[k].second;
        std::sort(lists[k].begin(), lists[k].end(), [](const Tile& x, const Tile& y) {
            return x.ti != y.ti ? x.ti < y.ti : x.tj < y.tj; });
        int wv = 0;
        for (auto& t : lists[k]) {
            const int selA = (pan_of(t.ti) == g.panA) ? 0 : 1;
            const int selB = (pan_of(t.tj) == g.panA) ? 0 : 1;
            const int n = g.nt[wv];
            g.tile[wv][n] = (unsigned short)((selA << 7) | ((t.ti & 7) << 4) | (selB << 3) | (t.tj & 7));
            g.nt[wv] = n + 1;
            nt_max = std::max(nt_max, n + 1);
            wv = (wv + 1) % GRAM_WAVES;
        }
        items.push_back(g);
    }
    return nt_max <= NT_CAP;
}

// Items: one or more per panel pair so that no wave holds more than NT_CAP tiles.
// p = 500: 6 panel pairs x 88 tiles = 528 upper-triangular tiles, exactly 11 per wave.
static void build_items(int p, std::vector<GramItem>& items, int& nt_max) {
    for (int sub = 1; sub < 64; ++sub)
        if (try_build_items(p, sub, items, nt_max)) return;
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
    pl.PP = pl.ntile * TILE;
    std::vector<GramItem> items;
    build_items(p, items, pl.nt_max);
    if (pl.nt_max > NT_CAP) { set_error("gram plan: %d tiles per wave exceeds %d", pl.nt_max, NT_CAP); return DLSA_ERR_INVALID; }
    pl.nitems = (int)items.size();
    // the item table is a few KB of immutable metadata, created once per (device, p)
    DLSA_HIP_CHECK(hipMalloc((void**)&pl.d_items, items.size() * sizeof(GramItem)));
    DLSA_HIP_CHECK(hipMemcpy(pl.d_items, items.data(), items.size() * sizeof(GramItem), hipMemcpyHostToDevice));
    g_plans[key] = pl;
    out = pl;
    return DLSA_OK;
}

static void choose_slabs(int64_t n, int nitems, int& nslab, int64_t& rows_per_slab) {
    // ~6 rounds of 256 resident workgroups for large n; at least 512 rows per slab
    const int target_blocks = 6 * kNumCU;   // one 8-wave workgroup per CU, ~6 rounds
    int64_t want = std::max<int64_t>(1, target_blocks / std::max(1, nitems));
    int64_t by_rows = std::max<int64_t>(1, (n + 511) / 512);
    int64_t ns = std::min(want, by_rows);
    if (ns >= kNumXCD) ns = ns / kNumXCD * kNumXCD;
    rows_per_slab = ((n + ns - 1) / ns + KC - 1) / KC * KC;
    if (rows_per_slab < KC) rows_per_slab = KC;
    ns = std::max<int64_t>(1, (n + rows_per_slab - 1) / rows_per_slab);
    if (ns >= kNumXCD && ns % kNumXCD) ns = (ns + kNumXCD - 1) / kNumXCD * kNumXCD;  // empty tail slabs write zeros
    nslab = (int)ns;
}

static size_t gram_ws_bytes(int64_t n, int p, int elem_bytes) {
    const int ntile = (p + TILE - 1) / TILE;
    std::vector<GramItem> items;
    int nt_max = 0;
    build_items(p, items, nt_max);   // host-only, cheap; gives the exact item count
    int nslab; int64_t rps;
    choose_slabs(n, (int)items.size(), nslab, rps);
    const size_t PP = (size_t)ntile * TILE;
    return align_up((size_t)nslab * PP * PP * elem_bytes, 256);
}

template <typename T, int NT>
static void launch_gram(const GramArgs<T>& a, bool hasw, bool vec, int blocks, hipStream_t s) {
    if (hasw) {
        if (vec) hipLaunchKernelGGL((gram_kernel<T, NT, true, true>), dim3(blocks), dim3(GRAM_THREADS), 0, s, a);
        else hipLaunchKernelGGL((gram_kernel<T, NT, true, false>), dim3(blocks), dim3(GRAM_THREADS), 0, s, a);
    } else {
        if (vec) hipLaunchKernelGGL((gram_kernel<T, NT, false, true>), dim3(blocks), dim3(GRAM_THREADS), 0, s, a);
        else hipLaunchKernelGGL((gram_kernel<T, NT, false, false>), dim3(blocks), dim3(GRAM_THREADS), 0, s, a);
    }
}

template <typename T>
int gram_impl(const T* X, int64_t ldx, const T* w, int64_t n, int p, T* H, int64_t ldh,
              int accumulate, void* ws, size_t ws_bytes, hipStream_t stream) {
    DLSA_REQUIRE(X && H, "gram: null X or H");
    DLSA_REQUIRE(p > 0 && n >= 0 && ldx >= p && ldh >= p, "gram: bad shape n=%lld p=%d ldx=%lld ldh=%lld",
                 (long long)n, p, (long long)ldx, (long long)ldh);
    GramPlan pl;
    int rc = get_plan(p, pl);
    if (rc) return rc;
    int nslab; int64_t rps;
    choose_slabs(n, pl.nitems, nslab, rps);
    const size_t need = (size_t)nslab * pl.PP * pl.PP * sizeof(T);
    if (!ws || ws_bytes < need || ((uintptr_t)ws & 255)) {
        set_error("gram: workspace %zu bytes needed (256-aligned), got %zu", need, ws_bytes);
        return DLSA_ERR_WORKSPACE;
    }
    GramArgs<T> a;
    a.X = X; a.w = w; a.partial = (T*)ws; a.items = pl.d_items; a.ldx = ldx; a.n = n;
    a.rows_per_slab = rps; a.p = p; a.PP = pl.PP; a.nitems = pl.nitems; a.nslab = nslab;
    a.xcd_map = (nslab % kNumXCD == 0) ? 1 : 0;
    const bool vec = (ldx % 2 == 0) && (((uintptr_t)X % (2 * sizeof(T))) == 0);
    const int blocks = pl.nitems * nslab;
    if (pl.nt_max <= 6) launch_gram<T, 6>(a, w != nullptr, vec, blocks, stream);
    else launch_gram<T, NT_CAP>(a, w != nullptr, vec, blocks, stream);
    DLSA_HIP_CHECK(hipGetLastError());
    dim3 rg((p + 127) / 128, p);
    hipLaunchKernelGGL((gram_reduce_kernel<T>), rg, dim3(128), 0, stream, (const T*)ws, nslab, pl.PP, p, H, ldh, accumulate);
    DLSA_HIP_CHECK(hipGetLastError());
    return DLSA_OK;
}

template int gram_impl<double>(const double*, int64_t, const double*, int64_t, int, double*, int64_t, int, void*, size_t, hipStream_t);
template int gram_impl<float>(const float*, int64_t, const float*, int64_t, int, float*, int64_t, int, void*, size_t, hipStream_t);

size_t gram_workspace_bytes_impl(int64_t n, int p, int elem_bytes) { return gram_ws_bytes(n, p, elem_bytes); }

int gram_impl_f64(const double* X, int64_t ldx, const double* w, int64_t n, int p, double* H, int64_t ldh,
                  int accumulate, void* ws, size_t ws_bytes, hipStream_t stream) {
    return gram_impl<double>(X, ldx, w, n, p, H, ldh, accumulate, ws, ws_bytes, stream);
}

// host-only self check of the tile plan: every upper-triangular tile exactly once, inside an
// item that stages both of its panels.  Used by the CPU test-suite.
int gram_plan_check(int p, int* nitems, int* nt_max_out, int* ntiles) {
    std::vector<GramItem> items;
    int nt_max = 0;
    build_items(p, items, nt_max);
    const int ntile = (p + TILE - 1) / TILE;
    std::vector<int> seen((size_t)ntile * ntile, 0);
    int count = 0;
    for (auto& g : items)
        for (int wv = 0; wv < GRAM_WAVES; ++wv)
            for (int t = 0; t < g.nt[wv]; ++t) {
                const int code = g.tile[wv][t];
                const int ti = (((code >> 7) & 1) ? g.panB : g.panA) * 8 + ((code >> 4) & 7);
                const int tj = (((code >> 3) & 1) ? g.panB : g.panA) * 8 + (code & 7);
                if (ti > tj || tj >= ntile) return -1;
                if (seen[(size_t)ti * ntile + tj]++) return -2;
                ++count;
            }
    if (count != ntile * (ntile + 1) / 2) return -3;
    if (nt_max > NT_CAP) return -4;
    if (nitems) *nitems = (int)items.size();
    if (nt_max_out) *nt_max_out = nt_max;
    if (ntiles) *ntiles = count;
    return 0;
}

}  // namespace dlsa

extern "C" {

// debugging/test hook (not part of the drop-in surface): validates the tile plan for p
int dlsa_gram_plan_check(int p, int* nitems, int* nt_max, int* ntiles) {
    return dlsa::gram_plan_check(p, nitems, nt_max, ntiles);
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

}  // extern "C"
