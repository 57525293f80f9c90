// Weighted Gram H = X' diag(w) X for NARROW fp64 designs (49 <= p <= 112 in an even row pitch: config 2's p = 100,
// config 1's p = 50, and the same with an intercept column),
// reference call site dlsa/models.py:130.
//
// At these widths the whole upper triangle of H is at most 28 tiles of 16x16, i.e. <= 224 accumulator registers
// in fp64 -- it fits in ONE wave's AGPRs.  So instead of cutting the tiles over the four waves of a workgroup (gram.hip's
// list plan: one fragment pair from LDS per MFMA, 86 instead of 64 cycles per tile-step, and 7 real tiles in 8
// slots at p = 100), the four waves split the ROWS: every wave owns all NT(NT+1)/2 tiles and takes every fourth
// 4-row k-step of a staged chunk.  Per k-step a wave reads its NT fragments once (A and B are the same fragments,
// B scaled by w), and issues NT(NT+1)/2 MFMAs with no wasted slot: 7 LDS reads per 28 MFMAs at p = 100.
// The accumulators live in AGPRs; the four partial triangles meet in LDS once per workgroup, at the end.
//
// Rows stream global -> LDS with the LDS-DMA (buffer_load ... lds, 16 bytes per lane, one instruction per row)
// through a ring of NSTAGE chunk buffers; two chunks are in flight while the third is consumed, with a counted
// s_waitcnt vmcnt(N) + raw s_barrier per chunk (the DMA completes in order).  One workgroup per CU at 6-7 tiles (the
// accumulators take most of the register file), two at 4-5; rows cut into as many slabs as workgroups fit at once.
#include "common.h"
#include <algorithm>
#include <stdlib.h>

namespace dlsa {

// gram.hip
template <typename T>
void gram_reduce_launch(const T* partial, int nslab, int PP, int p, T* H, int64_t ldh, int accumulate, hipStream_t stream);

#ifndef DLSA_NARROW_KC
#define DLSA_NARROW_KC 32
#endif
constexpr int NARROW_KC = DLSA_NARROW_KC;         // rows per chunk: a multiple of 16 (KC/16 k-steps per wave)
constexpr int NARROW_STAGES = 3;
constexpr int NARROW_MIN_P = 49, NARROW_MAX_P = 112;      // 4..7 tiles: 8 tiles (288 accumulator registers) spill
constexpr int64_t NARROW_MIN_ROWS = 8192;

struct NarrowArgs {
    const double* X;
    const double* w;
    double* partial;      // [nslab][PP][PP]
    int64_t ldx, n, rows_per_slab;
    int p, PP;
    int dbg;              // DLSA_GRAM_DBG (timing experiments only, wrong results): 1 = no DMA after the prologue, 128 = no MFMAs
};

// LDS row pitch in elements: a k-step's fragment read is 4 rows x 16 columns of 8 bytes; 32 lanes (two rows) are
// served per cycle, so consecutive rows must sit 128 bytes apart modulo 256: pitch = 16 mod 32.
constexpr int narrow_pitch(int nt) { return (nt % 2) ? nt * 16 : nt * 16 + 16; }
constexpr int narrow_buf_elems(int nt) { return NARROW_KC * narrow_pitch(nt) + NARROW_KC; }     // chunk + its w
// The MFMA block is inline assembly on explicitly numbered AGPRs (generated: tools/gen_gram_narrow_asm.py).  With the
// builtin -- or with "+a"-constrained operands -- hipcc carries the 224 accumulator registers of the loop in VGPRs
// and copies all of them into AGPRs and back around every k-step (448 v_accvgpr moves per 28 MFMAs: measured no
// faster than the tile-list kernel).  Named registers stay where the MFMAs want them.
#include "gram_narrow_asm.inc"

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

// Workgroups per CU: up to 5 tiles per side (120 AGPRs, 62 KB of LDS) two fit, and one's LDS waits, barriers and DMA
// issue hide under the other's MFMAs (p = 50: 1.22 -> 1.06 ms per 1e7 rows); 6 and 7 tiles take a CU alone.  (Splitting
// the 28 tiles of p = 100 over wave PAIRS -- eight waves, 112 AGPRs each, two per SIMD -- was built and measured: 2.53 vs
// 2.55 ms, no gain: with the DMA off the kernel runs 2.25 ms either way, i.e. the MFMA pipe is already ~90 % busy.)
constexpr int narrow_wgs_per_cu(int nt) { return nt <= 5 ? 2 : 1; }
constexpr int narrow_meetn(int nt) {      // tiles per meeting pass: three parked copies must fit in the ring
    const int ntri = nt * (nt + 1) / 2, fit = (int)((size_t)NARROW_STAGES * narrow_buf_elems(nt) * 8 / (3 * 2048));
    return fit < ntri ? fit : ntri;
}
constexpr size_t narrow_lds_bytes(int nt) { return (size_t)NARROW_STAGES * narrow_buf_elems(nt) * 8; }

template <bool HASW, int NT>
__global__ __launch_bounds__(256, narrow_wgs_per_cu(NT)) void gram_narrow_kernel(NarrowArgs a) {
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    constexpr int NWAVES = 4, THREADS = 64 * NWAVES;
    constexpr int KC = NARROW_KC, LDP = narrow_pitch(NT), BUF = narrow_buf_elems(NT), NTRI = NT * (NT + 1) / 2;
    constexpr int MEETN = narrow_meetn(NT);
    constexpr int DMA_PER_CHUNK = KC / NWAVES + (HASW ? 1 : 0);        // instructions per wave and chunk
    static_assert(KC % 16 == 0 && 3 * MEETN >= NTRI, "chunk / meeting shape");
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int slab = blockIdx.x;
    const int64_t rbeg = (int64_t)slab * a.rows_per_slab;
    const int64_t rend = min(rbeg + a.rows_per_slab, a.n);
    const int64_t nrows = rend > rbeg ? rend - rbeg : 0;
    const int nchunks = (int)((nrows + KC - 1) / KC);

    // rows past the slab end read as zeros through the buffer descriptors; columns p .. 16 NT - 1 are never written
    // by the DMA (lanes masked), so the ring is zeroed once
    const unsigned xbytes = nrows > 0 ? (unsigned)(((nrows - 1) * a.ldx + a.p) * 8) : 0u;
    __amdgpu_buffer_rsrc_t rsrcX = __builtin_amdgcn_make_buffer_rsrc((void*)(a.X + rbeg * a.ldx), 0, (int)xbytes, 0x00020000);
    __amdgpu_buffer_rsrc_t rsrcW =
        __builtin_amdgcn_make_buffer_rsrc((void*)(HASW ? a.w + rbeg : a.X), 0, HASW ? (int)(nrows * 8) : 0, 0x00020000);
    for (int e = tid; e < NARROW_STAGES * BUF; e += THREADS) lds[e] = 0.0;
    __syncthreads();

    const bool col_in = 2 * lane < a.p;                 // a.p is even: both columns of the lane's 16 bytes are loaded
    auto stage = [&](int chunk, int buf) {
        double* base = lds + buf * BUF;
#pragma unroll
        for (int ps = 0; ps < KC / NWAVES; ++ps) {
            const int row = wave + NWAVES * ps;
            const int soff = (int)(((int64_t)chunk * KC + row) * a.ldx * 8);
            if (col_in) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcX, (lds_ptr_t)(base + row * LDP), 16, lane * 16, soff, 0, 0);
        }
        // every wave fetches the chunk's w (same bytes to the same place): the in-order count is then the same in all waves
        if (HASW && lane < KC / 2)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcW, (lds_ptr_t)(base + KC * LDP), 16, lane * 16, chunk * KC * 8, 0, 0);
    };

    narrow_acc_zero<NTRI>();

    stage(0, 0);
    stage(1, 1);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DMA_PER_CHUNK) : "memory");
    asm volatile("s_barrier" ::: "memory");

    const int frag_off = (lane >> 4) * LDP + (lane & 15);
    int cur = 0, nxt2 = 2;                               // ring positions of chunk c and chunk c + 2
    for (int c = 0; c < nchunks; ++c) {
        if (!DLSA_DBG_WRONG(a.dbg, 1)) stage(c + 2, nxt2);            // past the slab end: bounds-checked zeros, no traffic
        const double* base = lds + cur * BUF;
        // all of this wave's fragments of the chunk are requested up front: only the first k-step waits for LDS
        double f[KC / 16][NT], wv[KC / 16];
#pragma unroll
        for (int kk = 0; kk < KC / 16; ++kk) {
            const int ks = wave + 4 * kk;
            const double* kb = base + ks * 4 * LDP + frag_off;
#pragma unroll
            for (int t = 0; t < NT; ++t) f[kk][t] = kb[t * 16];
            if (HASW) wv[kk] = base[KC * LDP + ks * 4 + (lane >> 4)];
        }
#pragma unroll
        for (int kk = 0; kk < KC / 16; ++kk) {
            double g[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) g[t] = HASW ? f[kk][t] * wv[kk] : f[kk][t];
            if (!DLSA_DBG_WRONG(a.dbg, 128)) narrow_kstep<NT>(f[kk], g);      // tile (ti, tj) += f[ti] (x) g[tj] for all ti <= tj
            else asm volatile("" ::"v"(g[0]), "v"(g[NT - 1]));
        }
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DMA_PER_CHUNK) : "memory");      // chunk c + 1 has landed
        asm volatile("s_barrier" ::: "memory");
        cur = (cur == NARROW_STAGES - 1) ? 0 : cur + 1;
        nxt2 = (nxt2 == NARROW_STAGES - 1) ? 0 : nxt2 + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the zero-fill DMA of the chunks past the end
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // the last MFMAs retire before the accumulators are read (the
    __syncthreads();                                     // compiler cannot see the hazard of an inline-asm MFMA)

    // the four waves' triangles meet in LDS, MEETN tiles at a time; wave 0 stores the slab's partial
    double* __restrict__ P = a.partial + (int64_t)slab * a.PP * a.PP;
    narrow_meet<0, (MEETN < NTRI ? MEETN : NTRI), MEETN>(lds, wave, lane, P, a.PP);
    narrow_meet<MEETN, (2 * MEETN < NTRI ? 2 * MEETN : NTRI), MEETN>(lds, wave, lane, P, a.PP);
    narrow_meet<2 * MEETN, NTRI, MEETN>(lds, wave, lane, P, a.PP);
}

static int narrow_slabs(int64_t n, int p, int64_t& rows_per_slab) {
    int64_t ns = std::min<int64_t>((int64_t)kNumCU * narrow_wgs_per_cu((p + 15) / 16), std::max<int64_t>(1, n / 1024));
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
    int64_t rps;
    const int ns = narrow_slabs(n, p, rps);
    const size_t PP = ((size_t)(p + 15) / 16 * 16 + 63) / 64 * 64;
    return align_up((size_t)ns * PP * PP * 8, 256);
}

int gram_narrow_f64(const double* X, int64_t ldx, const double* w, int64_t n, int p, double* H, int64_t ldh,
                    int accumulate, void* ws, size_t ws_bytes, hipStream_t stream) {
    NarrowArgs a;
    // odd p (in an even row pitch, checked by gram_narrow_eligible): load p + 1 columns; the pad column only reaches row
    // and column p of the tile grid, which nobody reads (see gram_impl)
    a.X = X; a.w = w; a.partial = (double*)ws; a.ldx = ldx; a.n = n; a.p = p + (p & 1);
    a.dbg = gram_dbg_env();
    const int nt = (p + 15) / 16;
    a.PP = (nt * 16 + 63) / 64 * 64;
    const int nslab = narrow_slabs(n, p, a.rows_per_slab);
    const size_t need = (size_t)nslab * a.PP * a.PP * 8;
    if (!ws || ws_bytes < need || ((uintptr_t)ws & 255)) {
        set_error("gram: workspace %zu bytes needed (256-aligned), got %zu", need, ws_bytes);
        return DLSA_ERR_WORKSPACE;
    }
#define DLSA_LAUNCH_NARROW(HW, NTV) do { \
        const size_t shm = narrow_lds_bytes(NTV); \
        if (shm > 48 * 1024) DLSA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(gram_narrow_kernel<HW, NTV>), \
                                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm)); \
        hipLaunchKernelGGL((gram_narrow_kernel<HW, NTV>), dim3(nslab), dim3(256), shm, stream, a); } while (0)
#define DLSA_LAUNCH_NARROW_NT(HW) do { switch (nt) { \
        case 4: DLSA_LAUNCH_NARROW(HW, 4); break; case 5: DLSA_LAUNCH_NARROW(HW, 5); break; \
        case 6: DLSA_LAUNCH_NARROW(HW, 6); break; default: DLSA_LAUNCH_NARROW(HW, 7); break; } } while (0)
    if (w) DLSA_LAUNCH_NARROW_NT(true);
    else DLSA_LAUNCH_NARROW_NT(false);
#undef DLSA_LAUNCH_NARROW_NT
#undef DLSA_LAUNCH_NARROW
    DLSA_HIP_CHECK(hipGetLastError());
    gram_reduce_launch<double>((const double*)ws, nslab, a.PP, p, H, ldh, accumulate, stream);
    DLSA_HIP_CHECK(hipGetLastError());
    return DLSA_OK;
}

}  // namespace dlsa
