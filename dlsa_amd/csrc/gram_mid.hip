// Weighted Gram H = X' diag(w) X for MID-WIDTH fp64 designs: 8 .. 17 full 16-column tiles (+ up to three 4-column tail
// groups), 125 <= p <= 284 -- BASELINE config 4's dense width (p ~ 250-260).  Reference call site: dlsa/models.py:130.
//
// gram.hip serves these widths with its tile-list plan (one fragment pair from LDS per MFMA: 86 instead of 64 pipe cycles
// per tile-step, 38.5 TF at p = 260) because its 4 x 4 wave blocks waste too many slots on ragged widths.  Here the idea of
// gram_narrow.hip -- the accumulators of the WHOLE upper triangle live in AGPRs, rows stream past them -- is carried over
// to widths whose triangle no longer fits one wave: the <= 153 tiles are dealt to the 8 waves of a workgroup (two per SIMD,
// <= 20 tiles = 160 AGPRs each) by a generated, balanced plan (tools/gen_gram_mid_asm.py: bands of 4 tile rows walked
// column by column and cut into 8 equal runs).  Every wave role has its own static tile list, so the fragment reads are
// `lane address + immediate`, the MFMAs name their AGPRs, the weight multiplies the side with fewer fragments and the
// epilogue stores each tile where it belongs; the remainder of p modulo 16 is covered by 4-column tail groups on
// v_mfma_f64_4x4x4_4b_f64 instead of a padded tile column (gram_narrow.hip).
// One workgroup per CU and slab streams the slab's full rows through a 4-stage LDS-DMA ring (8-row chunks, three chunks
// ahead, gram_cyclic.hip's pipeline); no workgroup shares rows with another, so there is nothing to keep in lock step.
#include "common.h"
#include <algorithm>

namespace dlsa {

template <typename T>
void gram_reduce_launch(const T* partial, int nslab, int PP, int p, T* H, int64_t ldh, int accumulate, hipStream_t stream);   // gram.hip

constexpr int MID_NT_MIN = 8, MID_NT_MAX = 17;
constexpr int MID_KC = 8, MID_NST = 4;
constexpr int64_t MID_MIN_ROWS = 32768;

struct MidArgs {
    const double* X;
    const double* w;
    double* partial;      // [nslab][PP][PP]
    int64_t ldx, n, rows_per_slab;
    int p;                // columns loaded (even)
    int PP;
};

#include "gram_mid_asm.inc"

// LDS row pitch in doubles: >= 16 tile columns, = 16 mod 32 (two rows per ds_read_b64 lane group on distinct banks)
constexpr int mid_pitch(int ntc) { return (ntc % 2) ? 16 * ntc : 16 * ntc + 16; }
constexpr int mid_buf(int ntc) { return MID_KC * mid_pitch(ntc) + MID_KC; }          // a chunk + its w

template <bool HASW, int NT, int G, int W>
__device__ __forceinline__ void gram_mid_wave(const MidArgs& a, double* lds, int lane, int slab) {
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    typedef MidPlan<NT, G> Plan;
    constexpr int NTC = NT + (G > 0 ? 1 : 0), LDP = mid_pitch(NTC), BUF = mid_buf(NTC), GA = G > 0 ? G : 1, KC = MID_KC;
    constexpr int NPQ = (LDP * 8 + 1023) / 1024;                  // 1 KB DMA pieces per row (the last one masked to the pitch)
    constexpr int DMA_PER_CHUNK = NPQ + (HASW ? 1 : 0);           // per wave: wave W fetches row W of a chunk
    const int64_t rbeg = (int64_t)slab * a.rows_per_slab;
    const int64_t rend = min(rbeg + a.rows_per_slab, a.n);
    const int64_t nrows = rend > rbeg ? rend - rbeg : 0;
    const int nchunks = (int)((nrows + KC - 1) / KC);

    // The DMA copies whole 1 KB pieces of every row up to the LDS pitch, whatever p: columns p .. of the LDS rows hold whatever
    // follows the row in memory (zeros past the slab end through the descriptor's bounds check).  Harmless: an MFMA output
    // element depends on one column of A and one of B, so those columns only reach rows / columns >= p, which nobody reads.
    const unsigned xbytes = nrows > 0 ? (unsigned)(((nrows - 1) * a.ldx + a.p) * 8) : 0u;
    __amdgpu_buffer_rsrc_t rsrcX = __builtin_amdgcn_make_buffer_rsrc((void*)(a.X + rbeg * a.ldx), 0, (int)xbytes, 0x00020000);
    __amdgpu_buffer_rsrc_t rsrcW =
        __builtin_amdgcn_make_buffer_rsrc((void*)(HASW ? a.w + rbeg : a.X), 0, HASW ? (int)(nrows * 8) : 0, 0x00020000);
    auto dma_chunk = [&](int chunk, int buf) {
        const int soff0 = (int)(((int64_t)chunk * KC + W) * a.ldx * 8);
#pragma unroll
        for (int q = 0; q < NPQ; ++q) {
            if (128 * q + 2 * lane + 1 < LDP)                     // constant-folded for every piece but the last
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcX, (lds_ptr_t)(lds + buf * BUF + W * LDP + 128 * q), 16, lane * 16, soff0 + 1024 * q, 0, 0);
        }
        if (HASW && lane < KC / 2)       // every wave fetches the chunk's w: same in-order count in all waves
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcW, (lds_ptr_t)(lds + buf * BUF + KC * LDP), 16, lane * 16, chunk * KC * 8, 0, 0);
    };

    mid_acc_zero<Plan::NREG>();
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) dma_chunk(ch, ch);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * DMA_PER_CHUNK) : "memory");
    asm volatile("s_barrier" ::: "memory");

    const char* ldsb = (const char*)lds;
    const int lane_addr = ((lane >> 4) * LDP + (lane & 15)) * 8;           // fragment: row (lane >> 4) of a k-step, column 16 t + (lane & 15)
    const int tail_addr = ((lane >> 4) * LDP + 16 * NT + (lane & 3)) * 8;  // tail columns, broadcast to the 4 blocks
    const int w_addr = (KC * LDP + (lane >> 4)) * 8;
    struct Frag { double fa[Plan::MAXA], fb[Plan::MAXB], bt[GA], wv; };
    auto load_frags = [&](int st, int ks, Frag& f) {
        const int base = (st * BUF + ks * 4 * LDP) * 8;
#pragma unroll
        for (int i = 0; i < Plan::MAXA; ++i) f.fa[i] = 0.0;
#pragma unroll
        for (int j = 0; j < Plan::MAXB; ++j) f.fb[j] = 0.0;
        const int addr = lane_addr + base;
        Plan::template load<W>([&](int imm) { return *(const double*)(ldsb + addr + imm); }, f.fa, f.fb);
#pragma unroll
        for (int g = 0; g < GA; ++g) f.bt[g] = (G > 0 && Plan::template has_tails<W>()) ? *(const double*)(ldsb + tail_addr + base + 32 * g) : 0.0;
        f.wv = HASW ? *(const double*)(ldsb + w_addr + (st * BUF + ks * 4) * 8) : 1.0;
    };
    auto kstep = [&](Frag& f) {
        if (HASW) Plan::template scale<W>(f.wv, f.fa, f.fb, f.bt);
        Plan::template mfma<W>(f.fa, f.fb, f.bt);
    };

    // Pipeline (chunk c = k-steps (c, 0), (c, 1); stage = c mod 4): see gram_cyclic.hip.  Chunks past the end of the slab are
    // fetched and computed all the same (zeros through the bounds check): the vmcnt bookkeeping stays a constant.
    Frag fr0, fr1;
    load_frags(0, 0, fr0);
    int st = 0;
    for (int c = 0; c < nchunks; ++c) {
        const int st1 = (st + 1) & 3, st3 = (st + 3) & 3;
        load_frags(st, 1, fr1);                                  // (c, 1), while (c, 0) computes
        __builtin_amdgcn_sched_barrier(0);
        kstep(fr0);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DMA_PER_CHUNK) : "memory");      // chunk c + 1 has landed (c + 2 may be in flight)
        asm volatile("s_barrier" ::: "memory");
        load_frags(st1, 0, fr0);                                 // (c + 1, 0), while (c, 1) computes
        dma_chunk(c + 3, st3);                                   // every wave has left chunk c - 1, whose stage this overwrites
        __builtin_amdgcn_sched_barrier(0);
        kstep(fr1);
        __builtin_amdgcn_sched_barrier(0);
        st = st1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // the zero-fill DMA of the chunks past the end
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");           // the last MFMAs retire before the accumulators are read
    Plan::template store<W>(lane, a.partial + (int64_t)slab * a.PP * a.PP, a.PP);
}

template <bool HASW, int NT, int G>
__global__ __launch_bounds__(512, 2) void gram_mid_kernel(MidArgs a) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int slab = blockIdx.x;
    switch (wave) {          // every wave role has its own static tile plan: eight copies of the loop, no dispatch inside it
        case 0: gram_mid_wave<HASW, NT, G, 0>(a, lds, lane, slab); break;
        case 1: gram_mid_wave<HASW, NT, G, 1>(a, lds, lane, slab); break;
        case 2: gram_mid_wave<HASW, NT, G, 2>(a, lds, lane, slab); break;
        case 3: gram_mid_wave<HASW, NT, G, 3>(a, lds, lane, slab); break;
        case 4: gram_mid_wave<HASW, NT, G, 4>(a, lds, lane, slab); break;
        case 5: gram_mid_wave<HASW, NT, G, 5>(a, lds, lane, slab); break;
        case 6: gram_mid_wave<HASW, NT, G, 6>(a, lds, lane, slab); break;
        default: gram_mid_wave<HASW, NT, G, 7>(a, lds, lane, slab); break;
    }
}

// p columns = NT full tiles + G tail groups of 4 (a fourth group makes a full tile)
static void mid_shape(int p, int& nt, int& g) {
    nt = p / 16;
    g = (p - 16 * nt + 3) / 4;
    if (g == 4) { ++nt; g = 0; }
}

static int mid_slabs(int64_t n, int64_t& rows_per_slab) {
    int64_t ns = kNumCU;
    while (ns > kNumXCD && n / ns < 4 * MID_KC) ns -= kNumXCD;
    rows_per_slab = ((n + ns - 1) / ns + MID_KC - 1) / MID_KC * MID_KC;
    return (int)ns;
}

bool gram_mid_shape_ok(int64_t n, int p) {
    int nt, g;
    mid_shape(p + (p & 1), nt, g);
    if (nt == MID_NT_MAX && g == 3) return false;            // 178 AGPRs + 80 VGPRs: one wave per SIMD, the 8-wave workgroup would not fit
    return nt >= MID_NT_MIN && nt <= MID_NT_MAX && n >= MID_MIN_ROWS;
}

bool gram_mid_eligible(const double* X, int64_t ldx, const double* w, int64_t n, int p) {
    if (!gram_mid_shape_ok(n, p)) return false;
    if (gram_dbg_env() & 32) return false;                   // DLSA_GRAM_DBG 32: keep gram.hip's plans (valid results, A/B runs)
    int64_t rps;
    mid_slabs(n, rps);
    return ldx % 2 == 0 && ((uintptr_t)X % 16) == 0 && (!w || ((uintptr_t)w % 16) == 0) &&
           (double)(rps + 8 * MID_KC) * (double)ldx * 8.0 < 2.0e9;            // 32-bit DMA offsets
}

static int mid_pp(int p) { return ((p + 15) / 16 * 16 + 63) / 64 * 64; }

size_t gram_mid_ws_bytes(int64_t n, int p) {
    int64_t rps;
    const int ns = mid_slabs(n, rps);
    return align_up((size_t)ns * mid_pp(p) * mid_pp(p) * 8, 256);
}

template <bool HASW, int NT, int G>
static int mid_launch(const MidArgs& a, int nslab, hipStream_t stream) {
    constexpr int NTC = NT + (G > 0 ? 1 : 0);
    const size_t shm = (size_t)MID_NST * mid_buf(NTC) * 8;
    if (shm > 48 * 1024)
        DLSA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(gram_mid_kernel<HASW, NT, G>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
    hipLaunchKernelGGL((gram_mid_kernel<HASW, NT, G>), dim3(nslab), dim3(512), shm, stream, a);
    return DLSA_OK;
}

template <bool HASW, int NT>
static int mid_launch_g(const MidArgs& a, int g, int nslab, hipStream_t stream) {
    switch (g) {
        case 0: return mid_launch<HASW, NT, 0>(a, nslab, stream);
        case 1: return mid_launch<HASW, NT, 1>(a, nslab, stream);
        case 2: return mid_launch<HASW, NT, 2>(a, nslab, stream);
        default: return mid_launch<HASW, NT, 3>(a, nslab, stream);
    }
}

template <bool HASW>
static int mid_launch_nt(const MidArgs& a, int nt, int g, int nslab, hipStream_t stream) {
    switch (nt) {
        case 8: return mid_launch_g<HASW, 8>(a, g, nslab, stream);
        case 9: return mid_launch_g<HASW, 9>(a, g, nslab, stream);
        case 10: return mid_launch_g<HASW, 10>(a, g, nslab, stream);
        case 11: return mid_launch_g<HASW, 11>(a, g, nslab, stream);
        case 12: return mid_launch_g<HASW, 12>(a, g, nslab, stream);
        case 13: return mid_launch_g<HASW, 13>(a, g, nslab, stream);
        case 14: return mid_launch_g<HASW, 14>(a, g, nslab, stream);
        case 15: return mid_launch_g<HASW, 15>(a, g, nslab, stream);
        case 16: return mid_launch_g<HASW, 16>(a, g, nslab, stream);
        default: return mid_launch_g<HASW, 17>(a, g, nslab, stream);
    }
}

int gram_mid_f64(const double* X, int64_t ldx, const double* w, int64_t n, int p, double* H, int64_t ldh,
                 int accumulate, void* ws, size_t ws_bytes, hipStream_t stream) {
    MidArgs a;
    a.X = X; a.w = w; a.partial = (double*)ws; a.ldx = ldx; a.n = n;
    a.p = p + (p & 1);       // odd p in an even row pitch: the pad column only reaches row / column p of H, which nobody reads
    a.PP = mid_pp(p);
    const int nslab = mid_slabs(n, a.rows_per_slab);
    const size_t need = (size_t)nslab * a.PP * a.PP * 8;
    if (!ws || ws_bytes < need || ((uintptr_t)ws & 255)) {
        set_error("gram: workspace %zu bytes needed (256-aligned), got %zu", need, ws_bytes);
        return DLSA_ERR_WORKSPACE;
    }
    int nt, g;
    mid_shape(a.p, nt, g);
    const int rc = w ? mid_launch_nt<true>(a, nt, g, nslab, stream) : mid_launch_nt<false>(a, nt, g, nslab, stream);
    if (rc) return rc;
    DLSA_HIP_CHECK(hipGetLastError());
    gram_reduce_launch<double>((const double*)ws, nslab, a.PP, p, H, ldh, accumulate, stream);
    DLSA_HIP_CHECK(hipGetLastError());
    return DLSA_OK;
}

}  // namespace dlsa
