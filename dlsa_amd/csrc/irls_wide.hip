// Wide Newton pass (121 <= p + intercept <= 512 columns): the logit pass of a wide design -- w, g = X'(y - mu), loglik in fp64, exactly
// as logit.hip computes them (dlsa/models.py:110-114) -- that ALSO yields a REDUCED-PRECISION copy of the partition's own Hessian
// H~ = X' diag(w) X  (models.py:130) from the bf16 matrix cores:
//     S = sqrt(w) . [X | 1]  rounded to bf16,   H~ = S'S  accumulated in fp32 (v_mfma_f32_32x32x16_bf16), slab sums in fp64.
// H~ is NEVER returned to a caller: it is the preconditioner of the Newton iteration (irls.hip: the step solves H~ delta = g).  The
// fixed point of that iteration is g = 0; the gradient, the stopping rule, the log-likelihood safeguard and the Sig_inv the fit
// returns stay fp64 (the closing pass is the fp64 Gram kernel), so the MLE and every golden are untouched -- only the number of
// passes changes: the partition's OWN curvature makes the iteration contract quadratically (5e-2, 1.5e-3, 2e-6, 1.6e-10, 2e-14 at the
// 1e6 x 500 partitions of config 3; the rounding moves the spectrum of H^-1 H~ by 1.5e-4) where another sample's factor contracts by
// ~0.05 per pass (nine to ten passes).
//
// Two launches.  (1) logit.hip's pass in its IMG form: next to w / g / loglik it stores S as bf16 CHUNK IMAGES of 16 rows (the K of
// one MFMA) in the layout the MFMA fragments read: [k-group of 8 rows][column block of 32][half of 4 rows][32 columns][4 rows x 2 B]
// -- nb KiB per 16 rows, 25 % of the fp64 bytes.  (2) wide_syrk_kernel: H~ = S'S from the images.  A chunk costs one MFMA per tile
// of the block triangle, spread over the 8 waves of two workgroups (rows of tiles in pairs, below).
// (The one-launch form of round 5 -- rows through an LDS ring, conversion and MFMAs in the logit kernel itself, two workgroups per
// slab -- is kept as bench/experiments/irls_wide_fused_r05.hip.txt: correct, 1.29 ms per 1e6 x 500 against 0.62 for the logit pass;
// one wave per SIMD (272 accumulator registers) cannot hide its own latencies.  docs/lab_notes_r05.md.)
#include "common.h"
#include <algorithm>

#ifndef DLSA_IMG_F16
#define DLSA_IMG_F16 0
#endif
#if DLSA_IMG_F16
#define WIDE_MFMA "v_mfma_f32_32x32x16_f16"
#else
#define WIDE_MFMA "v_mfma_f32_32x32x16_bf16"
#endif
namespace dlsa {

typedef __bf16 wide_bf16x8 __attribute__((ext_vector_type(8)));
typedef float wide_f16v __attribute__((ext_vector_type(16)));
typedef unsigned wide_u4v __attribute__((ext_vector_type(4)));
typedef unsigned wide_u2v __attribute__((ext_vector_type(2)));

constexpr int WIDE_WGS = 256;             // workgroups per launch (one per CU)
constexpr int WIDE_RING_BYTES = 160 * 1024;      // the whole LDS of a CU: what bounds the syrk is the bytes its ring keeps in flight
constexpr int64_t WIDE_MIN_ROWS = 32768;

// Tiles in CYCLIC CLASSES: block row r owns the tiles (r, (r + d) mod NB), d = 0 .. NB / 2 (the last class only for r < NB / 2 when
// NB is even) -- every unordered pair of column blocks exactly once, a tile below the diagonal standing for its transpose.  Wave gw of
// NWV = ceil(NB / 2) owns rows gw and gw + NWV: NB + 1 tiles (17 at 16 column blocks: 8 waves = HV = 2 workgroups, blockIdx b and b + 8
// -- the same XCD -- streaming the same row slab).  Walking the column blocks t = 0 .. from its own (block (gw + t) mod NB), a wave
// meets the first row's classes at t = 0 .. D0 and the second row's at t = NWV .. NWV + D1: one B fragment per step feeds one or two
// MFMAs whose accumulators are compile-time constants -- no select, no branch (hipcc copies accumulators at every join).  Per chunk a
// wave reads 2 + NB fragments of 1 KiB for NB + 1 MFMAs of 32 cycles: the LDS (128 B / clock for four SIMDs) stays under the matrix pipe.
__host__ __device__ constexpr int wide_nwave(int nb) { return (nb + 1) / 2; }
__host__ __device__ constexpr int wide_hv(int nb) { return (wide_nwave(nb) + 3) / 4; }
__host__ __device__ constexpr int wide_groups(int nb) { return WIDE_WGS / wide_hv(nb); }
__host__ __device__ constexpr int wide_d0(int nb) { return nb / 2; }                           // classes 0 .. d0 of a wave's first row (gw < NB / 2 always)
__host__ __device__ constexpr int wide_d1(int nb) { return (nb % 2) ? nb / 2 : nb / 2 - 1; }   // ... of its second row (gw + NWV >= NB / 2 always)
__host__ __device__ constexpr int wide_slots(int nb) { return wide_d0(nb) + wide_d1(nb) + 2; }

template <int NB>
__global__ __launch_bounds__(256, 1) void wide_syrk_kernel(const unsigned* __restrict__ img, int64_t nchunks, float* __restrict__ hpart) {
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    constexpr int PW = (NB + 3) / 4;                              // 1 KiB pieces per wave and chunk
    constexpr int STG = 4 * PW * 1024, CH = NB * 1024;            // stage pitch, image bytes per chunk
    constexpr int NS = (WIDE_RING_BYTES / STG) < 16 ? (WIDE_RING_BYTES / STG) : 16, D = NS - 2;      // ring stages; chunks in flight (two per trip)
    constexpr int HV = wide_hv(NB), NWV = wide_nwave(NB), D0 = wide_d0(NB), D1 = wide_d1(NB), T = (NB % 2) ? NB + 1 : NB;
    static_assert(D >= 2 && (D - 2) * PW <= 63 && NS * STG <= WIDE_RING_BYTES, "vmcnt immediate; ring fits");
    static_assert(NWV + D1 < T && D0 < T, "the second row's classes end within the walk");
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bid = blockIdx.x;
    const int grp = (bid / (8 * HV)) * 8 + (bid % 8), half = (bid / 8) % HV, ngrp = gridDim.x / HV;
    const int gw0 = half * 4 + wave;                              // (>= NWV: no tiles -- the wave only stages; it multiplies wave 0's and stores nothing)
    const int gw = gw0 < NWV ? gw0 : 0;
    const int r0 = gw, r1 = min(gw + NWV, NB - 1);                // (odd NB: the last wave has one row; its second run repeats tiles nobody sums)
    const int lo = (lane >> 5) * (NB * 512) + (lane & 31) * 8;
    const int offA0 = lo + r0 * 512, offA1 = lo + r1 * 512;
    int offB[T];
#pragma unroll
    for (int t = 0; t < T; ++t) offB[t] = lo + ((gw + t) % NB) * 512;
    wide_f16v acc0[D0 + 1], acc1[D1 + 1];
#pragma unroll
    for (int k = 0; k <= D0; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc0[k][r] = 0.0f;
#pragma unroll
    for (int k = 0; k <= D1; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc1[k][r] = 0.0f;
    const int nj = nchunks > grp ? (int)((nchunks - grp + ngrp - 1) / ngrp) : 0;
    auto issue = [&](int j) {
        const int64_t c = grp + (int64_t)j * ngrp;
        const bool live = c < nchunks;                            // (past the end: an empty range -- the DMA stores zeros, the MFMAs add nothing)
        __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(img + (live ? c : 0) * (CH / 4)), 0, live ? CH : 0, 0x00020000);
        char* dst = lds + (j % NS) * STG;
#pragma unroll
        for (int i = 0; i < PW; ++i) {
            const int pz = wave + 4 * i;                          // (a piece past the image is out of the descriptor's range: zeros)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(dst + pz * 1024), 16, lane * 16 + pz * 1024, 0, 0, 0);
        }
    };
    auto frag = [&](const char* st, int off) -> wide_bf16x8 {
        const wide_u2v a0 = *reinterpret_cast<const wide_u2v*>(st + off), a1 = *reinterpret_cast<const wide_u2v*>(st + off + 256);
        return __builtin_bit_cast(wide_bf16x8, wide_u4v{a0.x, a0.y, a1.x, a1.y});
    };
    auto chunk = [&](const char* st) {
        const wide_bf16x8 a0 = frag(st, offA0), a1 = frag(st, offA1);
        wide_bf16x8 b[3];                                         // fragments two steps ahead of their MFMAs
        b[0] = frag(st, offB[0]);
        b[1] = frag(st, offB[1]);
#pragma unroll
        for (int t = 0; t < T; ++t) {
            if (t + 2 < T) b[(t + 2) % 3] = frag(st, offB[t + 2]);
            // (the MFMAs as asm statements with the accumulator TIED to an AGPR tuple -- the last of 17 to VGPRs, there are 256 AGPRs: as
            // builtins hipcc rotates the loop-carried accumulators through v_accvgpr_read / _write every trip)
            if (t <= D0) asm volatile(WIDE_MFMA " %0, %1, %2, %0" : "+a"(acc0[t]) : "v"(a0), "v"(b[t % 3]));
            if (t >= NWV && t <= NWV + D1) {
                if (D0 + 1 + t - NWV < 16) asm volatile(WIDE_MFMA " %0, %1, %2, %0" : "+a"(acc1[t - NWV]) : "v"(a1), "v"(b[t % 3]));
                else asm volatile(WIDE_MFMA " %0, %1, %2, %0" : "+v"(acc1[t - NWV]) : "v"(a1), "v"(b[t % 3]));
            }
        }
    };
    if (nj > 0) {
#pragma unroll
        for (int j = 0; j < D; ++j) issue(j);
    }
    for (int j = 0; j < nj; j += 2) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 2) * PW) : "memory");       // this wave's pieces of chunks j, j + 1 have landed
        asm volatile("s_barrier" ::: "memory");                                      // ... everybody's; and everybody has left chunks j - 2, j - 1
        issue(j + D);                                                                // (into stages nobody reads any more)
        issue(j + D + 1);
        chunk(lds + (j % NS) * STG);
        chunk(lds + ((j + 1) % NS) * STG);
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");        // (the last MFMAs' results: the asm statements hide their latency from hipcc)
    if (gw0 < NWV) {
        float* hp = hpart + ((int64_t)grp * NWV + gw) * (wide_slots(NB) * 1024) + lane;     // [group][wave][slot: first row's classes, second row's][register][lane]
#pragma unroll
        for (int k = 0; k <= D0; ++k)
#pragma unroll
            for (int r = 0; r < 16; ++r) hp[k * 1024 + r * 64] = acc0[k][r];
        if (gw + NWV < NB) {
#pragma unroll
            for (int k = 0; k <= D1; ++k)
#pragma unroll
                for (int r = 0; r < 16; ++r) hp[(D0 + 1 + k) * 1024 + r * 64] = acc1[k][r];
        }
    }
}

// H~[r][c] = sum over the groups of the tile partials, fixed order, in fp64; both triangles; with the intercept the design's last
// column goes FIRST (models.py:121-122,136-142).  One thread per accumulator element of every stored slot.
__global__ __launch_bounds__(256) void wide_reduce_kernel(const float* __restrict__ hpart, int ngrp, int nb, int pe, int icpt,
                                                          double* __restrict__ H, int64_t ldh) {
    const int nwave = wide_nwave(nb), slots = wide_slots(nb), d0 = wide_d0(nb);
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= nwave * slots * 1024) return;
    const int gw = e / (slots * 1024), slot = (e >> 10) % slots, reg = (e >> 6) & 15, lane = e & 63;
    const int row = slot <= d0 ? gw : gw + nwave, d = slot <= d0 ? slot : slot - d0 - 1;
    if (row >= nb) return;                               // (odd NB: the last wave has no second row)
    const int colb = (row + d) % nb;
    // the tile is block (row, colb) of H~; below the diagonal it stands for its transpose
    const int I = row, J = colb;
    if (I >= nb || J >= nb) return;                      // (padding tiles)
    const int n = lane & 31, m = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
    const int r = I * 32 + m, c = J * 32 + n;            // (r > c for a tile below the diagonal: the two stores below fill both triangles either way)
    if (r >= pe || c >= pe) return;
    if (I == J && m > n) return;                         // (a diagonal tile holds both triangles: its upper one is written, and mirrored)
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    const float* src = hpart + e;
    const int64_t stride = (int64_t)nwave * slots * 1024;
    int q = 0;
    for (; q + 3 < ngrp; q += 4) {
        s0 += (double)src[(int64_t)q * stride];
        s1 += (double)src[(int64_t)(q + 1) * stride];
        s2 += (double)src[(int64_t)(q + 2) * stride];
        s3 += (double)src[(int64_t)(q + 3) * stride];
    }
    for (; q < ngrp; ++q) s0 += (double)src[(int64_t)q * stride];
    const double v = (s0 + s1) + (s2 + s3);
    const int ro = icpt ? (r == pe - 1 ? 0 : r + 1) : r, co = icpt ? (c == pe - 1 ? 0 : c + 1) : c;
    H[(int64_t)ro * ldh + co] = v;
    H[(int64_t)co * ldh + ro] = v;
}

size_t logit_workspace_bytes_impl(int64_t n, int p);
int logit_pass_image_impl(const double* X, int64_t ldx, const double* y, const double* beta, int64_t n, int p, double* w_out, double* g,
                          double* loglik, void* ws, size_t ws_bytes, hipStream_t stream, int intercept, unsigned* img, int img_nb);

static int wide_nb(int pe) { return (pe + 31) / 32; }

bool irls_wide_eligible(const double* X, int64_t ldx, int64_t n, int p, int icpt) {
    const int pe = p + (icpt ? 1 : 0);
    (void)X;
    return pe >= 121 && pe <= 512 && p <= 512 && n >= WIDE_MIN_ROWS && ldx >= p;
}

static size_t wide_hpart_bytes(int nb) { return (size_t)wide_groups(nb) * wide_nwave(nb) * wide_slots(nb) * 1024 * sizeof(float); }

// (n rows: the chunk images, nb KiB per 16 rows)
size_t irls_wide_workspace_bytes(int64_t n, int p, int icpt) {
    const int pe = p + (icpt ? 1 : 0);
    if (pe < 121 || pe > 512 || n < 0) return 0;
    const int nb = wide_nb(pe);
    return align_up(logit_workspace_bytes_impl(n, pe), 256) + align_up((size_t)((n + 15) / 16) * nb * 1024, 256) + align_up(wide_hpart_bytes(nb), 256);
}

template <int NB>
static int wide_syrk_launch(const unsigned* img, int64_t nchunks, float* hpart, hipStream_t s) {
    constexpr int PW = (NB + 3) / 4, STG = 4 * PW * 1024, NS = (WIDE_RING_BYTES / STG) < 16 ? (WIDE_RING_BYTES / STG) : 16;
    constexpr size_t shm = (size_t)NS * STG;
    // (set on every launch, as the other large-LDS kernels do: the attribute belongs to the current device's function object, and
    // chain threads launch concurrently)
    DLSA_HIP_CHECK(hipFuncSetAttribute((const void*)wide_syrk_kernel<NB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
    hipLaunchKernelGGL((wide_syrk_kernel<NB>), dim3(WIDE_WGS), dim3(256), shm, s, img, nchunks, hpart);
    return DLSA_OK;
}

// g, loglik (and w) of the logit pass + the approximate Hessian Happrox (pe x pe, fp64 storage, both triangles, intercept first).
int irls_wide_pass_impl(const double* X, int64_t ldx, const double* y, const double* beta, int64_t n, int p, int icpt, double* w_out,
                        double* g, double* loglik, double* Happrox, int64_t ldh, void* ws, size_t ws_bytes, hipStream_t stream) {
    DLSA_REQUIRE(X && y && beta && g && Happrox, "wide pass: null argument");
    DLSA_REQUIRE(irls_wide_eligible(X, ldx, n, p, icpt), "wide pass: shape not served (121 <= p + intercept <= 512, >= %lld rows)", (long long)WIDE_MIN_ROWS);
    const int pe = p + (icpt ? 1 : 0);
    DLSA_REQUIRE(ldh >= pe, "wide pass: ldh < p + intercept");
    if (!ws || ws_bytes < irls_wide_workspace_bytes(n, p, icpt) || ((uintptr_t)ws & 255)) {
        set_error("wide pass: workspace %zu bytes needed (256-aligned), got %zu", irls_wide_workspace_bytes(n, p, icpt), ws_bytes);
        return DLSA_ERR_WORKSPACE;
    }
    const int nb = wide_nb(pe);
    Arena ar(ws, ws_bytes);
    const size_t lb = logit_workspace_bytes_impl(n, pe);
    void* lws = ar.take(lb);
    const int64_t nchunks = (n + 15) / 16;
    unsigned* img = (unsigned*)ar.take((size_t)nchunks * nb * 1024);
    float* hpart = (float*)ar.take(wide_hpart_bytes(nb));
    int rc = logit_pass_image_impl(X, ldx, y, beta, n, p, w_out, g, loglik, lws, lb, stream, icpt ? 1 : 0, img, nb);
    if (rc) return rc;
    rc = DLSA_ERR_INVALID;
    switch (nb) {
#define S(NB_) case NB_: rc = wide_syrk_launch<NB_>(img, nchunks, hpart, stream); break;
        S(4) S(5) S(6) S(7) S(8) S(9) S(10) S(11) S(12) S(13) S(14) S(15) S(16)
#undef S
        default: break;
    }
    if (rc) return rc;
    DLSA_HIP_CHECK(hipGetLastError());
    hipLaunchKernelGGL(wide_reduce_kernel, dim3((wide_nwave(nb) * wide_slots(nb) * 1024 + 255) / 256), dim3(256), 0, stream, (const float*)hpart, wide_groups(nb), nb, pe,
                       icpt ? 1 : 0, Happrox, ldh);
    DLSA_HIP_CHECK(hipGetLastError());
    return DLSA_OK;
}

}  // namespace dlsa

extern "C" {

size_t dlsa_newton_wide_workspace_bytes(int64_t n, int p, int intercept) { return dlsa::irls_wide_workspace_bytes(n, p, intercept); }

int dlsa_newton_wide_eligible(const double* X, int64_t ldx, int64_t n, int p, int intercept) {
    return dlsa::irls_wide_eligible(X, ldx, n, p, intercept) ? 1 : 0;
}

int dlsa_newton_wide_pass_f64(const double* X, int64_t ldx, const double* y, const double* beta, int64_t n, int p, int intercept,
                              double* w_out, double* g, double* loglik, double* H_approx, int64_t ldh, void* ws, size_t ws_bytes,
                              void* stream) {
    return dlsa::irls_wide_pass_impl(X, ldx, y, beta, n, p, intercept, w_out, g, loglik, H_approx, ldh, ws, ws_bytes, (hipStream_t)stream);
}

}  // extern "C"
