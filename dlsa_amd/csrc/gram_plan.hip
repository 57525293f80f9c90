// Weighted Gram H = X' diag(w) X for fp64 designs of 8 .. 35 full 16-column tiles (+ up to three 4-column tail groups),
// 125 <= p <= 572: BASELINE config 3 (p = 500), config 4's dense width (p ~ 250-260) and everything between.
// Reference call site: dlsa/models.py:130.
//
// The accumulators of the WHOLE upper triangle live in AGPRs and the rows stream past them (the idea of gram_narrow.hip);
// the NT (NT + 1) / 2 tiles are dealt to the 8 C waves of a group of C = 1, 2 or 4 workgroups (<= 20 tiles = 160 AGPRs per
// wave, two waves per SIMD) by a generated, balanced plan (tools/gen_gram_plan_asm.py: bands of 4 tile rows walked column
// by column, cut into 8 C equal runs, runs paired heaviest-with-lightest onto SIMDs).  Every wave role has its own static
// tile list, so it is compiled as its own loop: fragment reads are `lane address + immediate`, the MFMAs name their AGPRs,
// the weight multiplies the side with fewer fragments, the epilogue stores each tile where it belongs; the remainder of p
// modulo 16 is covered by 4-column tail groups on v_mfma_f64_4x4x4_4b_f64 instead of a padded tile column.
//
// LDS column layout: tiles 2g and 2g + 1 are the even and odd columns of the 32-column group g.  One ds_read_b128 then
// fetches the fragments of BOTH tiles, and a fragment feeds the A side, the B side or both: a wave with ~19 tiles issues
// ~5 LDS reads per k-step (a row-times-column block of a blocked layout needs 9-11).  That matters because every LDS read
// next to an fp64 MFMA costs the matrix pipe ~8 cycles.
//
// Each workgroup streams the slab's full rows through a 4-stage LDS-DMA ring (8-row chunks, three chunks ahead).  The C
// workgroups of a group stream the SAME rows, run on CUs of one XCD and carry the same MFMA time per SIMD (the plan pads
// the lighter ones), so they stay in lock step and the XCD's L2 fetches every row from HBM once.
#include "gram_plan.h"
#include <algorithm>

namespace dlsa {

template <typename T>
void gram_reduce_launch(const T* partial, int nslab, int PP, int p, T* H, int64_t ldh, int accumulate, hipStream_t stream);   // gram.hip

// p columns = NT full tiles + G tail groups of 4 (a fourth group makes a full tile)
static void plan_shape(int p, int& nt, int& g) {
    nt = p / 16;
    g = (p - 16 * nt + 3) / 4;
    if (g == 4) { ++nt; g = 0; }
    // 17 tiles + 3 tail groups (p = 281 .. 284) do not fit the registers of a single-CU plan at two waves per SIMD (168 accumulators
    // + 92 VGPRs): the shape runs as 18 full tiles on a two-CU group instead -- columns p .. 287 of the LDS rows hold whatever
    // follows the row, which only reaches rows / columns >= p of H (2.8 % more MFMAs than the tail groups would have cost)
    if (nt == 17 && g == 3) { nt = 18; g = 0; }
    // p = 121 .. 124 (7 tiles + 3 tail groups: 272 accumulator registers, beyond gram_narrow.hip's AGPR file) run as 8 full tiles
    if (nt == 7 && g == 3) { nt = 8; g = 0; }
}

static int plan_slabs(int64_t n, int C, int64_t& rows_per_slab) {
    int64_t ns = kNumCU / C;
    while (ns > kNumXCD && n / ns < 4 * PLAN_KC1) ns -= kNumXCD;
    rows_per_slab = ((n + ns - 1) / ns + PLAN_KC1 - 1) / PLAN_KC1 * PLAN_KC1;       // whole chunks of either plan class
    return (int)ns;
}

bool gram_plan_shape_ok(int64_t n, int p) {
    int nt, g;
    plan_shape(p + (p & 1), nt, g);
    return nt >= PLAN_NT_MIN && nt <= PLAN_NT_MAX && n >= PLAN_MIN_ROWS;
}

bool gram_plan_eligible(const double* X, int64_t ldx, const double* w, int64_t n, int p) {
    if (!gram_plan_shape_ok(n, p)) return false;
    if (gram_dbg_env() & 256) return false;                  // DLSA_GRAM_DBG 256: keep the older kernels (valid results, A/B runs)
    int nt, g;
    plan_shape(p + (p & 1), nt, g);
    int64_t rps;
    plan_slabs(n, plan_group(nt), rps);
    return ldx % 2 == 0 && ((uintptr_t)X % 16) == 0 && (!w || ((uintptr_t)w % 16) == 0) &&
           (double)(rps + 8 * PLAN_KC1) * (double)ldx * 8.0 < 2.0e9;            // 32-bit DMA offsets
}

static int plan_pp(int p) { return ((p + 15) / 16 * 16 + 63) / 64 * 64; }

size_t gram_plan_ws_bytes(int64_t n, int p) {
    int nt, g;
    plan_shape(p + (p & 1), nt, g);
    int64_t rps;
    const int ns = plan_slabs(n, plan_group(nt), rps);
    return align_up((size_t)ns * plan_pp(p) * plan_pp(p) * 8, 256) + align_up((size_t)ns * 16, 256) + 256 + kGramProbeBytes;    // partials, pacing counts, ones, clock probe
}

__global__ void plan_ones_kernel(double* ones) { ones[threadIdx.x] = 1.0; }

int gram_plan_f64(const double* X, int64_t ldx, const double* w, int64_t n, int p, double* H, int64_t ldh,
                  int accumulate, void* ws, size_t ws_bytes, hipStream_t stream) {
    PlanArgs a;
    a.X = X; a.w = w; a.partial = (double*)ws; a.ldx = ldx; a.n = n;
    a.p = p + (p & 1);       // odd p in an even row pitch: the pad column only reaches row / column p of H, which nobody reads
    a.PP = plan_pp(p);
    int nt, g;
    plan_shape(a.p, nt, g);
    const int nslab = plan_slabs(n, plan_group(nt), a.rows_per_slab);
    const size_t part = align_up((size_t)nslab * a.PP * a.PP * 8, 256), prog = align_up((size_t)nslab * 16, 256), need = part + prog + 256 + kGramProbeBytes;
    a.clk = (unsigned long long*)((char*)ws + part + prog + 256);
    if (!ws || ws_bytes < need || ((uintptr_t)ws & 255)) {
        set_error("gram: workspace %zu bytes needed (256-aligned), got %zu", need, ws_bytes);
        return DLSA_ERR_WORKSPACE;
    }
    a.progress = (int*)((char*)ws + part);
    if (plan_group(nt) > 1) DLSA_HIP_CHECK(hipMemsetAsync(a.progress, 0, (size_t)nslab * 16, stream));
    a.w_step = w ? 1 : 0;
    if (!w) {                // unweighted: the kernel multiplies by a streamed block of ones (gram_plan_kernel.inc plan_launch_g)
        double* ones = (double*)((char*)ws + part + prog);
        hipLaunchKernelGGL(plan_ones_kernel, dim3(1), dim3(PLAN_KC1), 0, stream, ones);
        a.w = ones;
    }
    int rc;
    if (nt <= 17) rc = gram_plan_launch_8(a, nt, g, nslab, stream);
    else if (nt <= 24) rc = gram_plan_launch_18(a, nt, g, nslab, stream);
    else if (nt <= 28) rc = gram_plan_launch_25(a, nt, g, nslab, stream);
    else if (nt <= 32) rc = gram_plan_launch_29(a, nt, g, nslab, stream);
    else rc = gram_plan_launch_33(a, nt, g, nslab, stream);
    if (rc) return rc;
    DLSA_HIP_CHECK(hipGetLastError());
    note_gram_kernel(a.clk, stream, "gram_plan_kernel<true,%d,%d>%s", nt, g, w ? "" : " (unweighted: a streamed block of ones)");
    gram_reduce_launch<double>((const double*)ws, nslab, a.PP, p, H, ldh, accumulate, stream);
    DLSA_HIP_CHECK(hipGetLastError());
    return DLSA_OK;
}

}  // namespace dlsa
