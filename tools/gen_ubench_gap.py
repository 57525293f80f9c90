#!/usr/bin/env python3
"""Generate bench/ubench_gap.hip: what does ONE instruction of each kind cost when it sits between two
v_mfma_f64_16x16x4_f64 of the same wave?  The guide's price list (MI355X_MICROARCH.md, 'one wave per SIMD: single-issue
instructions hidden per MFMA gap') is for bf16 32x32x16 (32 cycles per MFMA); the fp64 kernels of this repo live in
64-cycle gaps and mix fp64 VALU into them (irls_pass.hip), so the prices are measured here.

Every variant is one kernel whose loop body is ONE asm statement: 12 MFMAs on 12 independent accumulators (a[0:95]; a[96:127] serve the small-MFMA / accvgpr fillers), each
followed by the variant's filler instructions.  Fillers write v[64:127] (initialised to 1.0 / finite values).
"""
import sys

NGAP = 12


def mfma(g):
    return f"v_mfma_f64_16x16x4_f64 a[{8 * g}:{8 * g + 7}], %0, %1, a[{8 * g}:{8 * g + 7}]"


def dreg(k):            # k-th fp64 scratch pair
    k %= 24
    return f"v[{64 + 2 * k}:{65 + 2 * k}]"


def sreg(k):
    return f"v{64 + (k % 48)}"


def variants():
    V = []

    def add(name, fn, per_gap, tail=""):
        V.append((name, fn, per_gap, tail))

    add("none", lambda g: [], 0)
    for n in (1, 2, 4, 8):
        add(f"fma64_indep_x{n}", lambda g, n=n: [f"v_fma_f64 {dreg(g * n + j)}, %2, %3, {dreg(g * n + j)}" for j in range(n)], n)
    for n in (1, 2, 4):
        add(f"fma64_chain_x{n}", lambda g, n=n: [f"v_fma_f64 {dreg(0)}, {dreg(0)}, %2, %3" for j in range(n)], n)
    add("mul64_x2", lambda g: [f"v_mul_f64 {dreg(2 * g + j)}, %2, %3" for j in range(2)], 2)
    add("add64_x2", lambda g: [f"v_add_f64 {dreg(2 * g + j)}, %2, %3" for j in range(2)], 2)
    for n in (2, 4, 8):
        add(f"fma32_x{n}", lambda g, n=n: [f"v_fma_f32 {sreg(g * n + j)}, %4, %4, {sreg(g * n + j)}" for j in range(n)], n)
    for n in (2, 4, 8):
        add(f"mov32_x{n}", lambda g, n=n: [f"v_mov_b32 {sreg(g * n + j)}, %4" for j in range(n)], n)
    for n in (2, 4):
        add(f"movdpp_x{n}", lambda g, n=n: [f"v_mov_b32_dpp {sreg(g * n + j)}, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" for j in range(n)], n)
    add("add_u32_x4", lambda g: [f"v_add_u32 {sreg(4 * g + j)}, %4, %4" for j in range(4)], 4)
    add("pkfma32_x2", lambda g: [f"v_pk_fma_f32 {dreg(2 * g + j)}, %2, %3, {dreg(2 * g + j)}" for j in range(2)], 2)
    add("pkfma32_x4", lambda g: [f"v_pk_fma_f32 {dreg(4 * g + j)}, %2, %3, {dreg(4 * g + j)}" for j in range(4)], 4)
    add("rcp64_x1", lambda g: [f"v_rcp_f64 {dreg(g)}, %2"], 1)
    add("ldexp64_x1", lambda g: [f"v_ldexp_f64 {dreg(g)}, %2, 1"], 1)
    add("rndne64_x1", lambda g: [f"v_rndne_f64 {dreg(g)}, %2"], 1)
    add("cmp64_cnd_x1", lambda g: ["v_cmp_gt_f64 vcc, %2, %3", f"v_cndmask_b32 {sreg(2 * g)}, %4, %4, vcc", f"v_cndmask_b32 {sreg(2 * g + 1)}, %4, %4, vcc"], 3)
    add("exp32_x2", lambda g: [f"v_exp_f32 {sreg(2 * g + j)}, %4" for j in range(2)], 2)
    add("cvt_f32_f64_x2", lambda g: [f"v_cvt_f32_f64 {sreg(2 * g + j)}, %2" for j in range(2)], 2)
    add("cvt_f64_f32_x2", lambda g: [f"v_cvt_f64_f32 {dreg(2 * g + j)}, %4" for j in range(2)], 2)
    for n in (1, 2, 4):
        add(f"dsread64_x{n}", lambda g, n=n: [f"ds_read_b64 {dreg((g * n + j) % 8)}, %5 offset:{((g * n + j) % 32) * 512}" for j in range(n)], n,
            "s_waitcnt lgkmcnt(0)")
    for n in (1, 2):
        add(f"dsread128_x{n}", lambda g, n=n: [f"ds_read_b128 v[{112 + 4 * ((g * n + j) % 4)}:{115 + 4 * ((g * n + j) % 4)}], %6 offset:{((g * n + j) % 16) * 1024}" for j in range(n)],
            n, "s_waitcnt lgkmcnt(0)")
    add("dswrite64_x1", lambda g: [f"ds_write_b64 %5, %2 offset:{(g % 32) * 512}"], 1, "s_waitcnt lgkmcnt(0)")
    add("snop_x4", lambda g: ["s_nop 0"] * 4, 4)
    add("salu_x4", lambda g: [f"s_add_u32 s{20 + j}, s{20 + j}, 1" for j in range(4)], 4)
    add("accread_x2", lambda g: [f"v_accvgpr_read_b32 {sreg(2 * g + j)}, a{96 + ((2 * g + j) % 32)}" for j in range(2)], 2)
    add("accwrite_x2", lambda g: [f"v_accvgpr_write_b32 a{96 + ((2 * g + j) % 32)}, %4" for j in range(2)], 2)
    for n in (1, 2):
        add(f"mfma4x4_x{n}", lambda g, n=n: [f"v_mfma_f64_4x4x4_4b_f64 a[{96 + 2 * ((g * n + j) % 16)}:{97 + 2 * ((g * n + j) % 16)}], %0, %1, a[{96 + 2 * ((g * n + j) % 16)}:{97 + 2 * ((g * n + j) % 16)}]"
                                         for j in range(n)], n)
    # VMEM next to the MFMAs: a register load, an LDS-DMA piece (M0 + s_nop + buffer_load ... lds, as hipcc emits it), both L2-hot
    add("gload128_x1", lambda g: [f"global_load_dwordx4 v[{112 + 4 * (g % 4)}:{115 + 4 * (g % 4)}], %7, off"], 1, "s_waitcnt vmcnt(0)")
    add("gload128_x2", lambda g: [f"global_load_dwordx4 v[{112 + 4 * ((2 * g + j) % 4)}:{115 + 4 * ((2 * g + j) % 4)}], %7, off" for j in range(2)], 2, "s_waitcnt vmcnt(0)")
    add("ldsdma_x1", lambda g: [f"s_add_i32 m0, %9, {(g % 8) * 1024}", "s_nop 0", "buffer_load_dwordx4 %6, %8, 0 offen lds"], 1, "s_waitcnt vmcnt(0)")
    add("ldsdma_x2", lambda g: sum([[f"s_add_i32 m0, %9, {((2 * g + j) % 8) * 1024}", "s_nop 0", "buffer_load_dwordx4 %6, %8, 0 offen lds"] for j in range(2)], []), 2,
        "s_waitcnt vmcnt(0)")
    add("ldsdma_nt_x1", lambda g: [f"s_add_i32 m0, %9, {(g % 8) * 1024}", "s_nop 0", "buffer_load_dwordx4 %6, %8, 0 offen nt lds"], 1, "s_waitcnt vmcnt(0)")
    # the same DMA piece in every THIRD gap only (4 per 12 MFMAs: the fused pass's density is 9 per 56)
    add("ldsdma_every3", lambda g: ([f"s_add_i32 m0, %9, {(g % 8) * 1024}", "s_nop 0", "buffer_load_dwordx4 %6, %8, 0 offen lds"] if g % 3 == 0 else []), 1.0 / 3,
        "s_waitcnt vmcnt(0)")
    # the logistic piece as irls_pass.hip spreads it: two dependent fp64 FMAs + a 16-byte LDS read + two moves
    add("mix_2fma64chain_1ds128_2mov", lambda g: [f"v_fma_f64 {dreg(0)}, {dreg(0)}, %2, %3", f"ds_read_b128 v[{112 + 4 * (g % 4)}:{115 + 4 * (g % 4)}], %6 offset:{(g % 16) * 1024}",
                                                 f"v_fma_f64 {dreg(0)}, {dreg(0)}, %2, %3", f"v_mov_b32 {sreg(20 + g)}, %4", f"v_mov_b32 {sreg(21 + g)}, %4"], 5,
        "s_waitcnt lgkmcnt(0)")
    # the same dependent fp64 work in a SECOND, independent chain (two rows' chains interleaved)
    add("fma64_2chains_x4", lambda g: [f"v_fma_f64 {dreg(j & 1)}, {dreg(j & 1)}, %2, %3" for j in range(4)], 4)
    add("fma64_4chains_x4", lambda g: [f"v_fma_f64 {dreg(j & 3)}, {dreg(j & 3)}, %2, %3" for j in range(4)], 4)
    add("fma64_4chains_x8", lambda g: [f"v_fma_f64 {dreg(j & 3)}, {dreg(j & 3)}, %2, %3" for j in range(8)], 8)
    return V


HEADER = r'''// GENERATED by tools/gen_ubench_gap.py -- do not edit.
// Price of one filler instruction between two v_mfma_f64_16x16x4_f64 of the same wave (gfx950).
//   build: hipcc --offload-arch=gfx950 -O3 bench/ubench_gap.hip -o bench/ubench_gap
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

#define CLOB_A "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15","a16","a17","a18","a19","a20","a21","a22","a23","a24","a25","a26","a27","a28","a29","a30","a31","a32","a33","a34","a35","a36","a37","a38","a39","a40","a41","a42","a43","a44","a45","a46","a47","a48","a49","a50","a51","a52","a53","a54","a55","a56","a57","a58","a59","a60","a61","a62","a63","a64","a65","a66","a67","a68","a69","a70","a71","a72","a73","a74","a75","a76","a77","a78","a79","a80","a81","a82","a83","a84","a85","a86","a87","a88","a89","a90","a91","a92","a93","a94","a95","a96","a97","a98","a99","a100","a101","a102","a103","a104","a105","a106","a107","a108","a109","a110","a111","a112","a113","a114","a115","a116","a117","a118","a119","a120","a121","a122","a123","a124","a125","a126","a127"
#define CLOB_V "v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75","v76","v77","v78","v79","v80","v81","v82","v83","v84","v85","v86","v87","v88","v89","v90","v91","v92","v93","v94","v95","v96","v97","v98","v99","v100","v101","v102","v103","v104","v105","v106","v107","v108","v109","v110","v111","v112","v113","v114","v115","v116","v117","v118","v119","v120","v121","v122","v123","v124","v125","v126","v127"
#define CLOB_S "s20","s21","s22","s23","vcc"

'''

KERNEL = r'''
__global__ __launch_bounds__(256, 2) void k_%(name)s(double* out, int iters, long long* cyc, const double* seed) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int tid = threadIdx.x;
    for (int i = tid; i < 4096; i += 256) lds[i] = seed[i & 255];
    __syncthreads();
    const double a = seed[tid & 63], b = seed[64 + (tid & 63)];
    const double x = 0.5, y = 0.5;
    const float f = 0.25f;
    const unsigned ldsaddr = (unsigned)((tid & 63) * 8);
    const double* gptr = seed + 2 * (tid & 63);
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)seed, 0, 2048, 0x00020000);
    const int ldsbase = __builtin_amdgcn_readfirstlane(16384 + (tid >> 6) * 0);
    // scratch pairs = 1.0 (even register 0, odd 0x3ff00000); accumulators 0
    asm volatile(
%(init)s
        ::: CLOB_A, CLOB_V, CLOB_S);
    const long long c0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        asm volatile(
%(body)s
            :: "v"(a), "v"(b), "v"(x), "v"(y), "v"(f), "v"(ldsaddr), "v"(ldsaddr * 2), "v"(gptr), "s"(rsrc), "s"(ldsbase) : CLOB_A, CLOB_V, CLOB_S, "memory");
    }
    const long long c1 = __builtin_readcyclecounter();
    double s;
    asm volatile("s_nop 15\n\ts_nop 15\n\tv_accvgpr_read_b32 %%0, a0\n\tv_accvgpr_read_b32 %%1, a1" : "=v"(((int*)&s)[0]), "=v"(((int*)&s)[1]) :: CLOB_A);
    if (s == 12345.678) out[0] = s;
    if (blockIdx.x == 0 && tid == 0) cyc[0] = c1 - c0;
}
'''


def main(path):
    out = [HEADER]
    V = variants()
    init = []
    for r in range(64, 128, 2):
        init.append(f'        "v_mov_b32 v{r}, 0\\n\\tv_mov_b32 v{r + 1}, 0x3ff00000\\n\\t"')
    for r in range(0, 128, 1):
        init.append(f'        "v_accvgpr_write_b32 a{r}, 0\\n\\t"')
    init.append('        "s_mov_b32 s20, 0\\n\\ts_mov_b32 s21, 0\\n\\ts_mov_b32 s22, 0\\n\\ts_mov_b32 s23, 0\\n\\t"')
    init_s = "\n".join(init)
    for name, fn, per_gap, tail in V:
        lines = []
        for g in range(NGAP):
            lines.append(mfma(g))
            lines.extend(fn(g))
        if tail:
            lines.append(tail)
        body = "\n".join(f'            "{l}\\n\\t"' for l in lines)
        out.append(KERNEL % dict(name=name, init=init_s, body=body))
    out.append("struct Var { const char* name; void (*fn)(double*, int, long long*, const double*); double per_gap; };\n")
    out.append("static const Var VARS[] = {\n")
    for name, fn, per_gap, tail in V:
        out.append(f'    {{"{name}", k_{name}, {float(per_gap)}}},\n')
    out.append("};\n")
    out.append(r'''
int main(int argc, char** argv) {
    const char* only = argc > 1 ? argv[1] : nullptr;
    hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
    printf("device: %s  CUs=%d\n", pr.name, pr.multiProcessorCount);
    double* d; CK(hipMalloc(&d, 64));
    long long* dc; CK(hipMalloc(&dc, 64));
    double hs[256];
    for (int i = 0; i < 256; ++i) hs[i] = 0.001 * (i % 97) - 0.04;
    double* seed; CK(hipMalloc(&seed, sizeof hs)); CK(hipMemcpy(seed, hs, sizeof hs, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 4000;
    printf("%-32s %4s %9s %9s %9s %9s\n", "variant", "w/S", "ms", "TF(mfma)", "cyc/mfma", "cyc/filler");
    for (int wps = 1; wps <= 2; ++wps) {
        double base = 0;
        for (const Var& v : VARS) {
            if (only && strcmp(only, v.name) && strcmp(v.name, "none")) continue;
            const int blocks = 256 * wps;
            hipLaunchKernelGGL(v.fn, dim3(blocks), dim3(256), 32768, 0, d, 50, dc, seed);
            CK(hipDeviceSynchronize());
            float best = 1e9f; long long hc = 0;
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipEventRecord(e0));
                hipLaunchKernelGGL(v.fn, dim3(blocks), dim3(256), 32768, 0, d, iters, dc, seed);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) { best = ms; CK(hipMemcpy(&hc, dc, 8, hipMemcpyDeviceToHost)); }
            }
            const double tf = (double)blocks * 4 * iters * 12 * 2048.0 / best * 1e-9;
            // per SIMD: wps waves share the pipe, so the SIMD's cycles per MFMA = wave cycles / (16 * wps)
            const double cpm = (double)hc / iters / 12.0 / wps;
            if (!strcmp(v.name, "none")) base = cpm;
            printf("%-32s %4d %9.3f %9.2f %9.2f %9.2f\n", v.name, wps, best, tf, cpm, v.per_gap > 0 ? (cpm - base) / v.per_gap : 0.0);
        }
    }
    return 0;
}
''')
    with open(path, "w") as f:
        f.write("".join(out))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "bench/ubench_gap.hip")
