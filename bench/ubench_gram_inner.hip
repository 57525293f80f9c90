// Inner-loop ablation for the fp64 Gram kernel on gfx950: which ingredient of the
// "4x4x4 MFMA tile-step" limits throughput?  8 waves per workgroup (2 per SIMD), 11 tiles/wave.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef double d4 __attribute__((ext_vector_type(4)));
constexpr int NT = 11;
constexpr int LDP = 144;

template <int N>
__device__ __forceinline__ double row_ror_f64(double v) {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)u, 0x120 + N, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(u >> 32), 0x120 + N, 0xF, 0xF, false);
    return __builtin_bit_cast(double, ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
}

// MODE 0: 4x4x4, operands in fixed registers     MODE 1: + per-tile A/B from register arrays
// MODE 2: MODE 1 + v_mul + 3 DPP rotations        MODE 3: LDS reads (A,B) + v_mul, no DPP (b reused unrotated)
// MODE 4: LDS reads + v_mul + DPP (the kernel)    MODE 5: 16x16x4 with LDS reads + v_mul (v1 kernel)
// MODE 6: LDS reads of 4 rotated B + v_mul (no DPP)
template <int MODE>
__global__ __launch_bounds__(512, 2) void inner(double* out, int iters, const double* seed) {
    __shared__ __attribute__((aligned(16))) double lds[2 * 32 * LDP + 64];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 2 * 32 * LDP + 64; i += 512) lds[i] = seed[i & 1023];
    __syncthreads();
    d4 acc[NT];
    for (int t = 0; t < NT; ++t) acc[t] = d4{0, 0, 0, 0};
    int offA[NT], offB[NT];
    for (int t = 0; t < NT; ++t) {
        const int s = __builtin_amdgcn_readfirstlane((int)seed[t + 5] & 7);
        offA[t] = ((t * 3 + s) & 7) * 16;
        offB[t] = 32 * LDP + ((t * 5 + s) & 7) * 16;
    }
    const int lane_off = (lane >> 4) * LDP + (lane & 15);
    double ra[NT], rb[NT];
    for (int t = 0; t < NT; ++t) { ra[t] = seed[t] + lane; rb[t] = seed[t + 20] - lane; }
    const double wv = seed[3];
    for (int it = 0; it < iters; ++it) {
#pragma unroll 1
        for (int ks = 0; ks < 8; ++ks) {
            const double* kb = lds + ks * 4 * LDP + lane_off;
            const double* kr = lds + ks * 4 * LDP + (lane >> 4) * LDP;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                if (MODE == 0) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc[t][q] = __builtin_amdgcn_mfma_f64_4x4x4f64(ra[0], rb[0], acc[t][q], 0, 0, 0);
                } else if (MODE == 1) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc[t][q] = __builtin_amdgcn_mfma_f64_4x4x4f64(ra[t], rb[(t + q) % NT], acc[t][q], 0, 0, 0);
                } else if (MODE == 2) {
                    const double b0 = rb[t] * wv;
                    acc[t][0] = __builtin_amdgcn_mfma_f64_4x4x4f64(ra[t], b0, acc[t][0], 0, 0, 0);
                    acc[t][1] = __builtin_amdgcn_mfma_f64_4x4x4f64(ra[t], row_ror_f64<4>(b0), acc[t][1], 0, 0, 0);
                    acc[t][2] = __builtin_amdgcn_mfma_f64_4x4x4f64(ra[t], row_ror_f64<8>(b0), acc[t][2], 0, 0, 0);
                    acc[t][3] = __builtin_amdgcn_mfma_f64_4x4x4f64(ra[t], row_ror_f64<12>(b0), acc[t][3], 0, 0, 0);
                } else if (MODE == 3) {
                    const double av = kb[offA[t]];
                    const double b0 = kb[offB[t]] * wv;
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc[t][q] = __builtin_amdgcn_mfma_f64_4x4x4f64(av, b0, acc[t][q], 0, 0, 0);
                } else if (MODE == 4) {
                    const double av = kb[offA[t]];
                    const double b0 = kb[offB[t]] * wv;
                    acc[t][0] = __builtin_amdgcn_mfma_f64_4x4x4f64(av, b0, acc[t][0], 0, 0, 0);
                    acc[t][1] = __builtin_amdgcn_mfma_f64_4x4x4f64(av, row_ror_f64<4>(b0), acc[t][1], 0, 0, 0);
                    acc[t][2] = __builtin_amdgcn_mfma_f64_4x4x4f64(av, row_ror_f64<8>(b0), acc[t][2], 0, 0, 0);
                    acc[t][3] = __builtin_amdgcn_mfma_f64_4x4x4f64(av, row_ror_f64<12>(b0), acc[t][3], 0, 0, 0);
                } else if (MODE == 5) {
                    const double av = kb[offA[t]];
                    const double b0 = kb[offB[t]] * wv;
                    acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, b0, acc[t], 0, 0, 0);
                } else {
                    const double av = kr[offA[t] + (lane & 15)] * wv;
                    acc[t][0] = __builtin_amdgcn_mfma_f64_4x4x4f64(av, kr[offB[t] + (lane & 15)], acc[t][0], 0, 0, 0);
                    acc[t][1] = __builtin_amdgcn_mfma_f64_4x4x4f64(av, kr[offB[t] + ((lane - 4) & 15)], acc[t][1], 0, 0, 0);
                    acc[t][2] = __builtin_amdgcn_mfma_f64_4x4x4f64(av, kr[offB[t] + ((lane - 8) & 15)], acc[t][2], 0, 0, 0);
                    acc[t][3] = __builtin_amdgcn_mfma_f64_4x4x4f64(av, kr[offB[t] + ((lane - 12) & 15)], acc[t][3], 0, 0, 0);
                }
            }
        }
    }
    double s = 0;
    for (int t = 0; t < NT; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
    if (s == 12345.678) out[0] = s;
}

// Register-blocked variants: a wave owns an MR x NR block of tiles (static structure).
// BMODE 0: 16x16x4, A/B frags from LDS (MR + NR reads, NR muls)   BMODE 1: 4x4x4 with 4 rotated B reads from LDS
// BMODE 2: 4x4x4 with DPP-rotated B                              BMODE 3: 16x16x4 operands in registers only
template <int BMODE, int MR, int NR>
__global__ __launch_bounds__(512, 2) void blocked(double* out, int iters, const double* seed) {
    __shared__ __attribute__((aligned(16))) double lds[2 * 32 * LDP + 64];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 2 * 32 * LDP + 64; i += 512) lds[i] = seed[i & 1023];
    __syncthreads();
    d4 acc[MR][NR];
    for (int i = 0; i < MR; ++i) for (int j = 0; j < NR; ++j) acc[i][j] = d4{0, 0, 0, 0};
    int offA[MR], offB[NR];
    for (int i = 0; i < MR; ++i) offA[i] = ((i * 3 + __builtin_amdgcn_readfirstlane((int)seed[i + 5] & 7)) & 7) * 16;
    for (int j = 0; j < NR; ++j) offB[j] = 32 * LDP + ((j * 5 + __builtin_amdgcn_readfirstlane((int)seed[j + 9] & 7)) & 7) * 16;
    const int lane_off = (lane >> 4) * LDP + (lane & 15);
    const double wv = seed[3];
    double ra[MR], rb[NR];
    for (int i = 0; i < MR; ++i) ra[i] = seed[i] + lane;
    for (int j = 0; j < NR; ++j) rb[j] = seed[j + 20] - lane;
    for (int it = 0; it < iters; ++it) {
#pragma unroll 1
        for (int ks = 0; ks < 8; ++ks) {
            const double* kb = lds + ks * 4 * LDP + lane_off;
            const double* kr = lds + ks * 4 * LDP + (lane >> 4) * LDP;
            if (BMODE == 3) {
#pragma unroll
                for (int i = 0; i < MR; ++i)
#pragma unroll
                    for (int j = 0; j < NR; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(ra[i], rb[j], acc[i][j], 0, 0, 0);
            } else if (BMODE == 0) {
                double av[MR], bv[NR];
#pragma unroll
                for (int i = 0; i < MR; ++i) av[i] = kb[offA[i]];
#pragma unroll
                for (int j = 0; j < NR; ++j) bv[j] = kb[offB[j]] * wv;
#pragma unroll
                for (int i = 0; i < MR; ++i)
#pragma unroll
                    for (int j = 0; j < NR; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[i], bv[j], acc[i][j], 0, 0, 0);
            } else if (BMODE == 1) {
                double av[MR];
#pragma unroll
                for (int i = 0; i < MR; ++i) av[i] = kb[offA[i]] * wv;
#pragma unroll
                for (int j = 0; j < NR; ++j) {
                    const double b0 = kr[offB[j] + (lane & 15)], b1 = kr[offB[j] + ((lane - 4) & 15)];
                    const double b2 = kr[offB[j] + ((lane - 8) & 15)], b3 = kr[offB[j] + ((lane - 12) & 15)];
#pragma unroll
                    for (int i = 0; i < MR; ++i) {
                        acc[i][j][0] = __builtin_amdgcn_mfma_f64_4x4x4f64(av[i], b0, acc[i][j][0], 0, 0, 0);
                        acc[i][j][1] = __builtin_amdgcn_mfma_f64_4x4x4f64(av[i], b1, acc[i][j][1], 0, 0, 0);
                        acc[i][j][2] = __builtin_amdgcn_mfma_f64_4x4x4f64(av[i], b2, acc[i][j][2], 0, 0, 0);
                        acc[i][j][3] = __builtin_amdgcn_mfma_f64_4x4x4f64(av[i], b3, acc[i][j][3], 0, 0, 0);
                    }
                }
            } else {
                double av[MR];
#pragma unroll
                for (int i = 0; i < MR; ++i) av[i] = kb[offA[i]] * wv;
#pragma unroll
                for (int j = 0; j < NR; ++j) {
                    const double b0 = kb[offB[j]];
                    const double b1 = row_ror_f64<4>(b0), b2 = row_ror_f64<8>(b0), b3 = row_ror_f64<12>(b0);
#pragma unroll
                    for (int i = 0; i < MR; ++i) {
                        acc[i][j][0] = __builtin_amdgcn_mfma_f64_4x4x4f64(av[i], b0, acc[i][j][0], 0, 0, 0);
                        acc[i][j][1] = __builtin_amdgcn_mfma_f64_4x4x4f64(av[i], b1, acc[i][j][1], 0, 0, 0);
                        acc[i][j][2] = __builtin_amdgcn_mfma_f64_4x4x4f64(av[i], b2, acc[i][j][2], 0, 0, 0);
                        acc[i][j][3] = __builtin_amdgcn_mfma_f64_4x4x4f64(av[i], b3, acc[i][j][3], 0, 0, 0);
                    }
                }
            }
        }
    }
    double s = 0;
    for (int i = 0; i < MR; ++i) for (int j = 0; j < NR; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (s == 12345.678) out[0] = s;
}

template <int BMODE, int MR, int NR>
static void runb(const char* name, const double* dseed, double* dout) {
    const int iters = 300;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((blocked<BMODE, MR, NR>), dim3(256), dim3(512), 0, 0, dout, 3, dseed);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((blocked<BMODE, MR, NR>), dim3(256), dim3(512), 0, 0, dout, iters, dseed);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double tilesteps = 256.0 * 8 * iters * 8 * MR * NR;
    printf("%-52s %dx%d %8.3f ms  %6.2f TF executed   %.1f cycles/tile-step/SIMD @2.3GHz\n", name, MR, NR, ms, tilesteps * 2048 / ms * 1e-9,
           ms * 1e-3 * 2.3e9 / (tilesteps / 1024));
}

template <int MODE>
static void run(const char* name, const double* dseed, double* dout) {
    const int iters = 300;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((inner<MODE>), dim3(256), dim3(512), 0, 0, dout, 3, dseed);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((inner<MODE>), dim3(256), dim3(512), 0, 0, dout, iters, dseed);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double tilesteps = 256.0 * 8 * iters * 8 * NT;
    printf("%-52s %8.3f ms  %6.2f TF executed   %.1f cycles/tile-step/SIMD @2.3GHz\n", name, ms, tilesteps * 2048 / ms * 1e-9,
           ms * 1e-3 * 2.3e9 / (tilesteps / 1024));
}

int main() {
    double h[1024];
    for (int i = 0; i < 1024; ++i) h[i] = 0.3 + 0.001 * ((i * 7919) % 1000);
    double *dseed, *dout; CK(hipMalloc(&dseed, sizeof(h))); CK(hipMalloc(&dout, 64));
    CK(hipMemcpy(dseed, h, sizeof(h), hipMemcpyHostToDevice));
    run<0>("0: 4x4x4, fixed operands", dseed, dout);
    run<1>("1: 4x4x4, per-tile A/B registers", dseed, dout);
    run<2>("2: 4x4x4, registers + v_mul + 3 DPP rotations", dseed, dout);
    run<3>("3: 4x4x4, LDS A/B reads + v_mul (no rotation)", dseed, dout);
    run<4>("4: 4x4x4, LDS A/B reads + v_mul + DPP (kernel)", dseed, dout);
    run<6>("6: 4x4x4, LDS A + 4 rotated B reads + v_mul", dseed, dout);
    run<5>("5: 16x16x4, LDS A/B reads + v_mul (v1 kernel)", dseed, dout);
    runb<3, 3, 4>("B3: 16x16x4 registers only", dseed, dout);
    runb<0, 1, 11>("B0: 16x16x4 blocked, LDS frags", dseed, dout);
    runb<0, 2, 4>("B0: 16x16x4 blocked, LDS frags", dseed, dout);
    runb<0, 2, 6>("B0: 16x16x4 blocked, LDS frags", dseed, dout);
    runb<0, 3, 4>("B0: 16x16x4 blocked, LDS frags", dseed, dout);
    runb<0, 4, 4>("B0: 16x16x4 blocked, LDS frags", dseed, dout);
    runb<1, 2, 4>("B1: 4x4x4 blocked, 4 rotated B reads from LDS", dseed, dout);
    runb<1, 3, 4>("B1: 4x4x4 blocked, 4 rotated B reads from LDS", dseed, dout);
    runb<1, 4, 3>("B1: 4x4x4 blocked, 4 rotated B reads from LDS", dseed, dout);
    runb<1, 4, 4>("B1: 4x4x4 blocked, 4 rotated B reads from LDS", dseed, dout);
    runb<2, 3, 4>("B2: 4x4x4 blocked, DPP-rotated B", dseed, dout);
    runb<2, 4, 3>("B2: 4x4x4 blocked, DPP-rotated B", dseed, dout);
    runb<2, 4, 4>("B2: 4x4x4 blocked, DPP-rotated B", dseed, dout);
    return 0;
}
