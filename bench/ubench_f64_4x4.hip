// Micro-benchmark: v_mfma_f64_4x4x4_4b_f64 issue rate on gfx950 (is it faster per flop than 16x16x4?)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int NACC>
__global__ __launch_bounds__(256) void k4(double* out, int iters, double seed, long long* cyc) {
    const long long c0 = clock64();
    double acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = seed;
    double a = seed + threadIdx.x * 1e-9, b = seed * 0.5;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i];
    if (s == 12345.678) out[0] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = clock64() - c0;
}

template <int NACC>
static void run(int wps) {
    double* d; CK(hipMalloc(&d, 64));
    long long* dc; CK(hipMalloc(&dc, 64));
    const int iters = 50000;
    const int blocks = 256 * wps;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k4<NACC>), dim3(blocks), dim3(256), 0, 0, d, 100, 1.0, dc);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k4<NACC>), dim3(blocks), dim3(256), 0, 0, d, iters, 1.0, dc);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    long long hc = 0; CK(hipMemcpy(&hc, dc, 8, hipMemcpyDeviceToHost));
    const double fl = (double)blocks * 4 * iters * NACC * 512.0;
    printf("mfma_f64_4x4x4_4b x%-2d acc  w/SIMD=%d %8.3f ms  %6.2f TF  s_memtime %.0f MHz  cyc per mfma (block0) %.1f\n",
           NACC, wps, ms, fl / ms * 1e-9, hc / (ms * 1e3), (double)hc / iters / NACC / wps);
}

int main() {
    run<1>(1); run<4>(1); run<8>(1); run<16>(1); run<16>(2); run<8>(4); run<8>(8);
    return 0;
}
