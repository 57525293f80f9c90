// Micro-benchmark: fp64 matrix (v_mfma_f64_16x16x4_f64) and vector (v_fma_f64) peaks on gfx950,
// alone and co-issued, plus a streaming-read bandwidth probe.  The in-container guide has no
// fp64 MFMA row; this measures the ceiling the Gram kernel's roofline is priced against.
//   build: hipcc --offload-arch=gfx950 -O3 bench/ubench_f64.hip -o bench/ubench_f64
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef double d4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int NACC, bool MFMA, bool VALU>
__global__ __launch_bounds__(256) void peak_kernel(double* out, int iters, double seed, long long* cyc) {
    const long long c0 = clock64();
    d4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = d4{seed, seed, seed, seed};
    double a = seed + threadIdx.x * 1e-9, b = seed * 0.5;
    double v[8];
    for (int i = 0; i < 8; ++i) v[i] = seed + i;
    for (int it = 0; it < iters; ++it) {
        if (MFMA) {
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
        }
        if (VALU) {
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = fma(v[i], a, b);
        }
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 8; ++i) s += v[i];
    if (s == 12345.678) out[0] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = clock64() - c0;
}

__global__ void read_kernel(const double2* __restrict__ in, size_t n, double* out) {
    double s = 0;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        double2 v = in[i];
        s += v.x + v.y;
    }
    if (s == 12345.678) out[0] = s;
}

template <int NACC, bool MFMA, bool VALU>
static void run(const char* name, int waves_per_simd) {
    double* d; CK(hipMalloc(&d, 64));
    long long* dc; CK(hipMalloc(&dc, 64));
    const int iters = 20000;
    const int blocks = 256 * waves_per_simd;   // 256 threads = 4 waves = 1 per SIMD per block
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((peak_kernel<NACC, MFMA, VALU>), dim3(blocks), dim3(256), 0, 0, d, 100, 1.0, dc);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((peak_kernel<NACC, MFMA, VALU>), dim3(blocks), dim3(256), 0, 0, d, iters, 1.0, dc);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double waves = (double)blocks * 4;
    const double mf = MFMA ? waves * iters * NACC * 2048.0 : 0.0;
    const double vf = VALU ? waves * iters * 16 * 128.0 : 0.0;
    long long hc = 0; CK(hipMemcpy(&hc, dc, 8, hipMemcpyDeviceToHost));
    printf("%-30s w/SIMD=%d %8.3f ms  mfma %6.2f TF  valu %6.2f TF  total %6.2f TF  s_memtime %.0f MHz  cyc/iter %.1f\n", name,
           waves_per_simd, ms, mf / ms * 1e-9, vf / ms * 1e-9, (mf + vf) / ms * 1e-9, hc / (ms * 1e3), (double)hc / iters);
    CK(hipFree(d));
}

int main() {
    hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
    printf("device: %s  CUs=%d  clock=%d kHz\n", pr.name, pr.multiProcessorCount, pr.clockRate);
    run<1, true, false>("mfma_f64_16x16x4 x1 acc", 1);
    run<2, true, false>("mfma_f64_16x16x4 x2 acc", 1);
    run<4, true, false>("mfma_f64_16x16x4 x4 acc", 1);
    run<8, true, false>("mfma_f64_16x16x4 x8 acc", 1);
    run<8, true, false>("mfma_f64_16x16x4 x8 acc", 2);
    run<4, true, false>("mfma_f64_16x16x4 x4 acc", 2);
    run<4, true, false>("mfma_f64_16x16x4 x4 acc", 4);
    run<4, true, false>("mfma_f64_16x16x4 x4 acc", 8);
    run<2, true, false>("mfma_f64_16x16x4 x2 acc", 8);
    run<1, false, true>("v_fma_f64 only", 1);
    run<1, false, true>("v_fma_f64 only", 2);
    run<1, false, true>("v_fma_f64 only", 4);
    run<8, true, true>("mfma x8 + 16 v_fma per iter", 1);
    run<8, true, true>("mfma x8 + 16 v_fma per iter", 2);
    run<4, true, true>("mfma x4 + 16 v_fma per iter", 2);
    // streaming read
    size_t bytes = (size_t)8 << 30;
    double2* buf; CK(hipMalloc(&buf, bytes)); CK(hipMemset(buf, 0, bytes));
    double* d; CK(hipMalloc(&d, 64));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(read_kernel, dim3(256 * 16), dim3(256), 0, 0, buf, bytes / 16, d);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("stream read 8 GiB: %.3f ms  %.2f TB/s\n", ms, bytes / ms * 1e-9);
    }
    return 0;
}
