// Timing probe for ONE width of the plan-driven Gram kernel, built with experiment knobs (bench/probe/build_probe.sh).
// Not part of the product: links gram_plan_unit.hip compiled for a single NT with -DPLAN_PROBE_* knobs.
#include "../../dlsa_amd/csrc/gram_plan.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#ifndef PROBE_NT
#error PROBE_NT
#endif
#define CAT2(a, b) a##b
#define CAT(a, b) CAT2(a, b)
namespace dlsa { int CAT(gram_plan_launch_, PROBE_NT)(const PlanArgs& a, int nt, int g, int nslab, hipStream_t stream); }

__global__ void fill(double* x, size_t n, unsigned seed) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u + seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        x[i] = (double)(int)(h & 0xffff) * (1.0 / 65536.0) - 0.5 + 1e-9 * (double)(h >> 16);
    }
}

int main(int argc, char** argv) {
    const long rows = argc > 1 ? atol(argv[1]) : 4000000;
    const int p = argc > 2 ? atoi(argv[2]) : 500;
    const int reps = argc > 3 ? atoi(argv[3]) : 5;
    int nt = p / 16, g = (p - 16 * nt + 3) / 4;
    if (g == 4) { ++nt; g = 0; }
    const int C = dlsa::plan_group(nt), nslab = 256 / C, PP = ((p + 15) / 16 * 16 + 63) / 64 * 64;
    double *X, *w, *part; int* prog;
    hipMalloc(&X, (size_t)rows * p * 8); hipMalloc(&w, rows * 8); hipMalloc(&part, (size_t)nslab * PP * PP * 8); hipMalloc(&prog, nslab * 16);
    fill<<<1024, 256>>>(X, (size_t)rows * p, 1u); fill<<<256, 256>>>(w, rows, 7u);
    dlsa::PlanArgs a;
    a.X = X; a.w = w; a.w_step = 1; a.partial = part; a.progress = prog; a.ldx = p; a.n = rows; a.p = p; a.PP = PP;
    a.rows_per_slab = ((rows + nslab - 1) / nslab + 15) / 16 * 16;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<float> ts;
    for (int r = 0; r < reps + 1; ++r) {
        hipMemsetAsync(prog, 0, nslab * 16, 0);
        hipEventRecord(e0, 0);
        int rc = dlsa::CAT(gram_plan_launch_, PROBE_NT)(a, nt, g, nslab, 0);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        if (rc || hipGetLastError() != hipSuccess) { printf("launch failed %d\n", rc); return 1; }
        float ms; hipEventElapsedTime(&ms, e0, e1); if (r) ts.push_back(ms);
    }
    std::sort(ts.begin(), ts.end());
    const double ms = ts[ts.size() / 2];
    printf("PROBE %s rows=%ld p=%d: median %.3f ms min %.3f  %.2f TF(alg)\n", PROBE_TAG, rows, p, ms, ts[0], rows * ((double)p * (p + 1) + p) / ms * 1e-9);
    return 0;
}
