// Cost of a grid-wide barrier (monotonic counter, agent-scope release/acquire) on gfx950, with the workgroups
// spread over all XCDs or confined to one (workgroups are dealt round-robin to the XCDs: blockIdx % 8).
// Each round also passes a value through global memory to the next workgroup and checks it.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ void grid_barrier(unsigned* cnt, unsigned nwg, unsigned& phase) {
    __syncthreads();
    if (threadIdx.x == 0) {
        ++phase;
        const unsigned target = phase * nwg;
        __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(cnt, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
}

__global__ __launch_bounds__(1024) void bar_kernel(unsigned* cnt, double* box, int rounds, int stride, int nwg, int* bad) {
    if (blockIdx.x % stride != 0) return;
    const int me = blockIdx.x / stride;
    unsigned phase = 0;
    int errors = 0;
    for (int r = 0; r < rounds; ++r) {
        if (threadIdx.x < 64) box[me * 64 + threadIdx.x] = r * 1000.0 + me;
        grid_barrier(cnt, nwg, phase);
        const int nb = (me + 1) % nwg;
        if (threadIdx.x < 64) {
            const double v = box[nb * 64 + threadIdx.x];      // plain cached load: the acquire must have invalidated stale lines
            if (v != r * 1000.0 + nb) ++errors;
        }
        grid_barrier(cnt, nwg, phase);
    }
    if (errors) atomicAdd(bad, errors);
}

int main() {
    unsigned* cnt; double* box; int* bad;
    (void)hipMalloc(&cnt, 4); (void)hipMalloc(&box, 256 * 64 * 8); (void)hipMalloc(&bad, 4);
    const int rounds = 2000;
    for (int stride : {1, 8}) {
        for (int nwg : {2, 8, 16, 32, 64, 128, 256}) {
            if (nwg * stride > 256) continue;
            float best = 1e9f; int hbad = 0;
            for (int rep = 0; rep < 3; ++rep) {
                (void)hipMemset(cnt, 0, 4); (void)hipMemset(bad, 0, 4);
                hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
                (void)hipEventRecord(e0);
                hipLaunchKernelGGL(bar_kernel, dim3(nwg * stride), dim3(1024), 0, 0, cnt, box, rounds, stride, nwg, bad);
                (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
                float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
                (void)hipMemcpy(&hbad, bad, 4, hipMemcpyDeviceToHost);
            }
            printf("stride %d nwg %3d: %.2f us per barrier (errors %d)\n", stride, nwg, best * 1e3 / (2 * rounds), hbad);
        }
    }
    return 0;
}
