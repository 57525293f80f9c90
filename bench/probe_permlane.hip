// lane-mapping probe of the gfx950 cross-row primitives used for wave reductions:
// v_permlane16_swap / v_permlane32_swap and the DPP controls row_ror:8, row_half_mirror, quad_perm.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(int* out) {
    const int l = threadIdx.x;
    auto s16 = __builtin_amdgcn_permlane16_swap(l, l, false, false);
    auto s32 = __builtin_amdgcn_permlane32_swap(l, l, false, false);
    out[0 * 64 + l] = s16[0]; out[1 * 64 + l] = s16[1];
    out[2 * 64 + l] = s32[0]; out[3 * 64 + l] = s32[1];
    out[4 * 64 + l] = __builtin_amdgcn_update_dpp(l, l, 0x128, 0xF, 0xF, false);   // row_ror:8
    int t = __builtin_amdgcn_update_dpp(l, l, 0x141, 0xF, 0xF, false);               // row_half_mirror
    out[5 * 64 + l] = __builtin_amdgcn_update_dpp(t, t, 0x1B, 0xF, 0xF, false);      // then quad_perm [3,2,1,0]
    out[6 * 64 + l] = __builtin_amdgcn_update_dpp(l, l, 0x4E, 0xF, 0xF, false);     // quad_perm [2,3,0,1]
    out[7 * 64 + l] = __builtin_amdgcn_update_dpp(l, l, 0xB1, 0xF, 0xF, false);     // quad_perm [1,0,3,2]
}
int main() {
    int* d; hipMalloc(&d, 8 * 64 * sizeof(int));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    int h[8 * 64]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[8] = {"permlane16_swap[0]", "permlane16_swap[1]", "permlane32_swap[0]", "permlane32_swap[1]", "row_ror:8", "half_mirror+qp3210", "quad_perm 2301", "quad_perm 1032"};
    for (int r = 0; r < 8; ++r) { printf("%-20s", names[r]); for (int l = 0; l < 64; ++l) printf(" %d", h[r * 64 + l]); printf("\n"); }
    return 0;
}
