// Probe the operand/result lane layout of v_mfma_f64_4x4x4_4b_f64 on gfx950, with and without
// the CBSZ/ABID A-block broadcast.  Prints, for hypotheses H, whether D matches.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int CBSZ, int ABID>
__global__ void k(const double* a, const double* b, double* d) {
    const int l = threadIdx.x;
    d[l] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[l], b[l], 0.0, CBSZ, ABID, 0);
}

// indicator probe: for every (la, lb) which output lanes get a*b ?
__global__ void kind(int* out) {
    const int l = threadIdx.x;
    for (int la = 0; la < 64; ++la)
        for (int lb = 0; lb < 64; ++lb) {
            double d = __builtin_amdgcn_mfma_f64_4x4x4f64(l == la ? 1.0 : 0.0, l == lb ? 1.0 : 0.0, 0.0, 0, 0, 0);
            if (d != 0.0) out[la * 64 + lb] = l + 1;   // at most one lane per pair if a bijection
        }
}

int main() {
    double ha[64], hb[64], hd[64];
    for (int l = 0; l < 64; ++l) { ha[l] = 1.0 + 0.37 * l + 0.011 * l * l; hb[l] = 2.0 - 0.21 * l + 0.007 * l * l; }
    double *a, *b, *d; CK(hipMalloc(&a, 512)); CK(hipMalloc(&b, 512)); CK(hipMalloc(&d, 512));
    CK(hipMemcpy(a, ha, 512, hipMemcpyHostToDevice)); CK(hipMemcpy(b, hb, 512, hipMemcpyHostToDevice));
    int* dout; CK(hipMalloc(&dout, 64 * 64 * 4)); CK(hipMemset(dout, 0, 64 * 64 * 4));
    hipLaunchKernelGGL(kind, dim3(1), dim3(64), 0, 0, dout);
    static int hout[64 * 64];
    CK(hipMemcpy(hout, dout, sizeof(hout), hipMemcpyDeviceToHost));
    printf("indicator map (la, lb) -> output lane, cbsz=0:\n");
    int cnt = 0;
    for (int la = 0; la < 64; ++la) for (int lb = 0; lb < 64; ++lb) if (hout[la * 64 + lb]) {
        if (cnt < 40) printf("  a-lane %2d  b-lane %2d -> d-lane %2d\n", la, lb, hout[la * 64 + lb] - 1);
        ++cnt;
    }
    printf("pairs contributing: %d (expect 4 blocks*4*4*4 = 256)\n", cnt);
    // hypothesis: A[i_glob = l&15][k = l>>4], B[k = l>>4][j_glob = l&15], D lane l: row (l>>4) in block (l&15)>>2, col l&15
    auto check = [&](int cbsz, int abid, const char* name) {
        CK(hipMemcpy(hd, d, 512, hipMemcpyDeviceToHost));
        double maxerr = 0;
        for (int l = 0; l < 64; ++l) {
            const int col = l & 15, i = l >> 4, blk = col >> 2;
            const int ablk = cbsz ? abid : blk;
            double s = 0;
            for (int kk = 0; kk < 4; ++kk) s += ha[(kk << 4) | (ablk * 4 + i)] * hb[(kk << 4) | col];
            maxerr = fmax(maxerr, fabs(s - hd[l]));
        }
        double e_ign = 0, e_bb = 0;
        for (int l = 0; l < 64; ++l) {
            const int col = l & 15, i = l >> 4, blk = col >> 2;
            double s0 = 0, s1 = 0;
            for (int kk = 0; kk < 4; ++kk) {
                s0 += ha[(kk << 4) | (blk * 4 + i)] * hb[(kk << 4) | col];                       // modifiers ignored
                s1 += ha[(kk << 4) | (blk * 4 + i)] * hb[(kk << 4) | (abid * 4 + (col & 3))];    // B block broadcast
            }
            e_ign = fmax(e_ign, fabs(s0 - hd[l])); e_bb = fmax(e_bb, fabs(s1 - hd[l]));
        }
        printf("%s: |D - A-bcast| = %.3e   |D - ignored| = %.3e   |D - B-bcast| = %.3e\n", name, maxerr, e_ign, e_bb);
    };
    hipLaunchKernelGGL((k<0, 0>), dim3(1), dim3(64), 0, 0, a, b, d); check(0, 0, "cbsz=0 abid=0 (4 independent blocks)");
    hipLaunchKernelGGL((k<2, 0>), dim3(1), dim3(64), 0, 0, a, b, d); check(2, 0, "cbsz=2 abid=0 (A block 0 -> all blocks)");
    hipLaunchKernelGGL((k<2, 1>), dim3(1), dim3(64), 0, 0, a, b, d); check(2, 1, "cbsz=2 abid=1");
    hipLaunchKernelGGL((k<2, 2>), dim3(1), dim3(64), 0, 0, a, b, d); check(2, 2, "cbsz=2 abid=2");
    hipLaunchKernelGGL((k<2, 3>), dim3(1), dim3(64), 0, 0, a, b, d); check(2, 3, "cbsz=2 abid=3");
    CK(hipMemcpy(hd, d, 512, hipMemcpyDeviceToHost));
    return 0;
}
