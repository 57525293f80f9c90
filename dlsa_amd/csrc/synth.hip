// Counter-based synthetic rows (replaces simulate_logistic, dlsa/models.py:6-40, whose
// per-row Python loop is O(n^2) and unseeded).  Row i is a pure function of (seed, i):
// Philox-4x32-10 with counter (i_lo, i_hi, pair, stream) and key (seed, 0).  The integer
// pipeline and the uint32 -> double conversion are bit-identical to oracle/dlsa_oracle.py, so
// the uniform variant gives the CPU oracle and every GPU shard exactly the same rows.
#include "common.h"
#include <algorithm>

namespace dlsa {

struct u4 { unsigned x, y, z, w; };

__device__ __forceinline__ u4 philox4x32_10(u4 c, unsigned k0, unsigned k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = 0xD2511F53ull * c.x;
        const unsigned long long p1 = 0xCD9E8D57ull * c.z;
        u4 n;
        n.x = (unsigned)(p1 >> 32) ^ c.y ^ k0;
        n.y = (unsigned)p1;
        n.z = (unsigned)(p0 >> 32) ^ c.w ^ k1;
        n.w = (unsigned)p0;
        c = n;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return c;
}

__device__ __forceinline__ double u53(unsigned hi, unsigned lo) {
    return ((double)(hi >> 5) * 67108864.0 + (double)(lo >> 6)) * (1.0 / 9007199254740992.0);
}

// one thread per (row, column pair); also accumulates eta = x . beta_true per row when y != NULL
template <typename T>
__global__ void synth_features_kernel(unsigned long long seed, int64_t row0, int64_t n, int p, int kind,
                                      int ones_col, T* __restrict__ X, int64_t ldx) {
    const int npair = (p + 1) / 2;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n * npair) return;
    const int64_t r = idx / npair;
    const int jp = (int)(idx % npair);
    const unsigned long long gi = (unsigned long long)(row0 + r);
    u4 c; c.x = (unsigned)gi; c.y = (unsigned)(gi >> 32); c.z = (unsigned)jp; c.w = 0u;
    const u4 o = philox4x32_10(c, (unsigned)seed, 0u);
    const double ua = u53(o.x, o.y), ub = u53(o.z, o.w);
    double a, b;
    if (kind == 0) { a = ua - 0.5; b = ub - 0.5; }
    else {
        const double rad = sqrt(-2.0 * log(1.0 - ua)) * 0.28867513459481287;   // sd = sqrt(1/12)
        const double ang = 6.283185307179586 * ub;
        a = rad * cos(ang); b = rad * sin(ang);
    }
    T* row = X + r * ldx + (ones_col ? 1 : 0);
    row[2 * jp] = (T)a;
    if (2 * jp + 1 < p) row[2 * jp + 1] = (T)b;
    if (ones_col && jp == 0) X[r * ldx] = (T)1;
}

// y_i = 1{u_i < sigmoid(x_i . beta)} with u_i from counter (i_lo, i_hi, 0, 1), key (seed+1, 0).
// One wave per row; X already holds the features.
template <typename T>
__global__ void synth_labels_kernel(unsigned long long seed, int64_t row0, int64_t n, int pcols,
                                    const T* __restrict__ X, int64_t ldx, const T* __restrict__ beta,
                                    int p_true_ones, int first_col, T* __restrict__ y) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (r >= n) return;
    double s = 0.0;
    const T* row = X + r * ldx;
    if (beta) {
        for (int k = lane; k < pcols; k += 64) s += (double)row[k] * (double)beta[k];
    } else {
        // default truth: ones on the first p_true_ones feature columns (models.py:12,18-19)
        for (int k = lane; k < p_true_ones; k += 64) s += (double)row[first_col + k];
    }
    for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m, 64);
    if (lane == 0) {
        const unsigned long long gi = (unsigned long long)(row0 + r);
        u4 c; c.x = (unsigned)gi; c.y = (unsigned)(gi >> 32); c.z = 0u; c.w = 1u;
        const u4 o = philox4x32_10(c, (unsigned)(seed + 1ull), 0u);
        const double u = u53(o.x, o.y);
        const double prob = 1.0 / (1.0 + exp(-s));
        y[r] = (u < prob) ? (T)1 : (T)0;
    }
}

// Linear-model response (y = X beta + N(0, sigma^2)): y_i = x_i . beta + sigma z_i, z_i standard normal
// by Box-Muller on the uniforms of counter (i_lo, i_hi, 0, 2), key (seed+1, 0).  One wave per row; X already holds the features.
template <typename T>
__global__ void synth_response_kernel(unsigned long long seed, int64_t row0, int64_t n, int pcols,
                                      const T* __restrict__ X, int64_t ldx, const T* __restrict__ beta,
                                      int p_true_ones, int first_col, double sigma, T* __restrict__ y) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (r >= n) return;
    double s = 0.0;
    const T* row = X + r * ldx;
    if (beta) {
        for (int k = lane; k < pcols; k += 64) s += (double)row[k] * (double)beta[k];
    } else {
        for (int k = lane; k < p_true_ones; k += 64) s += (double)row[first_col + k];
    }
    s = wave_allreduce_sum(s);
    if (lane == 0) {
        const unsigned long long gi = (unsigned long long)(row0 + r);
        u4 c; c.x = (unsigned)gi; c.y = (unsigned)(gi >> 32); c.z = 0u; c.w = 2u;
        const u4 o = philox4x32_10(c, (unsigned)(seed + 1ull), 0u);
        const double ua = u53(o.x, o.y), ub = u53(o.z, o.w);
        const double z = sqrt(-2.0 * log(1.0 - ua)) * cos(6.283185307179586 * ub);
        y[r] = (T)(s + sigma * z);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// fp32-NATIVE linear rows (config 5's stream at its stated size): features AND response in one launch, every
// value formed in fp32.  The fp64-defined generator above costs 41 ms per 2^22 x 2000 chunk -- Philox for two columns per call
// and double-precision log / sqrt / sincos, all VALU time the matrix pipe cannot hide (the Gram of the same chunk takes 121 ms;
// run side by side on two streams the two kernels take exactly the sum of their times, bench/overlap_probe.py) -- and the
// response a second read of the chunk.  Definition (oracle/dlsa_oracle.py: synth_linear32):
//   row i, column quad q = j / 4: Philox-4x32-10 counter (i_lo, i_hi, q, 3), key (seed, 0) -> four 24-bit uniforms
//   u_k = (o_k >> 8) 2^-24;  columns 4q, 4q + 1 = r cos(2 pi u_1), r sin(2 pi u_1) with r = sqrt(-2 ln(1 - u_0)) sqrt(1/12);
//   columns 4q + 2, 4q + 3 likewise from (u_2, u_3);
//   y_i = sum_j x_ij beta_j + sigma sqrt(-2 ln(1 - v_0)) cos(2 pi v_1),  v from counter (i_lo, i_hi, 0, 4), key (seed + 1, 0).
// The logarithm, root, sine and cosine are the hardware's fp32 instructions (v_log_f32, v_sqrt_f32, v_sin_f32, v_cos_f32): the
// rows are a pure function of (seed, i) on this device, and agree with the oracle's numpy fp32 evaluation to ~1e-6 absolute.
// One wave per row at a time (grid-stride): lane l forms the quads l, l + 64, ... (16-byte stores, 1 KB per wave instruction)
// and keeps its share of x . beta; one wave reduction per row.
__device__ __forceinline__ float u24(unsigned o) { return (float)(o >> 8) * 5.9604644775390625e-08f; }
__device__ __forceinline__ void box_muller32(float u0, float u1, float scale, float& a, float& b) {
    // -2 ln(1 - u0) = -2 ln2 log2(1 - u0);  v_sin / v_cos take their argument in revolutions
    const float r = __builtin_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(1.0f - u0)) * scale;
    a = r * __builtin_amdgcn_cosf(u1);
    b = r * __builtin_amdgcn_sinf(u1);
}
__global__ __launch_bounds__(256) void synth_linear32_kernel(unsigned long long seed, int64_t row0, int64_t n, int p, int ones_col,
                                                             float* __restrict__ X, int64_t ldx, const float* __restrict__ beta,
                                                             int p_true_ones, float sigma, float* __restrict__ y) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
    const int nquad = (p + 3) / 4;
    const bool vec = !ones_col && (ldx % 4 == 0) && (((uintptr_t)X & 15) == 0);      // 16-byte stores when every quad is aligned
    for (int64_t r = wave; r < n; r += nwaves) {
        const unsigned long long gi = (unsigned long long)(row0 + r);
        float* row = X + r * ldx + (ones_col ? 1 : 0);
        float dot = 0.f;
        for (int q = lane; q < nquad; q += 64) {
            u4 c; c.x = (unsigned)gi; c.y = (unsigned)(gi >> 32); c.z = (unsigned)q; c.w = 3u;
            const u4 o = philox4x32_10(c, (unsigned)seed, 0u);
            float v[4];
            box_muller32(u24(o.x), u24(o.y), 0.28867513459481287f, v[0], v[1]);
            box_muller32(u24(o.z), u24(o.w), 0.28867513459481287f, v[2], v[3]);
            const int j = 4 * q;
            if (vec && j + 3 < p) *reinterpret_cast<float4*>(row + j) = make_float4(v[0], v[1], v[2], v[3]);
            else {
#pragma unroll
                for (int e = 0; e < 4; ++e) if (j + e < p) row[j + e] = v[e];
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (j + e < p) dot += beta ? v[e] * beta[(ones_col ? 1 : 0) + j + e] : (j + e < p_true_ones ? v[e] : 0.f);
        }
        if (ones_col && lane == 0) { X[r * ldx] = 1.f; if (beta) dot += beta[0]; }
        for (int m = 32; m >= 1; m >>= 1) dot += __shfl_xor(dot, m, 64);
        if (y && lane == 0) {
            u4 c; c.x = (unsigned)gi; c.y = (unsigned)(gi >> 32); c.z = 0u; c.w = 4u;
            const u4 o = philox4x32_10(c, (unsigned)(seed + 1ull), 0u);
            float z, unused;
            box_muller32(u24(o.x), u24(o.y), 1.0f, z, unused);
            y[r] = dot + sigma * z;
        }
    }
}

int synth_linear32_impl(uint64_t seed, int64_t row0, int64_t n, int p, int ones_col, float* X, int64_t ldx, const float* beta_true,
                        double sigma, float* y, hipStream_t s) {
    DLSA_REQUIRE(X, "synth_linear32: null X");
    DLSA_REQUIRE(p > 0 && n >= 0 && ldx >= p + (ones_col ? 1 : 0) && sigma >= 0.0, "synth_linear32: bad shape n=%lld p=%d ldx=%lld",
                 (long long)n, p, (long long)ldx);
    if (n == 0) return DLSA_OK;
    const int64_t blocks = std::min<int64_t>((n + 3) / 4, (int64_t)kNumCU * 32);       // 8 waves per SIMD, grid-stride over the rows
    hipLaunchKernelGGL(synth_linear32_kernel, dim3((unsigned)blocks), dim3(256), 0, s, (unsigned long long)seed, row0, n, p, ones_col, X, ldx,
                       beta_true, (int)(p * 0.4), (float)sigma, y);
    DLSA_HIP_CHECK(hipGetLastError());
    return DLSA_OK;
}

template <typename T>
int synth_response_impl(uint64_t seed, int64_t row0, int64_t n, int p, int ones_col, const T* X, int64_t ldx,
                        const T* beta_true, double sigma, T* y, hipStream_t s) {
    DLSA_REQUIRE(X && y, "synth_response: null X or y");
    DLSA_REQUIRE(p > 0 && n >= 0 && ldx >= p + (ones_col ? 1 : 0) && sigma >= 0.0, "synth_response: bad shape n=%lld p=%d ldx=%lld",
                 (long long)n, p, (long long)ldx);
    if (n == 0) return DLSA_OK;
    const int64_t lb = (n + 3) / 4;
    DLSA_REQUIRE(lb < (1ll << 31), "synth_response: too many rows for one launch; generate in chunks");
    hipLaunchKernelGGL((synth_response_kernel<T>), dim3((unsigned)lb), dim3(256), 0, s, (unsigned long long)seed, row0, n,
                       p + (ones_col ? 1 : 0), X, ldx, beta_true, (int)(p * 0.4), ones_col ? 1 : 0, sigma, y);
    DLSA_HIP_CHECK(hipGetLastError());
    return DLSA_OK;
}

template <typename T>
int synth_impl(uint64_t seed, int64_t row0, int64_t n, int p, int kind, int ones_col, T* X, int64_t ldx,
               T* y, const T* beta_true, hipStream_t s) {
    DLSA_REQUIRE(X, "synth: null X");
    DLSA_REQUIRE(p > 0 && n >= 0 && ldx >= p + (ones_col ? 1 : 0), "synth: bad shape n=%lld p=%d ldx=%lld",
                 (long long)n, p, (long long)ldx);
    DLSA_REQUIRE(kind == 0 || kind == 1, "synth: kind must be 0 (uniform) or 1 (gaussian)");
    if (n == 0) return DLSA_OK;
    const int npair = (p + 1) / 2;
    const int64_t total = n * npair;
    const int64_t blocks = (total + 255) / 256;
    DLSA_REQUIRE(blocks < (1ll << 31), "synth: too many elements for one launch; generate in chunks");
    hipLaunchKernelGGL((synth_features_kernel<T>), dim3((unsigned)blocks), dim3(256), 0, s,
                       (unsigned long long)seed, row0, n, p, kind, ones_col, X, ldx);
    DLSA_HIP_CHECK(hipGetLastError());
    if (y) {
        const int64_t lb = (n + 3) / 4;
        DLSA_REQUIRE(lb < (1ll << 31), "synth: too many rows for one launch; generate in chunks");
        hipLaunchKernelGGL((synth_labels_kernel<T>), dim3((unsigned)lb), dim3(256), 0, s,
                           (unsigned long long)seed, row0, n, p + (ones_col ? 1 : 0), (const T*)X, ldx, beta_true,
                           (int)(p * 0.4), ones_col ? 1 : 0, y);
        DLSA_HIP_CHECK(hipGetLastError());
    }
    return DLSA_OK;
}

}  // namespace dlsa

extern "C" {
int dlsa_synth_f64(uint64_t seed, int64_t row0, int64_t n, int p, int kind, int ones_col, double* X,
                   int64_t ldx, double* y, const double* beta_true, void* stream) {
    return dlsa::synth_impl<double>(seed, row0, n, p, kind, ones_col, X, ldx, y, beta_true, (hipStream_t)stream);
}
int dlsa_synth_f32(uint64_t seed, int64_t row0, int64_t n, int p, int kind, int ones_col, float* X,
                   int64_t ldx, float* y, const float* beta_true, void* stream) {
    return dlsa::synth_impl<float>(seed, row0, n, p, kind, ones_col, X, ldx, y, beta_true, (hipStream_t)stream);
}
int dlsa_synth_response_f64(uint64_t seed, int64_t row0, int64_t n, int p, int ones_col, const double* X, int64_t ldx,
                            const double* beta_true, double sigma, double* y, void* stream) {
    return dlsa::synth_response_impl<double>(seed, row0, n, p, ones_col, X, ldx, beta_true, sigma, y, (hipStream_t)stream);
}
int dlsa_synth_response_f32(uint64_t seed, int64_t row0, int64_t n, int p, int ones_col, const float* X, int64_t ldx,
                            const float* beta_true, double sigma, float* y, void* stream) {
    return dlsa::synth_response_impl<float>(seed, row0, n, p, ones_col, X, ldx, beta_true, sigma, y, (hipStream_t)stream);
}
int dlsa_synth_linear_f32(uint64_t seed, int64_t row0, int64_t n, int p, int ones_col, float* X, int64_t ldx, const float* beta_true,
                          double sigma, float* y, void* stream) {
    return dlsa::synth_linear32_impl(seed, row0, n, p, ones_col, X, ldx, beta_true, sigma, y, (hipStream_t)stream);
}
}
