// Shared helpers for libdlsa_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include "../../include/dlsa_hip.h"

namespace dlsa {

void set_error(const char* fmt, ...);
// Which Gram kernel the calling thread's last Gram call dispatched (dlsa_gram_last_kernel), and where that launch's clock
// probe lands in device memory: wave 0 of workgroup 0 stores its s_memtime delta (shader cycles from its first to its last
// instruction) there; null for kernels without the probe.
void note_gram_kernel(const void* clk_dev, hipStream_t stream, const char* fmt, ...);
constexpr size_t kGramProbeBytes = 256;       // the probe's slot at the end of a Gram workspace

#define DLSA_HIP_CHECK(expr)                                                              \
    do {                                                                                  \
        hipError_t _e = (expr);                                                           \
        if (_e != hipSuccess) {                                                           \
            dlsa::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),        \
                            __FILE__, __LINE__);                                          \
            return DLSA_ERR_HIP;                                                          \
        }                                                                                 \
    } while (0)

#define DLSA_REQUIRE(cond, ...)                                                           \
    do {                                                                                  \
        if (!(cond)) {                                                                    \
            dlsa::set_error(__VA_ARGS__);                                                 \
            return DLSA_ERR_INVALID;                                                      \
        }                                                                                 \
    } while (0)

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// bump allocator over the caller's workspace
struct Arena {
    char* base;
    size_t size;
    size_t off;
    Arena(void* p, size_t n) : base((char*)p), size(n), off(0) {}
    void* take(size_t bytes) {
        size_t o = align_up(off, 256);
        if (o + bytes > size) return nullptr;
        off = o + bytes;
        return base + o;
    }
};

// ---------------------------------------------------------------------------------------------------------------
// Wave-wide fp64 exchanges and reductions WITHOUT LDS round trips (device code; lane maps probed on the box with
// bench/probe_permlane.hip):  xor 1, 2: DPP quad_perm;  xor 4: row_half_mirror then quad_perm [3,2,1,0];
// xor 8: row_ror:8;  xor 16 / 32: v_permlane16_swap / v_permlane32_swap (gfx950) -- swap(a, b) returns (a', b')
// that hold, in both halves, {the value this lane keeps, its partner's copy of it}.  __shfl_xor compiles to
// ds_bpermute: an LDS round trip per step, six dependent ones per reduction.
// ---------------------------------------------------------------------------------------------------------------
#if defined(__HIPCC__)
template <int CTRL>
__device__ __forceinline__ double dpp_mov_f64(double v) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    // old = 0 with bound_ctrl: every lane is written (full masks, in-range sources), and hipcc then emits the bare v_mov_b32_dpp;
    // with old = the source it copied the register first (v_mov + s_nop + v_mov_dpp per dword)
    return __hiloint2double(__builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true),
                            __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true));
}
template <int M>
__device__ __forceinline__ double dpp_xor_f64(double v) {      // value of lane ^ M, M in {1, 2, 4, 8}
    if constexpr (M == 1) return dpp_mov_f64<0xB1>(v);
    else if constexpr (M == 2) return dpp_mov_f64<0x4E>(v);
    else if constexpr (M == 4) return dpp_mov_f64<0x1B>(dpp_mov_f64<0x141>(v));
    else return dpp_mov_f64<0x128>(v);
}
// permlane swap of two doubles at distance M (16 or 32): a', b' as described above
template <int M>
__device__ __forceinline__ void swap_f64(double a, double b, double& a2, double& b2) {
    const int alo = __double2loint(a), ahi = __double2hiint(a), blo = __double2loint(b), bhi = __double2hiint(b);
    if constexpr (M == 32) {
        const auto x = __builtin_amdgcn_permlane32_swap(alo, blo, false, false);
        const auto y = __builtin_amdgcn_permlane32_swap(ahi, bhi, false, false);
        a2 = __hiloint2double(y[0], x[0]); b2 = __hiloint2double(y[1], x[1]);
    } else {
        const auto x = __builtin_amdgcn_permlane16_swap(alo, blo, false, false);
        const auto y = __builtin_amdgcn_permlane16_swap(ahi, bhi, false, false);
        a2 = __hiloint2double(y[0], x[0]); b2 = __hiloint2double(y[1], x[1]);
    }
}
struct WaveOpSum { __device__ __forceinline__ double operator()(double a, double b) const { return a + b; } };
struct WaveOpMax { __device__ __forceinline__ double operator()(double a, double b) const { return fmax(a, b); } };
struct WaveOpMin { __device__ __forceinline__ double operator()(double a, double b) const { return fmin(a, b); } };
template <class Op>
__device__ __forceinline__ double wave_allreduce(double s, Op op) {      // same pairing order as the xor butterfly 32..1
    double a, b;
    swap_f64<32>(s, s, a, b); s = op(a, b);
    swap_f64<16>(s, s, a, b); s = op(a, b);
    s = op(s, dpp_xor_f64<8>(s));
    s = op(s, dpp_xor_f64<4>(s));
    s = op(s, dpp_xor_f64<2>(s));
    s = op(s, dpp_xor_f64<1>(s));
    return s;
}
__device__ __forceinline__ double wave_allreduce_sum(double s) { return wave_allreduce(s, WaveOpSum()); }
__device__ __forceinline__ double wave_allreduce_max(double s) { return wave_allreduce(s, WaveOpMax()); }
__device__ __forceinline__ double wave_allreduce_min(double s) { return wave_allreduce(s, WaveOpMin()); }
#endif

// Timing-experiment knobs that produce WRONG RESULTS (DLSA_GRAM_DBG bits 1 / 16 / 128, DLSA_OH_DBG) exist only in
// builds made with -DDLSA_DEBUG_KNOBS (bench/ experiment builds; never `make`): in the shipped library the tests
// below are the constant 0 and the environment cannot change a result.  The valid-result A/B switches
// (DLSA_GRAM_DBG 2 / 4 / 8 / 32 / 64 / 256, DLSA_GRAM_NOWIDE, DLSA_IRLS_*, DLSA_LARS_WGS) stay runtime switches.
#ifdef DLSA_DEBUG_KNOBS
#define DLSA_DBG_WRONG(mask, bit) ((mask) & (bit))
#else
#define DLSA_DBG_WRONG(mask, bit) 0
#endif
constexpr int kGramDbgValidBits = 2 | 4 | 8 | 32 | 64 | 256;
const char* kernel_knob(const char* env_name);      // options.cpp: dlsa_kernel_options (the environment only in DLSA_DEBUG_KNOBS builds)
static inline int gram_dbg_env() {
    const char* e = kernel_knob("DLSA_GRAM_DBG");
    const int v = e ? atoi(e) : 0;
#ifdef DLSA_DEBUG_KNOBS
    return v;
#else
    return v & kGramDbgValidBits;
#endif
}

// a block from the stream-ordered pool that goes back on every way out of the scope (the error returns of DLSA_HIP_CHECK included)
struct PoolBlock {
    char* p = nullptr;
    hipStream_t s = nullptr;
    PoolBlock() = default;
    PoolBlock(const PoolBlock&) = delete;
    PoolBlock& operator=(const PoolBlock&) = delete;
    ~PoolBlock() { release(); }
    hipError_t alloc(size_t bytes, hipStream_t stream) { release(); s = stream; return hipMallocAsync((void**)&p, bytes, stream); }
    void release() { if (p) { (void)hipFreeAsync(p, s); p = nullptr; } }
};

constexpr int kNumXCD = 8;
constexpr int kNumCU = 256;
constexpr int kLdsBytes = 160 * 1024;     // LDS per CU (one workgroup may take all of it)

}  // namespace dlsa
