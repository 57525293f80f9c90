// Shared helpers for libdlsa_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <stdio.h>
#include "../../include/dlsa_hip.h"

namespace dlsa {

void set_error(const char* fmt, ...);

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

constexpr int kNumXCD = 8;
constexpr int kNumCU = 256;

}  // namespace dlsa
