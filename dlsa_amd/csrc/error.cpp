// Thread-local error text + version for libdlsa_hip.so
#include "common.h"
#include <stdarg.h>
#include <string.h>

namespace dlsa {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
static thread_local char g_kernel[160] = "";
static thread_local const void* g_clk = nullptr;
static thread_local hipStream_t g_clk_stream = nullptr;
void note_gram_kernel(const void* clk_dev, hipStream_t stream, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_kernel, sizeof(g_kernel), fmt, ap);
    va_end(ap);
    g_clk = clk_dev;
    g_clk_stream = stream;
}
}  // namespace dlsa

extern "C" {
int dlsa_version(void) { return 100; }
int dlsa_gram_last_kernel(char* name, int len, uint64_t* shader_cycles) {
    if (!name || len <= 0) return DLSA_ERR_INVALID;
    strncpy(name, dlsa::g_kernel, (size_t)len - 1);
    name[len - 1] = 0;
    if (shader_cycles) {
        *shader_cycles = 0;
        if (dlsa::g_clk) {       // waits for the launch (the stream it was enqueued on), then reads the 8 bytes back
            DLSA_HIP_CHECK(hipStreamSynchronize(dlsa::g_clk_stream));
            DLSA_HIP_CHECK(hipMemcpy(shader_cycles, dlsa::g_clk, 8, hipMemcpyDeviceToHost));
        }
    }
    return DLSA_OK;
}
int dlsa_last_error(char* buf, int len) {
    if (!buf || len <= 0) return DLSA_ERR_INVALID;
    strncpy(buf, dlsa::g_err, (size_t)len - 1);
    buf[len - 1] = 0;
    return DLSA_OK;
}
}
