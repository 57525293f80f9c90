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
}  // namespace dlsa

extern "C" {
int dlsa_version(void) { return 100; }
int dlsa_last_error(char* buf, int len) {
    if (!buf || len <= 0) return DLSA_ERR_INVALID;
    strncpy(buf, dlsa::g_err, (size_t)len - 1);
    buf[len - 1] = 0;
    return DLSA_OK;
}
}
