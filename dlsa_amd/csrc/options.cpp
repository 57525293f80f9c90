// dlsa_irls_options: per-thread options of the IRLS driver (see options.h).
#include "options.h"
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

namespace dlsa {
void set_error(const char* fmt, ...);

static thread_local dlsa_irls_options g_opt = {0, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1.0};
static thread_local bool g_opt_set = false;

struct KnobField { const char* env; size_t off; };
#define DLSA_KNOB(env, field) {env, offsetof(dlsa_irls_options, field)}
static const KnobField kIntFields[] = {
    DLSA_KNOB("DLSA_IRLS_CHAINS", chains),       DLSA_KNOB("DLSA_IRLS_SEED", seeded),         DLSA_KNOB("DLSA_IRLS_SUBSAMPLE", subsample_div),
    DLSA_KNOB("DLSA_IRLS_FACTOR_DIV", factor_div), DLSA_KNOB("DLSA_IRLS_WARM", warm),         DLSA_KNOB("DLSA_IRLS_INHERIT", inherit),
    DLSA_KNOB("DLSA_IRLS_POOL", pool),           DLSA_KNOB("DLSA_IRLS_SECANT", secant),       DLSA_KNOB("DLSA_IRLS_INVERSE", inverse),
    DLSA_KNOB("DLSA_IRLS_PREDICT", predict),     DLSA_KNOB("DLSA_IRLS_FUSED", fused),         DLSA_KNOB("DLSA_IRLS_FUSE_LAST", fuse_last),
    DLSA_KNOB("DLSA_IRLS_SMALL", small),         DLSA_KNOB("DLSA_QN_THREADS", qn_threads),    DLSA_KNOB("DLSA_IRLS_TRACE", trace),
    DLSA_KNOB("DLSA_IRLS_BATCHED", batched),     DLSA_KNOB("DLSA_IRLS_LEAN", lean),           DLSA_KNOB("DLSA_IRLS_SMALL_CLUSTER", small_cluster),
    DLSA_KNOB("DLSA_IRLS_OWN_HESSIAN", own_hessian),
};
#undef DLSA_KNOB
constexpr int kNumIntFields = (int)(sizeof(kIntFields) / sizeof(kIntFields[0]));

const char* knob(const char* env_name) {
    static thread_local char text[kNumIntFields + 1][32];
    if (g_opt_set) {
        for (int i = 0; i < kNumIntFields; ++i)
            if (!strcmp(kIntFields[i].env, env_name)) {
                const int v = *(const int*)((const char*)&g_opt + kIntFields[i].off);
                if (v < 0) break;
                snprintf(text[i], sizeof text[i], "%d", v);
                return text[i];
            }
        if (!strcmp(env_name, "DLSA_IRLS_FREEZE") && g_opt.freeze_at >= 0.0) {
            snprintf(text[kNumIntFields], sizeof text[kNumIntFields], "%.17g", g_opt.freeze_at);
            return text[kNumIntFields];
        }
    }
    return getenv(env_name);
}

dlsa_irls_options irls_options_snapshot() { return g_opt_set ? g_opt : dlsa_irls_options{0, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1.0}; }
void irls_options_adopt(const dlsa_irls_options& o) { g_opt = o; g_opt_set = o.struct_bytes != 0; }

}  // namespace dlsa

extern "C" {

void dlsa_irls_options_init(dlsa_irls_options* o) {
    if (!o) return;
    *o = dlsa_irls_options{(int)sizeof(dlsa_irls_options), -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1.0};
}

int dlsa_irls_set_options(const dlsa_irls_options* o) {
    if (!o) { dlsa::g_opt_set = false; return DLSA_OK; }
    if (o->struct_bytes != (int)sizeof(dlsa_irls_options)) {
        dlsa::set_error("dlsa_irls_set_options: struct_bytes %d, this library's dlsa_irls_options has %d (call dlsa_irls_options_init first)",
                        o->struct_bytes, (int)sizeof(dlsa_irls_options));
        return DLSA_ERR_INVALID;
    }
    if (o->chains == 0 || o->chains > 8) {
        dlsa::set_error("dlsa_irls_set_options: chains must be -1 (automatic) or 1..8, got %d", o->chains);
        return DLSA_ERR_INVALID;
    }
    dlsa::g_opt = *o;
    dlsa::g_opt_set = true;
    return DLSA_OK;
}

}  // extern "C"
