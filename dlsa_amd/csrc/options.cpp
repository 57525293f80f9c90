// dlsa_irls_options: per-thread options of the IRLS driver (see options.h).
#include "options.h"
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

namespace dlsa {
void set_error(const char* fmt, ...);

static thread_local dlsa_irls_options g_opt = {0, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1.0};
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
    DLSA_KNOB("DLSA_IRLS_OWN_HESSIAN", own_hessian), DLSA_KNOB("DLSA_IRLS_POOLED_START", pooled_start),
    DLSA_KNOB("DLSA_IRLS_GRAD_PASSES", grad_passes),
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

dlsa_irls_options irls_options_snapshot() { return g_opt_set ? g_opt : dlsa_irls_options{0, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1.0}; }
void irls_options_adopt(const dlsa_irls_options& o) { g_opt = o; g_opt_set = o.struct_bytes != 0; }

// ---- kernel switches (dlsa_kernel_options)
static const dlsa_kernel_options kKernelAuto = {0, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1};
static thread_local dlsa_kernel_options g_kopt = kKernelAuto;
static thread_local bool g_kopt_set = false;
#define DLSA_KKNOB(env, field) {env, offsetof(dlsa_kernel_options, field)}
static const KnobField kKernelFields[] = {
    DLSA_KKNOB("DLSA_LARS_Q", lars_q),             DLSA_KKNOB("DLSA_LARS_Q_WGS", lars_q_wgs),     DLSA_KKNOB("DLSA_LARS_Q_THREADS", lars_q_threads),
    DLSA_KKNOB("DLSA_LARS_Q_LDS", lars_q_lds),     DLSA_KKNOB("DLSA_LARS_WGS", lars_wgs),         DLSA_KKNOB("DLSA_LARS_THREADS", lars_threads),
    DLSA_KKNOB("DLSA_LOGIT_RING", logit_ring),     DLSA_KKNOB("DLSA_CHOL_SMALL", chol_small),     DLSA_KKNOB("DLSA_GRAM_WIDE_F32", gram_wide_f32),
    DLSA_KKNOB("DLSA_OH_ORDERED", onehot_ordered), DLSA_KKNOB("DLSA_GRAM_DBG", gram_variant),     DLSA_KKNOB("DLSA_COOPERATIVE", cooperative),
};
#undef DLSA_KKNOB
constexpr int kNumKernelFields = (int)(sizeof(kKernelFields) / sizeof(kKernelFields[0]));

const char* kernel_knob(const char* env_name) {
    static thread_local char text[kNumKernelFields][16];
    if (g_kopt_set)
        for (int i = 0; i < kNumKernelFields; ++i)
            if (!strcmp(kKernelFields[i].env, env_name)) {
                const int v = *(const int*)((const char*)&g_kopt + kKernelFields[i].off);
                if (v < 0) break;
                snprintf(text[i], sizeof text[i], "%d", v);
                return text[i];
            }
#ifdef DLSA_DEBUG_KNOBS
    return getenv(env_name);        // experiment builds only (make knobs): the A/B scripts under bench/
#else
    return nullptr;
#endif
}

hipError_t launch_cooperative(const void* func, dim3 grid, dim3 block, void** args, size_t shm, hipStream_t stream) {
    // OPT-IN (dlsa_kernel_options.cooperative = 1).  Measured on this runtime (round 3, again in round 5: bench/coop_streams.py): every
    // HIP stream created AFTER a process's first cooperative launch is serialised with the others -- the partition chains of a later
    // fit lose their overlap -- and a cooperative launch costs 15-19 us of host time.  The plain launch's barriers are bounded and its
    // give-up path reruns on one workgroup, so the default stays the plain launch.
    const char* e = kernel_knob("DLSA_COOPERATIVE");
    if (!e || atoi(e) == 0) return hipErrorNotSupported;
    const hipError_t rc = hipLaunchCooperativeKernel(func, grid, block, args, (unsigned)shm, stream);
    if (rc != hipSuccess) (void)hipGetLastError();        // (refused: the sticky error must not fail the plain launch that follows)
    return rc;
}

dlsa_kernel_options kernel_options_snapshot() { return g_kopt_set ? g_kopt : kKernelAuto; }
void kernel_options_adopt(const dlsa_kernel_options& o) { g_kopt = o; g_kopt_set = o.struct_bytes != 0; }

}  // namespace dlsa

extern "C" {

void dlsa_kernel_options_init(dlsa_kernel_options* o) {
    if (!o) return;
    *o = dlsa::kKernelAuto;
    o->struct_bytes = (int)sizeof(dlsa_kernel_options);
}

int dlsa_kernel_set_options(const dlsa_kernel_options* o) {
    if (!o) { dlsa::g_kopt_set = false; return DLSA_OK; }
    if (o->struct_bytes != (int)sizeof(dlsa_kernel_options)) {
        dlsa::set_error("dlsa_kernel_set_options: struct_bytes %d, this library's dlsa_kernel_options has %d (call dlsa_kernel_options_init first)",
                        o->struct_bytes, (int)sizeof(dlsa_kernel_options));
        return DLSA_ERR_INVALID;
    }
    dlsa::g_kopt = *o;
    dlsa::g_kopt_set = true;
    return DLSA_OK;
}


void dlsa_irls_options_init(dlsa_irls_options* o) {
    if (!o) return;
    *o = dlsa_irls_options{(int)sizeof(dlsa_irls_options), -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1.0};
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
