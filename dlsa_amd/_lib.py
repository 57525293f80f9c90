"""ctypes binding of libdlsa_hip.so (the C ABI declared in include/dlsa_hip.h).

There is no CPU fallback: importing the engine without the built library, or calling it
without a GPU, raises.  Build with `make` (or `python -c "import __graft_entry__ as g; g.build()"`).
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libdlsa_hip.so")

c_i64 = ctypes.c_int64
c_int = ctypes.c_int
c_sz = ctypes.c_size_t
c_dbl = ctypes.c_double
c_vp = ctypes.c_void_p
c_u64 = ctypes.c_uint64

# name -> (restype, argtypes); every symbol include/dlsa_hip.h declares
class IrlsOptionsC(ctypes.Structure):
    """include/dlsa_hip.h: dlsa_irls_options (field for field)"""
    _fields_ = [(n, c_int) for n in ("struct_bytes", "chains", "seeded", "subsample_div", "factor_div", "warm", "inherit", "pool", "secant",
                                     "inverse", "predict", "fused", "fuse_last", "small", "batched", "qn_threads", "trace", "lean", "small_cluster", "own_hessian", "pooled_start", "grad_passes")] + [("freeze_at", c_dbl)]


class KernelOptionsC(ctypes.Structure):
    """include/dlsa_hip.h: dlsa_kernel_options (field for field)"""
    _fields_ = [(n, c_int) for n in ("struct_bytes", "lars_q", "lars_q_wgs", "lars_q_threads", "lars_q_lds", "lars_wgs", "lars_threads", "logit_ring",
                                     "chol_small", "gram_wide_f32", "onehot_ordered", "gram_variant", "cooperative")]


SIGNATURES = {
    "dlsa_version": (c_int, []),
    "dlsa_irls_last_fit_path": (c_int, []),
    "dlsa_irls_options_init": (None, [ctypes.POINTER(IrlsOptionsC)]),
    "dlsa_irls_set_options": (c_int, [ctypes.POINTER(IrlsOptionsC)]),
    "dlsa_kernel_options_init": (None, [ctypes.POINTER(KernelOptionsC)]),
    "dlsa_kernel_set_options": (c_int, [ctypes.POINTER(KernelOptionsC)]),
    "dlsa_last_error": (c_int, [ctypes.c_char_p, c_int]),
    "dlsa_synth_f64": (c_int, [c_u64, c_i64, c_i64, c_int, c_int, c_int, c_vp, c_i64, c_vp, c_vp, c_vp]),
    "dlsa_synth_f32": (c_int, [c_u64, c_i64, c_i64, c_int, c_int, c_int, c_vp, c_i64, c_vp, c_vp, c_vp]),
    "dlsa_synth_response_f64": (c_int, [c_u64, c_i64, c_i64, c_int, c_int, c_vp, c_i64, c_vp, c_dbl, c_vp, c_vp]),
    "dlsa_synth_response_f32": (c_int, [c_u64, c_i64, c_i64, c_int, c_int, c_vp, c_i64, c_vp, c_dbl, c_vp, c_vp]),
    "dlsa_synth_linear_f32": (c_int, [c_u64, c_i64, c_i64, c_int, c_int, c_vp, c_i64, c_vp, c_dbl, c_vp, c_vp]),
    "dlsa_gram_workspace_bytes": (c_sz, [c_i64, c_int, c_int]),
    "dlsa_gram_f64": (c_int, [c_vp, c_i64, c_vp, c_i64, c_int, c_vp, c_i64, c_int, c_vp, c_sz, c_vp]),
    "dlsa_gram_f32": (c_int, [c_vp, c_i64, c_vp, c_i64, c_int, c_vp, c_i64, c_int, c_vp, c_sz, c_vp]),
    "dlsa_gram_f32_acc64": (c_int, [c_vp, c_i64, c_vp, c_i64, c_int, c_vp, c_i64, c_int, c_vp, c_sz, c_vp]),
    "dlsa_xtv_stats_workspace_bytes": (c_sz, [c_int, c_int]),
    "dlsa_xtv_stats_f64": (c_int, [c_vp, c_i64, c_vp, c_i64, c_int, c_vp, c_vp, c_vp, c_int, c_vp, c_sz, c_vp]),
    "dlsa_xtv_stats_f32": (c_int, [c_vp, c_i64, c_vp, c_i64, c_int, c_vp, c_vp, c_vp, c_int, c_vp, c_sz, c_vp]),
    "dlsa_gram_last_kernel": (c_int, [ctypes.c_char_p, c_int, ctypes.POINTER(c_u64)]),
    "dlsa_logit_workspace_bytes": (c_sz, [c_i64, c_int]),
    "dlsa_logit_pass_f64": (c_int, [c_vp, c_i64, c_vp, c_vp, c_i64, c_int, c_vp, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "dlsa_loglik_f64": (c_int, [c_vp, c_i64, c_vp, c_i64, c_int, c_vp, c_i64, c_int, c_vp, c_vp, c_sz, c_vp]),
    "dlsa_xtv_f64": (c_int, [c_vp, c_i64, c_vp, c_i64, c_int, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "dlsa_xtv_f32": (c_int, [c_vp, c_i64, c_vp, c_i64, c_int, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "dlsa_irls_pass_workspace_bytes": (c_sz, [c_i64, c_int]),
    "dlsa_irls_pass_f64": (c_int, [c_vp, c_i64, c_vp, c_vp, c_i64, c_int, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "dlsa_irls_workspace_bytes": (c_sz, [c_i64, c_int]),
    "dlsa_irls_fit_f64": (c_int, [c_vp, c_i64, c_vp, ctypes.POINTER(c_i64), c_int, c_int, c_dbl, c_int,
                                  c_vp, c_vp, c_vp, ctypes.POINTER(c_int), ctypes.POINTER(c_int),
                                  ctypes.POINTER(c_dbl), c_vp, c_sz, c_vp]),
    "dlsa_logit_pass_icpt_f64": (c_int, [c_vp, c_i64, c_vp, c_vp, c_i64, c_int, c_vp, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "dlsa_loglik_icpt_f64": (c_int, [c_vp, c_i64, c_vp, c_i64, c_int, c_vp, c_i64, c_int, c_vp, c_vp, c_sz, c_vp]),
    "dlsa_newton_wide_workspace_bytes": (c_sz, [c_i64, c_int, c_int]),
    "dlsa_newton_wide_eligible": (c_int, [c_vp, c_i64, c_i64, c_int, c_int]),
    "dlsa_newton_wide_pass_f64": (c_int, [c_vp, c_i64, c_vp, c_vp, c_i64, c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_sz, c_vp]),
    "dlsa_gram_icpt_workspace_bytes": (c_sz, [c_i64, c_int]),
    "dlsa_gram_icpt_f64": (c_int, [c_vp, c_i64, c_vp, c_i64, c_int, c_vp, c_i64, c_vp, c_sz, c_vp]),
    "dlsa_irls_ex_workspace_bytes": (c_sz, [c_i64, c_int, c_int, c_i64]),
    "dlsa_irls_fit_ex_f64": (c_int, [c_vp, c_i64, c_vp, ctypes.POINTER(c_i64), ctypes.POINTER(c_i64), c_i64, c_int, c_int, c_int,
                                     c_dbl, c_int, c_vp, c_vp, c_vp, ctypes.POINTER(c_int), ctypes.POINTER(c_int),
                                     ctypes.POINTER(c_dbl), c_vp, c_sz, c_vp]),
    "dlsa_sum_blocks_f64": (c_int, [c_vp, c_vp, c_vp, c_int, c_int, ctypes.POINTER(c_int), c_vp, c_vp]),
    "dlsa_comm_unique_id": (c_int, [ctypes.c_char_p]),
    "dlsa_comm_init_rank": (c_int, [ctypes.POINTER(c_vp), c_int, ctypes.c_char_p, c_int]),
    "dlsa_comm_destroy": (c_int, [c_vp]),
    "dlsa_allreduce_f64": (c_int, [c_vp, c_vp, c_i64, c_vp]),
    "dlsa_solve_workspace_bytes": (c_sz, [c_int]),
    "dlsa_spd_solve_f64": (c_int, [c_vp, c_i64, c_vp, c_int, c_vp, c_vp, c_sz, c_vp]),
    "dlsa_wls_solve_workspace_bytes": (c_sz, [c_int]),
    "dlsa_wls_solve_f64": (c_int, [c_vp, c_i64, c_vp, c_int, c_vp, ctypes.POINTER(c_int), c_vp, c_sz, c_vp]),
    "dlsa_sym_pinv_workspace_bytes": (c_sz, [c_int]),
    "dlsa_sym_pinv_solve_f64": (c_int, [c_vp, c_i64, c_vp, c_int, c_dbl, c_vp, ctypes.POINTER(c_int), ctypes.POINTER(c_dbl),
                                        c_vp, c_sz, c_vp]),
    "dlsa_lars_workspace_bytes": (c_sz, [c_int]),
    "dlsa_lars_lsa_f64": (c_int, [c_vp, c_i64, c_vp, c_int, c_int, c_dbl, c_int, c_dbl, c_int,
                                  c_vp, c_vp, c_vp, c_vp, ctypes.POINTER(c_int), c_vp, c_sz, c_vp]),
    "dlsa_lars_grid_barrier_timeout": (c_int, [c_dbl]),
    "dlsa_irls_small_cluster_timeout": (c_int, [c_dbl]),
    "dlsa_design_f64": (c_int, [c_vp, c_i64, c_int, c_vp, c_i64, c_int, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_int,
                                c_vp, c_i64, c_vp, c_vp]),
    "dlsa_design_f32": (c_int, [c_vp, c_i64, c_int, c_vp, c_i64, c_int, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_int,
                                c_vp, c_i64, c_vp, c_vp]),
    "dlsa_onehot_plan_create": (c_int, [c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_vp, c_vp, ctypes.POINTER(c_vp)]),
    "dlsa_onehot_plan_destroy": (None, [c_vp]),
    "dlsa_onehot_plan_roles": (c_int, [c_vp]),
    "dlsa_onehot_workspace_bytes": (c_sz, [c_vp, c_i64]),
    "dlsa_onehot_logit_pass_f64": (c_int, [c_vp, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "dlsa_onehot_gram_f64": (c_int, [c_vp, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_sz, c_vp]),
    "dlsa_onehot_irls_workspace_bytes": (c_sz, [c_vp, c_i64]),
    "dlsa_onehot_irls_fit_f64": (c_int, [c_vp, c_vp, c_i64, c_vp, c_i64, c_vp, ctypes.POINTER(c_i64), c_int, c_dbl, c_int,
                                         c_vp, c_vp, c_vp, ctypes.POINTER(c_int), ctypes.POINTER(c_int),
                                         ctypes.POINTER(c_dbl), c_vp, c_sz, c_vp]),
    "dlsa_onehot_irls_ex_workspace_bytes": (c_sz, [c_vp, c_i64, c_i64]),
    "dlsa_onehot_irls_fit_ex_f64": (c_int, [c_vp, c_vp, c_i64, c_vp, c_i64, c_vp, ctypes.POINTER(c_i64), ctypes.POINTER(c_i64), c_i64,
                                            c_int, c_dbl, c_int, c_vp, c_vp, c_vp, ctypes.POINTER(c_int), ctypes.POINTER(c_int),
                                            ctypes.POINTER(c_dbl), c_vp, c_sz, c_vp]),
    "dlsa_gram_plan_check": (c_int, [c_int, ctypes.POINTER(c_int), ctypes.POINTER(c_int), ctypes.POINTER(c_int)]),
    "dlsa_gram_wide_plan_check": (c_int, [c_int, ctypes.POINTER(c_int), ctypes.POINTER(c_int), ctypes.POINTER(c_int)]),
}

_lib = None


class DlsaError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libdlsa_hip status %d: %s" % (code, msg))
        self.code = code


def load():
    """Load the shared library (no GPU needed to load it or to resolve symbols)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("%s is missing: the HIP extension has not been built (run `make`). "
                          "dlsa_amd has no CPU fallback." % LIB_PATH)
    # torch first: its bundled HIP runtime must be the one libdlsa_hip.so binds to, so that device
    # pointers and streams are shared (loading the system libamdhip64 first gives two runtimes).
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)       # AttributeError here = header/library mismatch
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def last_error():
    buf = ctypes.create_string_buffer(512)
    load().dlsa_last_error(buf, 512)
    return buf.value.decode("utf-8", "replace")


def check(code):
    if code != 0:
        raise DlsaError(code, last_error())
