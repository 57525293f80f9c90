"""TEST INFRASTRUCTURE ONLY -- ctypes front of oracle/csrc/oracle_synth.c (the C restatement of the oracle's seeded
row generator).  Falls back to the numpy generator of oracle/dlsa_oracle.py when the library has not been built
(`__graft_entry__.build()` builds it).  Uniform rows are bit-identical to the numpy / HIP generators; Gaussian rows
agree to the last ulp of libm's log / sin / cos (tests/test_oracle_golden.py)."""
import ctypes
import os

import numpy as np

from . import dlsa_oracle as orc

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_build", "liboracle_synth.so")
SRC_PATH = os.path.join(_HERE, "csrc", "oracle_synth.c")
_lib = None


def build(force=False):
    """gcc -O2 -fPIC -shared -fopenmp oracle/csrc/oracle_synth.c (seconds; no GPU involved)."""
    import subprocess
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < os.path.getmtime(SRC_PATH):
        os.makedirs(os.path.dirname(LIB_PATH), exist_ok=True)
        subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-fopenmp", SRC_PATH, "-o", LIB_PATH, "-lm"])
    return LIB_PATH


def load():
    global _lib
    if _lib is None and os.path.exists(LIB_PATH):
        lib = ctypes.CDLL(LIB_PATH)
        lib.oracle_synth_features.restype = None
        lib.oracle_synth_features.argtypes = [ctypes.c_uint64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int, ctypes.c_int,
                                              ctypes.c_void_p, ctypes.c_int64]
        lib.oracle_synth_label_uniforms.restype = None
        lib.oracle_synth_label_uniforms.argtypes = [ctypes.c_uint64, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p]
        _lib = lib
    return _lib


def synth_features(seed, row0, nrows, p, kind=orc.SYNTH_UNIFORM):
    lib = load()
    if lib is None:
        return orc.synth_features(seed, row0, nrows, p, kind)
    X = np.empty((nrows, p), dtype=np.float64)
    lib.oracle_synth_features(int(seed), int(row0), int(nrows), int(p), int(kind), X.ctypes.data, p)
    return X


def synth_label_uniforms(seed, row0, nrows):
    lib = load()
    if lib is None:
        return orc.synth_label_uniforms(seed, row0, nrows)
    u = np.empty((nrows,), dtype=np.float64)
    lib.oracle_synth_label_uniforms(int(seed), int(row0), int(nrows), u.ctypes.data)
    return u


def synth_logistic(seed, row0, nrows, p, kind=orc.SYNTH_UNIFORM):
    """orc.synth_logistic on the fast generator: (X, y)."""
    X = synth_features(seed, row0, nrows, p, kind)
    prob = 1.0 / (1.0 + np.exp(-(X @ orc.true_beta(p))))
    y = (synth_label_uniforms(seed, row0, nrows) < prob).astype(np.float64)
    return X, y
