"""Tensor-level front end of the HIP engine.

PyTorch-ROCm is used for device memory and streams only: every function here hands raw
device pointers to libdlsa_hip.so through the C ABI of include/dlsa_hip.h and returns torch
tensors that own the results.  There is no CPU path -- a CPU tensor raises.
"""
import contextlib
import ctypes
import dataclasses
import threading
from typing import Optional

import numpy as np

import torch

from . import _lib
from ._lib import check

SYNTH_UNIFORM = 0
SYNTH_GAUSSIAN = 1

PART_STATUS = {0: "OK", 1: "NOT_CONVERGED", 2: "NOT_SPD", 3: "NAN", 4: "EMPTY"}

_ws_cache = {}


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _require_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("dlsa_amd runs on the GPU only: got a %s tensor (no CPU fallback)" % t.device)


def _f64(t, name, dtype=torch.float64):
    """The C ABI reads raw pointers: refuse anything that is not what the entry point expects -- the wrong dtype
    would be reinterpreted (an fp32 X read as n*ldx doubles runs past the buffer; integer labels read as ~0.0), a
    strided vector read as if it were contiguous.  2-D: row-major with unit column stride (any row pitch);
    1-D: contiguous.  None passes (nullable arguments)."""
    if t is None:
        return None
    if not torch.is_tensor(t):
        raise TypeError("%s must be a torch tensor on the GPU" % name)
    if t.dtype != dtype:
        raise TypeError("%s must be %s, got %s (no implicit casts at the C ABI: convert with .to(%s))"
                        % (name, dtype, t.dtype, dtype))
    if t.dim() == 2:
        if t.shape[1] > 1 and t.stride(1) != 1:
            raise ValueError("%s must be row-major (stride(1) == 1)" % name)
        if t.shape[0] > 1 and t.stride(0) < t.shape[1]:
            raise ValueError("%s: row pitch %d smaller than its %d columns" % (name, t.stride(0), t.shape[1]))
    elif not t.is_contiguous():
        raise ValueError("%s must be contiguous" % name)
    return t


_WS_MAX_STREAMS = 4


def _workspace(nbytes, device):
    """Scratch buffer per (device, current stream), 256-byte aligned by the torch allocator.  The C ABI is re-entrant across
    streams as long as concurrent calls bring their own workspace: work enqueued on two streams must not share scratch, and
    work on ONE stream is ordered, so a buffer per stream is exactly enough.  The cache is an LRU of _WS_MAX_STREAMS streams
    (a Gram scratch is ~0.5 GB at p = 500: a stream-per-request caller must not pin one per short-lived stream forever);
    an evicted buffer goes back to torch's stream-aware allocator, which keeps it alive until the work queued on it is done.
    release_workspace() drops everything."""
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device(),
           torch.cuda.current_stream(device).cuda_stream)
    buf = _ws_cache.pop(key, None)
    if buf is None or buf.numel() < nbytes:
        buf = None
        buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
    _ws_cache[key] = buf                     # (re)inserted last: dict order is the LRU order
    while len(_ws_cache) > _WS_MAX_STREAMS:
        old = next(iter(_ws_cache))
        victim = _ws_cache.pop(old)
        victim.record_stream(torch.cuda.current_stream(device))      # conservative: not reused before this stream's queued work
        del victim
    return buf


def release_workspace():
    _ws_cache.clear()


release_workspaces = release_workspace


def _rowmajor(X):
    # (a single column is row-major whatever stride(1) says: torch and numpy keep the parent's pitch on a size-1 dimension)
    if X.dim() != 2 or (X.shape[1] > 1 and X.stride(1) != 1):
        raise ValueError("X must be a 2-D row-major tensor (stride(1) == 1)")
    return max(X.stride(0), X.shape[1]) if X.shape[0] > 1 else max(X.stride(0), X.shape[1])


def rows_to_device(a, device="cuda"):
    """Host matrix [n, p] -> row-major device tensor.  pandas keeps the columns of one dtype as ONE [p, n] block, so
    `frame.to_numpy()` is a column-major VIEW of it; making that row-major on the host is a strided single-threaded copy
    (76 ms for 1e6 x 100 doubles), so a column-major array goes over the link as it lies and is transposed in HBM."""
    if isinstance(a, np.ndarray) and a.ndim == 2 and a.size > 0 and not a.flags.c_contiguous and a.flags.f_contiguous:
        t = torch.from_numpy(a.T).to(device).t()
    else:
        t = torch.from_numpy(np.ascontiguousarray(a)).to(device)
    if t.dim() == 2 and (t.stride(1) != 1 or t.stride(0) != t.shape[1]) and t.numel() > 0:
        # (a size-1 dimension counts as contiguous with ANY stride, for numpy and torch alike: an [n, 1] column view keeps its parent's pitch)
        out = torch.empty(t.shape, dtype=t.dtype, device=t.device)
        out.copy_(t)
        return out
    return t


def empty_rows(n, p, dtype=torch.float64, device="cuda"):
    """[n, p] row-major matrix whose row pitch is a whole number of 16-byte units (an odd fp64 p gets one pad
    element per row, zeroed): the Gram kernels stage such rows with vector loads / the LDS-DMA for any p, whereas
    rows that start at odd multiples of 8 bytes take the scalar staging path (p=501: 28.8 vs 20.6 ms per 5e6 rows).
    Exception (round 5): an odd fp64 width of the fused Newton pass's class (49 .. 120) stays PACKED -- that kernel streams packed
    odd rows (the piece of a row's last column carries the next row's first element: data, where a pad element would be bytes the
    library cannot vouch for), and a fit at these widths takes every Hessian from it."""
    unit = 16 // torch.empty((), dtype=dtype).element_size()
    ld = (p + unit - 1) // unit * unit
    if ld == p or (dtype == torch.float64 and (p & 1) and 49 <= p <= 120):
        return torch.empty((n, p), dtype=dtype, device=device)
    buf = torch.empty((n, ld), dtype=dtype, device=device)
    buf[:, p:] = 0
    return buf[:, :p]


def with_ones_column(X):
    """[1 | X] (the intercept column of dlsa/models.py:121-122) in an aligned row pitch."""
    out = empty_rows(X.shape[0], X.shape[1] + 1, X.dtype, X.device)
    out[:, 0] = 1
    out[:, 1:] = X
    return out


def row_major(X):
    """X itself when it is row-major with unit column stride (any row pitch), else a contiguous copy."""
    return X if (X.dim() == 2 and X.stride(1) == 1 and X.stride(0) >= X.shape[1]) else X.contiguous()


def synth(seed, row0, n, p, kind=SYNTH_UNIFORM, ones_col=False, labels=True, dtype=torch.float64,
          device="cuda", beta_true=None, out=None):
    """Seeded synthetic logistic rows (replaces simulate_logistic, dlsa/models.py:6-40).
    Returns (X [n, p + ones_col], y [n] or None)."""
    lib = _lib.load()
    cols = p + (1 if ones_col else 0)
    X = out if out is not None else empty_rows(n, cols, dtype, device)
    _require_gpu(X)
    y = torch.empty((n,), dtype=dtype, device=X.device) if labels else None
    fn = lib.dlsa_synth_f64 if dtype == torch.float64 else lib.dlsa_synth_f32
    # one launch handles < 2^31 workgroups: chunk the rows
    chunk = max(1, (1 << 28) // max(1, (p + 1) // 2))
    ld = _rowmajor(X)
    for r in range(0, n, chunk):
        m = min(chunk, n - r)
        check(fn(seed, row0 + r, m, p, kind, 1 if ones_col else 0,
                 _ptr(X[r:]), ld, _ptr(y[r:]) if labels else ctypes.c_void_p(0),
                 _ptr(beta_true), _stream()))
    return X, y


def synth_response(seed, row0, X, sigma=1.0, ones_col=False, beta_true=None, out=None):
    """Linear-model response y = X beta* + sigma N(0,1) for rows `synth` wrote (config 5; dlsa_synth_response_*): the noise of
    row i is a function of (seed, row0 + i) only.  X [n, p (+1 with ones_col)] fp64 / fp32 on the GPU; returns y [n]."""
    lib = _lib.load()
    _require_gpu(X, beta_true, out)
    if X.dtype not in (torch.float64, torch.float32):
        raise TypeError("synth_response: X must be float64 or float32")
    _f64(X, "X", X.dtype); _f64(beta_true, "beta_true", X.dtype); _f64(out, "out", X.dtype)
    n, cols = X.shape
    p = cols - (1 if ones_col else 0)
    y = out if out is not None else torch.empty((n,), dtype=X.dtype, device=X.device)
    if y.numel() != n or (beta_true is not None and beta_true.numel() != cols):
        raise ValueError("synth_response: y must have n and beta_true p (+1) elements")
    fn = lib.dlsa_synth_response_f64 if X.dtype == torch.float64 else lib.dlsa_synth_response_f32
    chunk = 1 << 30
    ld = _rowmajor(X)
    for r in range(0, n, chunk):
        m = min(chunk, n - r)
        check(fn(seed, row0 + r, m, p, 1 if ones_col else 0, _ptr(X[r:]), ld, _ptr(beta_true), float(sigma), _ptr(y[r:]), _stream()))
    return y


def synth_linear32(seed, row0, n, p, sigma=1.0, ones_col=False, beta_true=None, out=None, out_y=None, response=True):
    """fp32-native linear rows, features AND response in one launch (dlsa_synth_linear_f32: x ~ N(0, 1/12) from four 24-bit
    uniforms per Philox call, y = x . beta* + sigma N(0,1), everything in fp32 with the device's transcendental instructions).
    The chunk generator of the streaming map step at config 5's size: 4x cheaper than synth(kind=GAUSSIAN, float32) +
    synth_response.  Returns (X [n, p (+1)], y [n] or None)."""
    lib = _lib.load()
    cols = p + (1 if ones_col else 0)
    X = out if out is not None else empty_rows(n, cols, torch.float32, "cuda")
    _require_gpu(X, beta_true, out_y)
    _f64(X, "X", torch.float32); _f64(beta_true, "beta_true", torch.float32); _f64(out_y, "out_y", torch.float32)
    if tuple(X.shape) != (n, cols) or (beta_true is not None and beta_true.numel() != cols):
        raise ValueError("synth_linear32: X must be [n, p (+1)] and beta_true have p (+1) elements")
    y = None
    if response:
        y = out_y if out_y is not None else torch.empty((n,), dtype=torch.float32, device=X.device)
        if y.numel() != n:
            raise ValueError("synth_linear32: y must have n elements")
    check(lib.dlsa_synth_linear_f32(seed, row0, n, p, 1 if ones_col else 0, _ptr(X), _rowmajor(X), _ptr(beta_true), float(sigma),
                                    _ptr(y), _stream()))
    return X, y


def design(num, codes, kind, src, level, shift, scale, dtype=torch.float64, out=None):
    """Dense design matrix from raw numeric columns + integer level codes (models.py:56-104,121-122).
    num [n,q] (dtype) or None, codes [n,f] int32 or None; kind/src/level int32 [p], shift/scale fp64 [p]
    (device).  Returns (X [n,p], seen int32 [p]: 1 where the column has a non-zero entry)."""
    lib = _lib.load()
    _require_gpu(num, codes, kind, src, level, shift, scale, out)
    p = kind.numel()
    n = num.shape[0] if num is not None else codes.shape[0]
    q = num.shape[1] if num is not None else 0
    f = codes.shape[1] if codes is not None else 0
    if num is not None and num.dtype != dtype:
        raise ValueError("num must have the output dtype")
    if codes is not None and codes.dtype != torch.int32:
        raise ValueError("codes must be int32")
    dev = kind.device
    X = out if out is not None else empty_rows(n, p, dtype, dev)
    seen = torch.empty((p,), dtype=torch.int32, device=dev)
    fn = lib.dlsa_design_f64 if dtype == torch.float64 else lib.dlsa_design_f32
    check(fn(_ptr(num), _rowmajor(num) if num is not None else 0, q,
             _ptr(codes), _rowmajor(codes) if codes is not None else 0, f, n,
             _ptr(kind), _ptr(src), _ptr(level), _ptr(shift), _ptr(scale), p, _ptr(X), _rowmajor(X), _ptr(seen),
             _stream()))
    return X, seen


def gram(X, w=None, out=None, accumulate=False):
    """H = X' diag(w) X (dlsa/models.py:130) on the MFMA Gram kernel.  X [n,p] fp64/fp32."""
    lib = _lib.load()
    _require_gpu(X, w, out)
    if X.dtype not in (torch.float64, torch.float32):
        raise TypeError("gram: X must be float64 or float32, got %s" % X.dtype)
    _f64(X, "X", X.dtype); _f64(out, "out", X.dtype)
    n, p = X.shape
    ldx = _rowmajor(X)
    if out is not None and (out.dim() != 2 or tuple(out.shape) != (p, p)):
        raise ValueError("gram: out must be p x p")
    H = out if out is not None else torch.empty((p, p), dtype=X.dtype, device=X.device)
    es = X.element_size()
    nb = lib.dlsa_gram_workspace_bytes(n, p, es)
    ws = _workspace(nb, X.device)
    fn = lib.dlsa_gram_f64 if X.dtype == torch.float64 else lib.dlsa_gram_f32
    if w is not None and (w.dtype != X.dtype or not w.is_contiguous() or w.numel() != n):
        raise ValueError("w must be a contiguous vector of n elements with X's dtype")
    check(fn(_ptr(X), ldx, _ptr(w), n, p, _ptr(H), _rowmajor(H), 1 if accumulate else 0,
             _ptr(ws), ws.numel(), _stream()))
    return H


def gram_acc64(X, w=None, out=None, accumulate=False):
    """fp32 rows, fp64 result (dlsa_gram_f32_acc64): the MFMA passes of gram() on fp32 rows with the slab partials summed in
    fp64 and stored / added (accumulate) into the fp64 matrix `out` [p, p] (any row pitch: a sub-block view works).  The
    streaming linear map step adds chunk after chunk this way."""
    lib = _lib.load()
    _require_gpu(X, w, out)
    _f64(X, "X", torch.float32); _f64(w, "w", torch.float32); _f64(out, "out")
    n, p = X.shape
    H = out if out is not None else torch.empty((p, p), dtype=torch.float64, device=X.device)
    if H.dim() != 2 or tuple(H.shape) != (p, p):
        raise ValueError("gram_acc64: out must be p x p")
    if w is not None and w.numel() != n:
        raise ValueError("gram_acc64: w must have n elements")
    ws = _workspace(lib.dlsa_gram_workspace_bytes(n, p, 4), X.device)
    check(lib.dlsa_gram_f32_acc64(_ptr(X), _rowmajor(X), _ptr(w), n, p, _ptr(H), _rowmajor(H), 1 if accumulate else 0,
                                  _ptr(ws), ws.numel(), _stream()))
    return H


def xtv_stats(X, v, g=None, colsum=None, stats=None, want_colsum=False, accumulate=False):
    """One read of X: g = X'v, colsum = X'1 (optional: the implicit intercept's border), stats = [v'v, sum v], all fp64
    whatever X's type (dlsa_xtv_stats_*); accumulate adds to the given outputs.  Returns (g, colsum or None, stats)."""
    lib = _lib.load()
    _require_gpu(X, v, g, colsum, stats)
    if X.dtype not in (torch.float64, torch.float32):
        raise TypeError("xtv_stats: X must be float64 or float32")
    _f64(X, "X", X.dtype); _f64(v, "v", X.dtype); _f64(g, "g"); _f64(colsum, "colsum"); _f64(stats, "stats")
    n, p = X.shape
    if v.numel() != n:
        raise ValueError("xtv_stats: v must have n = %d elements" % n)
    dev = X.device
    if accumulate and (g is None or stats is None or (want_colsum and colsum is None)):
        raise ValueError("xtv_stats: accumulate needs the outputs to add to")
    g = g if g is not None else torch.empty((p,), dtype=torch.float64, device=dev)
    stats = stats if stats is not None else torch.empty((2,), dtype=torch.float64, device=dev)
    if want_colsum and colsum is None:
        colsum = torch.empty((p,), dtype=torch.float64, device=dev)
    if g.numel() != p or stats.numel() != 2 or (colsum is not None and colsum.numel() != p):
        raise ValueError("xtv_stats: g / colsum must have p and stats 2 elements")
    es = X.element_size()
    ws = _workspace(lib.dlsa_xtv_stats_workspace_bytes(p, es), dev)
    fn = lib.dlsa_xtv_stats_f64 if es == 8 else lib.dlsa_xtv_stats_f32
    check(fn(_ptr(X), _rowmajor(X), _ptr(v), n, p, _ptr(g), _ptr(colsum if want_colsum else None), _ptr(stats),
             1 if accumulate else 0, _ptr(ws), ws.numel(), _stream()))
    return g, (colsum if want_colsum else None), stats


def gram_last_kernel(want_cycles=False):
    """(name, shader cycles) of the Gram kernel this thread's last Gram launch dispatched (dlsa_gram_last_kernel):
    which of the kernels of DESIGN.md section 4.1 ran, and -- want_cycles, which synchronises -- the s_memtime delta of
    wave 0 of workgroup 0 (0 for kernels without the probe).  cycles / kernel time = the sustained shader clock."""
    lib = _lib.load()
    buf = ctypes.create_string_buffer(160)
    cyc = ctypes.c_uint64(0)
    check(lib.dlsa_gram_last_kernel(buf, 160, ctypes.byref(cyc) if want_cycles else None))
    return buf.value.decode(), int(cyc.value)


def logit_pass(X, y, beta, want_w=True, want_g=True, want_loglik=True, fit_intercept=False):
    """One fused pass: w = mu(1-mu), g = X'(y-mu), loglik.  Returns (w, g, loglik) tensors.  fit_intercept: the ones
    column is implicit -- beta and g have p + 1 entries, intercept first."""
    lib = _lib.load()
    _require_gpu(X, y, beta)
    _f64(X, "X"); _f64(y, "y"); _f64(beta, "beta")
    n, p = X.shape
    pe = p + (1 if fit_intercept else 0)
    if y.numel() != n or beta.numel() != pe:
        raise ValueError("logit_pass: y must have n = %d and beta %d elements" % (n, pe))
    ldx = _rowmajor(X)
    w = torch.empty((n,), dtype=torch.float64, device=X.device) if want_w else None
    g = torch.empty((pe,), dtype=torch.float64, device=X.device) if want_g else None
    ll = torch.empty((1,), dtype=torch.float64, device=X.device) if want_loglik else None
    nb = lib.dlsa_logit_workspace_bytes(n, p)
    ws = _workspace(nb, X.device)
    fn = lib.dlsa_logit_pass_icpt_f64 if fit_intercept else lib.dlsa_logit_pass_f64
    check(fn(_ptr(X), ldx, _ptr(y), _ptr(beta), n, p, _ptr(w), _ptr(g), _ptr(ll), _ptr(ws), ws.numel(), _stream()))
    return w, g, ll


def irls_pass(X, y, beta, want_w=False):
    """One Newton pass in ONE call (dlsa_irls_pass_f64): w = mu(1-mu) at beta, g = X'(y-mu), loglik and H = X' diag(w) X.
    Narrow designs (49 <= p <= 120, even, aligned rows, n >= 8192) run the FUSED kernel -- one read of X; every other shape
    the logit pass + the Gram pass behind the same entry (gram_last_kernel() names what ran).  Returns (H, g, loglik, w or None)."""
    lib = _lib.load()
    _require_gpu(X, y, beta)
    _f64(X, "X"); _f64(y, "y"); _f64(beta, "beta")
    n, p = X.shape
    if y.numel() != n or beta.numel() != p:
        raise ValueError("irls_pass: y must have n = %d and beta p = %d elements" % (n, p))
    dev = X.device
    H = torch.empty((p, p), dtype=torch.float64, device=dev)
    g = torch.empty((p,), dtype=torch.float64, device=dev)
    ll = torch.empty((1,), dtype=torch.float64, device=dev)
    w = torch.empty((n,), dtype=torch.float64, device=dev) if want_w else None
    ws = _workspace(lib.dlsa_irls_pass_workspace_bytes(n, p), dev)
    check(lib.dlsa_irls_pass_f64(_ptr(X), _rowmajor(X), _ptr(y), _ptr(beta), n, p, _ptr(H), p, _ptr(g), _ptr(ll), _ptr(w),
                                 _ptr(ws), ws.numel(), _stream()))
    return H, g, ll, w


def newton_wide_pass(X, y, beta, fit_intercept=False, want_w=False):
    """The wide Newton pass (dlsa_newton_wide_pass_f64): g, loglik (and w) of the logit pass at beta plus, from the SAME read of
    the rows, the reduced-precision Hessian that preconditions the fit's Newton steps (bf16 products, fp32 accumulation; never a
    result: exported for diagnostics and tests).  Returns (H_approx, g, loglik, w or None); raises when the shape is not served
    (121 <= p + intercept <= 512, >= 32768 rows; any pitch or alignment -- unaligned / odd-pitch rows take the scalar-load form)."""
    lib = _lib.load()
    _require_gpu(X, y, beta)
    _f64(X, "X"); _f64(y, "y"); _f64(beta, "beta")
    n, p = X.shape
    pe = p + (1 if fit_intercept else 0)
    if y.numel() != n or beta.numel() != pe:
        raise ValueError("newton_wide_pass: y must have n = %d and beta %d elements" % (n, pe))
    dev = X.device
    H = torch.empty((pe, pe), dtype=torch.float64, device=dev)
    g = torch.empty((pe,), dtype=torch.float64, device=dev)
    ll = torch.empty((1,), dtype=torch.float64, device=dev)
    w = torch.empty((n,), dtype=torch.float64, device=dev) if want_w else None
    ws = _workspace(lib.dlsa_newton_wide_workspace_bytes(n, p, 1 if fit_intercept else 0), dev)
    check(lib.dlsa_newton_wide_pass_f64(_ptr(X), _rowmajor(X), _ptr(y), _ptr(beta), n, p, 1 if fit_intercept else 0, _ptr(w), _ptr(g),
                                        _ptr(ll), _ptr(H), pe, _ptr(ws), ws.numel(), _stream()))
    return H, g, ll, w


def gram_icpt(X, w=None, out=None):
    """H = [1 | X]' diag(w) [1 | X], (p + 1) x (p + 1), without materialising the ones column (models.py:121-130)."""
    lib = _lib.load()
    _require_gpu(X, w, out)
    _f64(X, "X"); _f64(w, "w"); _f64(out, "out")
    n, p = X.shape
    if w is not None and w.numel() != n:
        raise ValueError("gram_icpt: w must have n elements")
    H = out if out is not None else torch.empty((p + 1, p + 1), dtype=torch.float64, device=X.device)
    if tuple(H.shape) != (p + 1, p + 1):
        raise ValueError("gram_icpt: out must be (p + 1) x (p + 1)")
    ws = _workspace(lib.dlsa_gram_icpt_workspace_bytes(n, p), X.device)
    check(lib.dlsa_gram_icpt_f64(_ptr(X), _rowmajor(X), _ptr(w), n, p, _ptr(H), _rowmajor(H), _ptr(ws), ws.numel(), _stream()))
    return H


def loglik(X, y, par, fit_intercept=False):
    """Log-likelihood of each column of par [p, c] (dlsa/models.py:217-222).  fit_intercept: par has p + 1 rows, the
    first one the intercepts; the ones column is implicit."""
    lib = _lib.load()
    _require_gpu(X, y, par)
    _f64(X, "X"); _f64(y, "y")
    n, p = X.shape
    par = par.contiguous()
    _f64(par, "par")
    if par.dim() != 2 or par.shape[0] != p + (1 if fit_intercept else 0) or y.numel() != n:
        raise ValueError("loglik: par must be [p (+1), c] and y [n]")
    c = par.shape[1]
    out = torch.empty((c,), dtype=torch.float64, device=X.device)
    nb = lib.dlsa_logit_workspace_bytes(n, p)
    ws = _workspace(nb, X.device)
    fn = lib.dlsa_loglik_icpt_f64 if fit_intercept else lib.dlsa_loglik_f64
    check(fn(_ptr(X), _rowmajor(X), _ptr(y), n, p, _ptr(par), par.stride(0), c, _ptr(out), _ptr(ws), ws.numel(), _stream()))
    return out


def xtv(X, v):
    """g = X'v and v'v in one read of X (linear-model map step).  Returns (g [p], vv [1]) in X's dtype."""
    lib = _lib.load()
    _require_gpu(X, v)
    if X.dtype not in (torch.float64, torch.float32):
        raise TypeError("xtv: X must be float64 or float32")
    _f64(X, "X", X.dtype)
    v = _f64(v.contiguous(), "v", X.dtype)
    n, p = X.shape
    if v.numel() != n:
        raise ValueError("xtv: v must have n = %d elements" % n)
    g = torch.empty((p,), dtype=X.dtype, device=X.device)
    vv = torch.empty((1,), dtype=X.dtype, device=X.device)
    nb = lib.dlsa_logit_workspace_bytes(n, p) * (2 if X.dtype == torch.float32 else 1)
    ws = _workspace(nb, X.device)
    if X.dtype == torch.float32:
        check(lib.dlsa_xtv_f32(_ptr(X), _rowmajor(X), _ptr(v.contiguous()), n, p, _ptr(g), _ptr(vv),
                               _ptr(ws), ws.numel(), _stream()))
        return g, vv
    check(lib.dlsa_xtv_f64(_ptr(X), _rowmajor(X), _ptr(v.contiguous()), n, p, _ptr(g), _ptr(vv),
                           _ptr(ws), ws.numel(), _stream()))
    return g, vv


@dataclasses.dataclass
class IrlsOptions:
    """Policy of the IRLS driver for the fits made inside `with engine.irls_options(opts):` (or passed as `options=` to
    fit_logistic_partitions / fit_logistic_design / logistic_model): include/dlsa_hip.h dlsa_irls_options, field for field.  None =
    automatic (the measured default for the shapes); every setting returns the same MLE and Hessian to the parity tolerance."""
    chains: Optional[int] = None            # partition chains of one call (host threads + streams), 1..8
    seeded: Optional[bool] = None           # partition 0 alone first, every chain seeded with its state
    subsample_div: Optional[int] = None     # cold start on rows / d of a partition; 0 or 1: none
    factor_div: Optional[int] = None
    warm: Optional[bool] = None             # partition k + 1 starts from partition k's MLE
    inherit: Optional[bool] = None
    pool: Optional[bool] = None
    secant: Optional[bool] = None
    inverse: Optional[bool] = None
    predict: Optional[bool] = None          # predicted convergence
    fused: Optional[bool] = None            # fused Newton pass (one read of the rows per fresh Hessian)
    fuse_last: Optional[bool] = None
    small: Optional[bool] = None            # one-launch kernel for many small partitions
    batched: Optional[bool] = None          # lock-step fit of all partitions together (narrow designs)
    qn_threads: Optional[int] = None
    trace: Optional[bool] = None
    lean: Optional[bool] = None             # fits at fused widths write no weight vector
    small_cluster: Optional[int] = None     # workgroups per partition of the one-launch kernel, 1..16
    own_hessian: Optional[bool] = None      # wide designs: Newton steps preconditioned by the partition's own reduced-precision Hessian
    pooled_start: Optional[bool] = None     # lock step: full-row iterations start from one fit on the leading rows of all partitions
    grad_passes: Optional[int] = None       # lock step: at most this many gradient-only passes before the Newton passes; 0: none
    freeze_at: Optional[float] = None       # 0: never freeze the factor

    def as_c(self):
        lib = _lib.load()
        c = _lib.IrlsOptionsC()
        lib.dlsa_irls_options_init(ctypes.byref(c))
        for f in dataclasses.fields(self):
            v = getattr(self, f.name)
            if v is not None:
                setattr(c, f.name, float(v) if f.name == "freeze_at" else int(v))
        return c


_options_stack = threading.local()          # the calling thread's active IrlsOptions, innermost last


@contextlib.contextmanager
def irls_options(options=None, **fields):
    """The calling thread's IRLS driver options for the fits (and workspace sizes) inside the block: an IrlsOptions, or its fields as
    keyword arguments (`with engine.irls_options(chains=1, fused=False): ...`).  Blocks nest: the fields set here are merged over the
    enclosing block's, and the enclosing options are restored on exit (automatic only when the outermost block ends)."""
    if options is None:
        options = IrlsOptions(**fields) if fields else None
    elif fields:
        options = dataclasses.replace(options, **fields)
    if options is None:
        yield
        return
    lib = _lib.load()
    stack = getattr(_options_stack, "items", None)
    if stack is None:
        stack = _options_stack.items = []
    if stack:           # merge over the enclosing block's options
        outer = stack[-1]
        options = dataclasses.replace(outer, **{f.name: getattr(options, f.name) for f in dataclasses.fields(options)
                                               if getattr(options, f.name) is not None})
    c = options.as_c()
    check(lib.dlsa_irls_set_options(ctypes.byref(c)))
    stack.append(options)
    try:
        yield
    finally:
        stack.pop()
        if stack:
            c = stack[-1].as_c()
            lib.dlsa_irls_set_options(ctypes.byref(c))
        else:
            lib.dlsa_irls_set_options(None)


@dataclasses.dataclass
class KernelOptions:
    """Which build of a kernel runs (never what it returns) for the calls inside `with engine.kernel_options(...)`:
    include/dlsa_hip.h dlsa_kernel_options, field for field.  None = automatic.  The library reads no environment variable for these."""
    lars_q: Optional[int] = None            # LARS form: 0 lars.hip everywhere; 1 lars_q.hip up to 1020 variables, lars_c.hip beyond; 2 lars_c.hip from 64; None: the measured hand-over (448)
    lars_q_wgs: Optional[int] = None        # workgroups that share its fused pass, 1..8
    lars_q_threads: Optional[int] = None    # 256 | 512 | 1024
    lars_q_lds: Optional[bool] = None
    lars_wgs: Optional[int] = None          # workgroups of lars.hip's grid kernel (1..32) / of lars_c.hip's column split (2..64)
    lars_threads: Optional[int] = None      # 512 | 1024
    logit_ring: Optional[bool] = None       # narrow designs' logit pass through the LDS-DMA ring
    chol_small: Optional[bool] = None       # one-launch SPD inverse for p <= 112
    gram_wide_f32: Optional[bool] = None    # the fp32 wide Gram kernel (p >= 768)
    onehot_ordered: Optional[int] = None    # 0 unordered, 1 ordered floating point
    gram_variant: Optional[int] = None      # valid-result A/B bits of the fp64 Gram dispatch (2 | 4 | 8 | 32 | 64 | 256)
    cooperative: Optional[bool] = None      # multi-workgroup kernels launched cooperatively

    def as_c(self):
        lib = _lib.load()
        c = _lib.KernelOptionsC()
        lib.dlsa_kernel_options_init(ctypes.byref(c))
        for f in dataclasses.fields(self):
            v = getattr(self, f.name)
            if v is not None:
                setattr(c, f.name, int(v))
        return c


_kernel_options_stack = threading.local()

# the environment variables earlier rounds' A/B scripts used for these switches, honoured by the HOST layer only (bench/ scripts call
# kernel_options_from_env(); the library itself never reads them)
_KERNEL_ENV = {"DLSA_LARS_Q": "lars_q", "DLSA_LARS_Q_WGS": "lars_q_wgs", "DLSA_LARS_Q_THREADS": "lars_q_threads", "DLSA_LARS_Q_LDS": "lars_q_lds",
               "DLSA_LARS_WGS": "lars_wgs", "DLSA_LARS_THREADS": "lars_threads", "DLSA_LOGIT_RING": "logit_ring", "DLSA_CHOL_SMALL": "chol_small",
               "DLSA_GRAM_WIDE_F32": "gram_wide_f32", "DLSA_OH_ORDERED": "onehot_ordered", "DLSA_GRAM_DBG": "gram_variant", "DLSA_COOPERATIVE": "cooperative"}


def kernel_options_from_env(environ=None):
    """A KernelOptions from the DLSA_* variables of the A/B scripts (None when none is set)."""
    import os
    environ = os.environ if environ is None else environ
    fields = {f: int(environ[k]) for k, f in _KERNEL_ENV.items() if environ.get(k, "") != ""}
    return KernelOptions(**fields) if fields else None


@contextlib.contextmanager
def kernel_options(options=None, **fields):
    """The calling thread's kernel switches for the calls inside the block (a KernelOptions, or its fields as keyword arguments);
    blocks nest like irls_options."""
    if options is None:
        options = KernelOptions(**fields) if fields else None
    elif fields:
        options = dataclasses.replace(options, **fields)
    if options is None:
        yield
        return
    lib = _lib.load()
    stack = getattr(_kernel_options_stack, "items", None)
    if stack is None:
        stack = _kernel_options_stack.items = []
    if stack:
        options = dataclasses.replace(stack[-1], **{f.name: getattr(options, f.name) for f in dataclasses.fields(options)
                                                    if getattr(options, f.name) is not None})
    c = options.as_c()
    check(lib.dlsa_kernel_set_options(ctypes.byref(c)))
    stack.append(options)
    try:
        yield
    finally:
        stack.pop()
        if stack:
            c = stack[-1].as_c()
            lib.dlsa_kernel_set_options(ctypes.byref(c))
        else:
            lib.dlsa_kernel_set_options(None)


IRLS_PATH_CHAINS, IRLS_PATH_SMALL, IRLS_PATH_BATCHED = 0, 1, 2


def irls_last_fit_path():
    """Which driver this thread's last irls_fit / irls_fit_ex took (dlsa_irls_last_fit_path): IRLS_PATH_CHAINS (host-driven partition
    chains), IRLS_PATH_SMALL (one launch, a workgroup per partition) or IRLS_PATH_BATCHED (lock step: all partitions together)."""
    return int(_lib.load().dlsa_irls_last_fit_path())


def irls_fit_ex(X, y, part_first, part_rows, row_step=1, fit_intercept=False, tol=1e-13, max_iter=100):
    """irls_fit without copies of the shard: partition k = rows part_first[k] + j * row_step, j < part_rows[k] (a strided
    view: partition_id = i % K is part_first = 0..K-1, row_step = K), and an IMPLICIT intercept column (fit_intercept:
    results have p + 1 columns, intercept first).  Same result dict as irls_fit."""
    lib = _lib.load()
    _require_gpu(X, y)
    _f64(X, "X"); _f64(y, "y")
    n, p = X.shape
    if y.numel() != n:
        raise ValueError("irls_fit_ex: y must have n = %d elements" % n)
    first = [int(v) for v in part_first]
    rows = [int(v) for v in part_rows]
    K, step = len(first), int(row_step)
    if len(rows) != K or K == 0 or step < 1:
        raise ValueError("irls_fit_ex: part_first / part_rows must have K >= 1 entries each, row_step >= 1")
    for f, r in zip(first, rows):
        if f < 0 or r < 0 or (r > 0 and f + (r - 1) * step >= n):
            raise ValueError("irls_fit_ex: partition outside the %d rows of X" % n)
    pe = p + (1 if fit_intercept else 0)
    dev = X.device
    coef = torch.empty((K, pe), dtype=torch.float64, device=dev)
    smc = torch.empty((K, pe), dtype=torch.float64, device=dev)
    sig = torch.empty((K, pe, pe), dtype=torch.float64, device=dev)
    ws = _workspace(lib.dlsa_irls_ex_workspace_bytes(max(rows), p, 1 if fit_intercept else 0, step), dev)
    c_first, c_rows = (ctypes.c_int64 * K)(*first), (ctypes.c_int64 * K)(*rows)
    n_iter, status, ll = (ctypes.c_int * K)(), (ctypes.c_int * K)(), (ctypes.c_double * K)()
    rc = lib.dlsa_irls_fit_ex_f64(_ptr(X), _rowmajor(X), _ptr(y), c_first, c_rows, step, K, p, 1 if fit_intercept else 0,
                                  tol, max_iter, _ptr(coef), _ptr(sig), _ptr(smc), n_iter, status, ll, _ptr(ws), ws.numel(),
                                  _stream())
    if rc not in (0, 4, 5, 6):     # per-partition soft failures are reported through `status`
        check(rc)
    return {"coef": coef, "Sig_invMcoef": smc, "Sig_inv": sig, "n_iter": list(n_iter), "status": list(status),
            "loglik": list(ll), "rc": rc}


def irls_fit(X, y, part_offsets, tol=1e-13, max_iter=100):
    """Per-partition exact-MLE fit + local quadratic approximation (dlsa/models.py:110-131).
    part_offsets: K+1 ints; partition k = rows [off[k], off[k+1]).  Returns a dict with
    coef [K,p], Sig_invMcoef [K,p], Sig_inv [K,p,p] (device) and n_iter/status/loglik (host)."""
    lib = _lib.load()
    _require_gpu(X, y)
    _f64(X, "X"); _f64(y, "y")
    n, p = X.shape
    if y.numel() != n:
        raise ValueError("irls_fit: y must have n = %d elements" % n)
    offs = [int(v) for v in part_offsets]
    K = len(offs) - 1
    if offs[0] < 0 or offs[-1] > n:
        raise ValueError("part_offsets out of range")
    dev = X.device
    coef = torch.empty((K, p), dtype=torch.float64, device=dev)
    smc = torch.empty((K, p), dtype=torch.float64, device=dev)
    sig = torch.empty((K, p, p), dtype=torch.float64, device=dev)
    max_rows = max(offs[k + 1] - offs[k] for k in range(K))
    nb = lib.dlsa_irls_workspace_bytes(max_rows, p)
    ws = _workspace(nb, dev)
    c_offs = (ctypes.c_int64 * (K + 1))(*offs)
    n_iter = (ctypes.c_int * K)()
    status = (ctypes.c_int * K)()
    ll = (ctypes.c_double * K)()
    rc = lib.dlsa_irls_fit_f64(_ptr(X), _rowmajor(X), _ptr(y), c_offs, K, p, tol, max_iter,
                               _ptr(coef), _ptr(sig), _ptr(smc), n_iter, status, ll,
                               _ptr(ws), ws.numel(), _stream())
    if rc not in (0, 4, 5, 6):     # per-partition soft failures are reported through `status`
        check(rc)
    return {"coef": coef, "Sig_invMcoef": smc, "Sig_inv": sig, "n_iter": list(n_iter),
            "status": list(status), "loglik": list(ll), "rc": rc}


def sum_blocks(coef, smc, sig, mask=None):
    """[sum Sig_inv | sum Sig_invMcoef | sum coef]: the rank's all-reduce message (dlsa.py:30-34)."""
    lib = _lib.load()
    _require_gpu(coef, smc, sig)
    _f64(coef, "coef"); _f64(smc, "Sig_invMcoef")
    if sig.dtype != torch.float64:
        raise TypeError("Sig_inv must be float64, got %s" % sig.dtype)
    K, p = coef.shape
    if tuple(smc.shape) != (K, p) or tuple(sig.shape) != (K, p, p):
        raise ValueError("sum_blocks: expected coef / Sig_invMcoef [K, p] and Sig_inv [K, p, p]")
    out = torch.empty((p * p + 2 * p,), dtype=torch.float64, device=coef.device)
    cmask = (ctypes.c_int * K)(*[int(v) for v in mask]) if mask is not None else None
    check(lib.dlsa_sum_blocks_f64(_ptr(coef.contiguous()), _ptr(sig.contiguous()), _ptr(smc.contiguous()),
                                  K, p, cmask, _ptr(out), _stream()))
    return out


def spd_solve(S, v):
    """theta = S^{-1} v by device Cholesky (the WLS combine, dlsa/dlsa.py:48-49)."""
    lib = _lib.load()
    _require_gpu(S, v)
    _f64(S, "S")
    v = _f64(v.contiguous(), "v")
    p = S.shape[0]
    if S.dim() != 2 or S.shape[1] != p or v.numel() != p:
        raise ValueError("spd_solve: S must be p x p and v of length p")
    theta = torch.empty((p,), dtype=torch.float64, device=S.device)
    nb = lib.dlsa_solve_workspace_bytes(p)
    ws = _workspace(nb, S.device)
    check(lib.dlsa_spd_solve_f64(_ptr(S), _rowmajor(S), _ptr(v), p, _ptr(theta),
                                 _ptr(ws), ws.numel(), _stream()))
    return theta


def _solve_args(S, v, who):
    _require_gpu(S, v)
    _f64(S, "S")
    v = _f64(v.contiguous(), "v")
    p = S.shape[0]
    if S.dim() != 2 or S.shape[1] != p or v.numel() != p:
        raise ValueError("%s: S must be p x p and v of length p" % who)
    return v, p


def wls_solve(S, v):
    """theta = lstsq(S, v, rcond=None)[0] (dlsa/dlsa.py:48-49): Cholesky solve when S is SPD, the minimum-norm
    least-squares solution (device Jacobi eigendecomposition) when it is singular.  Returns (theta, rank)."""
    lib = _lib.load()
    v, p = _solve_args(S, v, "wls_solve")
    theta = torch.empty((p,), dtype=torch.float64, device=S.device)
    ws = _workspace(lib.dlsa_wls_solve_workspace_bytes(p), S.device)
    rank = ctypes.c_int(0)
    check(lib.dlsa_wls_solve_f64(_ptr(S), _rowmajor(S), _ptr(v), p, _ptr(theta), ctypes.byref(rank),
                                 _ptr(ws), ws.numel(), _stream()))
    return theta, rank.value


def sym_pinv_solve(S, v, rcond=None):
    """theta = pinv(S) v for a symmetric S through its eigendecomposition (parallel Jacobi on the device), with
    lstsq's singular-value cut (rcond=None -> eps * p).  Returns (theta, rank, eigenvalues as a host list)."""
    lib = _lib.load()
    v, p = _solve_args(S, v, "sym_pinv_solve")
    theta = torch.empty((p,), dtype=torch.float64, device=S.device)
    ws = _workspace(lib.dlsa_sym_pinv_workspace_bytes(p), S.device)
    rank = ctypes.c_int(0)
    eig = (ctypes.c_double * p)()
    check(lib.dlsa_sym_pinv_solve_f64(_ptr(S), _rowmajor(S), _ptr(v), p, -1.0 if rcond is None else float(rcond),
                                      _ptr(theta), ctypes.byref(rank), eig, _ptr(ws), ws.numel(), _stream()))
    return theta, rank.value, list(eig)


def lars_path(Sigma0, b0, intercept, n, type="lar", eps=2.220446049250313e-16, max_steps=None):
    """LARS / lasso path of the LSA objective on the device (dlsa/lsa.py:90-212).
    Returns dict of device tensors AIC, BIC [steps+1], beta [steps+1, m], beta0 [steps+1]."""
    lib = _lib.load()
    _require_gpu(Sigma0, b0)
    _f64(Sigma0, "Sigma0")
    b0 = _f64(b0.contiguous(), "b0")
    p = Sigma0.shape[0]
    if Sigma0.dim() != 2 or Sigma0.shape[1] != p or b0.numel() != p:
        raise ValueError("lars_path: Sigma0 must be p x p and b0 of length p")
    m = p - (1 if intercept else 0)
    # lsa.py:93-94: max_steps defaults to 8 m; the C ABI reads <= 0 as that default, so the buffers are sized for it too
    ms = 8 * m if (max_steps is None or int(max_steps) <= 0) else int(max_steps)
    dev = Sigma0.device
    beta = torch.zeros((ms + 1, m), dtype=torch.float64, device=dev)
    beta0 = torch.zeros((ms + 1,), dtype=torch.float64, device=dev)
    aic = torch.zeros((ms + 1,), dtype=torch.float64, device=dev)
    bic = torch.zeros((ms + 1,), dtype=torch.float64, device=dev)
    nb = lib.dlsa_lars_workspace_bytes(p)
    ws = _workspace(nb, dev)
    steps = ctypes.c_int(0)
    check(lib.dlsa_lars_lsa_f64(_ptr(Sigma0), _rowmajor(Sigma0), _ptr(b0), p, 1 if intercept else 0,
                                float(n), {"lar": 0, "lasso": 1}[type], float(eps), ms,
                                _ptr(beta), _ptr(beta0), _ptr(aic), _ptr(bic), ctypes.byref(steps),
                                _ptr(ws), ws.numel(), _stream()))
    k = steps.value
    return {"AIC": aic[: k + 1], "BIC": bic[: k + 1], "beta": beta[: k + 1], "beta0": beta0[: k + 1]}


class OnehotPlan:
    """Handle of a structured one-hot design (include/dlsa_hip.h, dlsa_onehot_plan_create).  Host descriptor arrays:
    dense_kind/src/col int32 [D], dense_shift/scale fp64 [D], nlevels int32 [f], level_col int32 [sum nlevels]."""

    def __init__(self, p, dense_kind, dense_src, dense_shift, dense_scale, dense_col, nlevels, level_col):
        import numpy as np
        lib = _lib.load()
        if not torch.cuda.is_available():
            raise RuntimeError("dlsa_amd runs on the GPU only (no CPU fallback)")
        a = lambda v, t: np.ascontiguousarray(np.asarray(v, dtype=t))
        dk, dsrc, dcol = a(dense_kind, np.int32), a(dense_src, np.int32), a(dense_col, np.int32)
        dsh, dsc = a(dense_shift, np.float64), a(dense_scale, np.float64)
        nl, lc = a(nlevels, np.int32), a(level_col, np.int32)
        P = lambda x: ctypes.c_void_p(x.ctypes.data) if x.size else ctypes.c_void_p(0)
        h = ctypes.c_void_p(0)
        check(lib.dlsa_onehot_plan_create(int(p), int(dk.size), P(dk), P(dsrc), P(dsh), P(dsc), P(dcol), int(nl.size),
                                          P(nl), P(lc), ctypes.byref(h)))
        self._h, self._lib = h, lib
        self.p, self.D, self.f = int(p), int(dk.size), int(nl.size)
        self.roles = lib.dlsa_onehot_plan_roles(h)

    def __del__(self):
        try:
            if self._h:
                self._lib.dlsa_onehot_plan_destroy(self._h)
                self._h = None
        except Exception:
            pass


def _oh_args(plan, num, codes):
    _require_gpu(num, codes)
    if num is not None and num.dtype != torch.float64:
        raise ValueError("num must be fp64")
    if codes is not None and codes.dtype != torch.int32:
        raise ValueError("codes must be int32")
    return (_ptr(num), _rowmajor(num) if num is not None else 0, _ptr(codes), _rowmajor(codes) if codes is not None else 0)


def onehot_logit_pass(plan, num, codes, y, beta, want_w=True, want_g=True, want_loglik=True):
    """logit_pass on the raw representation of a one-hot design (num [n,q] fp64, codes [n,f] int32)."""
    lib = _lib.load()
    _require_gpu(y, beta)
    _f64(y, "y"); _f64(beta, "beta")
    n = y.numel()
    dev = y.device
    w = torch.empty((n,), dtype=torch.float64, device=dev) if want_w else None
    g = torch.empty((plan.p,), dtype=torch.float64, device=dev) if want_g else None
    ll = torch.empty((1,), dtype=torch.float64, device=dev) if want_loglik else None
    ws = _workspace(lib.dlsa_onehot_workspace_bytes(plan._h, n), dev)
    pn, ldn, pc, ldc = _oh_args(plan, num, codes)
    check(lib.dlsa_onehot_logit_pass_f64(plan._h, pn, ldn, pc, ldc, _ptr(y), _ptr(beta), n, _ptr(w), _ptr(g), _ptr(ll),
                                         _ptr(ws), ws.numel(), _stream()))
    return w, g, ll


def onehot_gram(plan, num, codes, w, n=None):
    """X' diag(w) X of a one-hot design from its raw representation: the same p x p matrix as gram()."""
    lib = _lib.load()
    _require_gpu(w)
    _f64(w, "w")
    n = int(n if n is not None else (w.numel() if w is not None else (codes.shape[0] if codes is not None else num.shape[0])))
    dev = (codes if codes is not None else num).device
    H = torch.empty((plan.p, plan.p), dtype=torch.float64, device=dev)
    ws = _workspace(lib.dlsa_onehot_workspace_bytes(plan._h, n), dev)
    pn, ldn, pc, ldc = _oh_args(plan, num, codes)
    check(lib.dlsa_onehot_gram_f64(plan._h, pn, ldn, pc, ldc, _ptr(w), n, _ptr(H), H.stride(0), _ptr(ws), ws.numel(), _stream()))
    return H


def onehot_irls_fit(plan, num, codes, y, part_offsets, tol=1e-13, max_iter=100):
    """irls_fit on the raw representation of a one-hot design; same result dict."""
    lib = _lib.load()
    _require_gpu(y)
    _f64(y, "y")
    p = plan.p
    offs = [int(v) for v in part_offsets]
    K = len(offs) - 1
    if offs[0] < 0 or offs[-1] > y.numel():
        raise ValueError("part_offsets out of range")
    dev = y.device
    coef = torch.empty((K, p), dtype=torch.float64, device=dev)
    smc = torch.empty((K, p), dtype=torch.float64, device=dev)
    sig = torch.empty((K, p, p), dtype=torch.float64, device=dev)
    max_rows = max(offs[k + 1] - offs[k] for k in range(K))
    ws = _workspace(lib.dlsa_onehot_irls_workspace_bytes(plan._h, max_rows), dev)
    c_offs = (ctypes.c_int64 * (K + 1))(*offs)
    n_iter, status, ll = (ctypes.c_int * K)(), (ctypes.c_int * K)(), (ctypes.c_double * K)()
    pn, ldn, pc, ldc = _oh_args(plan, num, codes)
    rc = lib.dlsa_onehot_irls_fit_f64(plan._h, pn, ldn, pc, ldc, _ptr(y), c_offs, K, tol, max_iter, _ptr(coef), _ptr(sig),
                                      _ptr(smc), n_iter, status, ll, _ptr(ws), ws.numel(), _stream())
    if rc not in (0, 4, 5, 6):
        check(rc)
    return {"coef": coef, "Sig_invMcoef": smc, "Sig_inv": sig, "n_iter": list(n_iter), "status": list(status),
            "loglik": list(ll), "rc": rc}


def onehot_irls_fit_ex(plan, num, codes, y, part_first, part_rows, row_step=1, tol=1e-13, max_iter=100):
    """onehot_irls_fit for partitions given as (first row, rows, common row step): partition_id = i % K is part_first = 0..K-1,
    row_step = K -- strided views of num / codes, no gather (dlsa_onehot_irls_fit_ex_f64).  Same result dict."""
    lib = _lib.load()
    _require_gpu(y)
    _f64(y, "y")
    p = plan.p
    first, rows = [int(v) for v in part_first], [int(v) for v in part_rows]
    K, step, n = len(first), int(row_step), y.numel()
    if len(rows) != K or K == 0 or step < 1:
        raise ValueError("onehot_irls_fit_ex: part_first / part_rows must have K >= 1 entries each, row_step >= 1")
    for f, r in zip(first, rows):
        if f < 0 or r < 0 or (r > 0 and f + (r - 1) * step >= n):
            raise ValueError("onehot_irls_fit_ex: partition outside the %d rows" % n)
    dev = y.device
    coef = torch.empty((K, p), dtype=torch.float64, device=dev)
    smc = torch.empty((K, p), dtype=torch.float64, device=dev)
    sig = torch.empty((K, p, p), dtype=torch.float64, device=dev)
    ws = _workspace(lib.dlsa_onehot_irls_ex_workspace_bytes(plan._h, max(rows), step), dev)
    c_first, c_rows = (ctypes.c_int64 * K)(*first), (ctypes.c_int64 * K)(*rows)
    n_iter, status, ll = (ctypes.c_int * K)(), (ctypes.c_int * K)(), (ctypes.c_double * K)()
    pn, ldn, pc, ldc = _oh_args(plan, num, codes)
    rc = lib.dlsa_onehot_irls_fit_ex_f64(plan._h, pn, ldn, pc, ldc, _ptr(y), c_first, c_rows, step, K, tol, max_iter, _ptr(coef),
                                         _ptr(sig), _ptr(smc), n_iter, status, ll, _ptr(ws), ws.numel(), _stream())
    if rc not in (0, 4, 5, 6):
        check(rc)
    return {"coef": coef, "Sig_invMcoef": smc, "Sig_inv": sig, "n_iter": list(n_iter), "status": list(status),
            "loglik": list(ll), "rc": rc}


class RcclComm:
    """An RCCL communicator opened through the C ABI (dlsa_comm_unique_id / dlsa_comm_init_rank), for hosts that do not
    use torch.distributed: rank 0 creates `RcclComm.unique_id()`, ships the 128 bytes to the other ranks out of band, every
    rank constructs RcclComm(nranks, id, rank) after selecting its device; `allreduce(msg)` is the algorithm's one round
    of communication (dlsa/dlsa.py:30-34), in place on the current stream."""

    def __init__(self, nranks, unique_id, rank):
        lib = _lib.load()
        if not torch.cuda.is_available():
            raise RuntimeError("dlsa_amd runs on the GPU only (no CPU fallback)")
        torch.cuda.current_device()         # the HIP context of this rank's device must exist before ncclCommInitRank
        h = ctypes.c_void_p(0)
        check(lib.dlsa_comm_init_rank(ctypes.byref(h), int(nranks), bytes(unique_id), int(rank)))
        self._h, self._lib, self.nranks, self.rank = h, lib, int(nranks), int(rank)

    @staticmethod
    def unique_id():
        lib = _lib.load()
        buf = ctypes.create_string_buffer(128)
        check(lib.dlsa_comm_unique_id(buf))
        return buf.raw

    def allreduce(self, msg):
        _require_gpu(msg)
        _f64(msg, "msg")
        if msg.dim() != 1:
            raise ValueError("allreduce: msg must be a contiguous vector")
        check(self._lib.dlsa_allreduce_f64(self._h, _ptr(msg), msg.numel(), _stream()))
        return msg

    def close(self):
        if self._h:
            self._lib.dlsa_comm_destroy(self._h)
            self._h = ctypes.c_void_p(0)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
