"""CPU-side checks of the C-ABI library: it loads without a GPU, exports every symbol the
header declares, validates arguments, and its Gram tile plan covers every tile exactly once."""
import ctypes
import os
import re

import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    g.build()
    from dlsa_amd import _lib
    return _lib.load()


def test_every_header_symbol_is_exported_and_bound(lib):
    from dlsa_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "dlsa_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(dlsa_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    for name in declared:
        assert hasattr(lib, name)


def test_version_and_error_text(lib):
    assert lib.dlsa_version() >= 100
    # argument validation happens before any HIP call, so it works without a GPU
    rc = lib.dlsa_gram_f64(None, 4, None, 10, 4, None, 4, 0, None, 0, None)
    assert rc == 1
    from dlsa_amd import _lib
    assert "null" in _lib.last_error()
    assert lib.dlsa_gram_workspace_bytes(1000, 0, 8) == 0
    assert lib.dlsa_gram_workspace_bytes(25_000_000, 500, 8) > 0
    assert lib.dlsa_irls_workspace_bytes(1000, 50) > lib.dlsa_gram_workspace_bytes(1000, 50, 8)


@pytest.mark.parametrize("p", [1, 15, 16, 17, 50, 64, 65, 100, 127, 128, 129, 250, 256, 257, 300, 384, 385,
                               500, 512, 513, 640, 1000, 1024, 2000, 2048])
def test_gram_tile_plan_covers_upper_triangle_once(lib, p):
    items, slots, tiles = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    assert lib.dlsa_gram_plan_check(p, items, slots, tiles) == 0
    nt = (p + 15) // 16
    assert tiles.value == nt * (nt + 1) // 2           # every tile on/above the diagonal, once
    assert slots.value >= tiles.value                  # computed tile slots (diagonal blocks: 10 of 16)
    assert abs(items.value) * 4 * 16 >= slots.value    # 4 waves x 16 tile slots per workgroup (items < 0: list plan)


def test_metric_config_plan_is_perfectly_balanced(lib):
    items, slots, tiles = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    assert lib.dlsa_gram_plan_check(500, items, slots, tiles) == 0
    # 6 off-diagonal panel pairs + 3 workgroups of diagonal blocks, every wave holds a full 4x4 block
    assert (items.value, slots.value, tiles.value) == (9, 528, 528)     # no wasted tile slot at p=500


@pytest.mark.parametrize("p", [768, 772, 780, 784, 788, 1000, 1024, 1028, 1040, 1044, 1500, 2000, 2048, 2052, 3000, 4096])
def test_wide_f32_gram_plan_covers_upper_triangle_once(lib, p):
    """the unit plan of gram_wide.hip: every 16 x 16 tile on/above the diagonal is stored by exactly one unit (64 x 64 group pair,
    64 x 16 quarter strip or the plain tile's own block), every unit's groups are staged by its item"""
    items, slots, tiles = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    assert lib.dlsa_gram_wide_plan_check(p, items, slots, tiles) == 0
    nt = (p + 15) // 16
    assert tiles.value == nt * (nt + 1) // 2
    assert items.value * 8 * 37 >= slots.value >= tiles.value      # 8 waves x (two 16-tile units + a quarter unit) per workgroup
    assert tiles.value / slots.value > 0.82                        # executed MFMAs per k-step vs tiles the triangle needs


def test_wide_f32_plan_for_config5(lib):
    items, slots, tiles = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    assert lib.dlsa_gram_wide_plan_check(2000, items, slots, tiles) == 0
    # 31 groups of 64 columns + the plain tile: 21 panel pairs + 10 items of packed within-panel / leftover-group units (the lower
    # bound: 160 loose units / 16), the 31 quarter strips on four of them -- 31.6 item equivalents where round 3's panel plan ran 34
    assert items.value == 31 and tiles.value == 125 * 126 // 2
    assert slots.value == 8 * (31 * 32 + 4 * 5)
    assert tiles.value / slots.value > 0.97


def test_onehot_plan_validates_its_descriptor_without_a_gpu(lib):
    """argument / capacity checks of the structured one-hot plan run before any HIP call"""
    import numpy as np
    from dlsa_amd import _lib
    P = lambda a: ctypes.c_void_p(a.ctypes.data)
    h = ctypes.c_void_p(0)
    i32, f64 = np.int32, np.float64
    # nine dense columns: over the limit of eight
    k = np.ones(9, i32); z = np.zeros(9, f64); o = np.ones(9, f64); col = np.arange(9, dtype=i32)
    rc = lib.dlsa_onehot_plan_create(9, 9, P(k), P(col), P(z), P(o), P(col), 0, None, None, ctypes.byref(h))
    assert rc == 1 and "dense columns" in _lib.last_error()
    # 3000 levels (one of them with a column): the dense-by-level block alone exceeds the LDS budget
    # (a pair table beyond the budget is no refusal any more: it is cut into row bands)
    nl = np.array([3000], i32); lc = np.full(3000, -1, dtype=i32); lc[0] = 0
    rc = lib.dlsa_onehot_plan_create(1, 0, None, None, None, None, None, 1, P(nl), P(lc), ctypes.byref(h))
    assert rc == 1 and "too many factor levels" in _lib.last_error()
    # a design column without a source
    nl = np.array([3], i32); lc = np.array([-1, 0, 1], i32)
    rc = lib.dlsa_onehot_plan_create(3, 0, None, None, None, None, None, 1, P(nl), P(lc), ctypes.byref(h))
    assert rc == 1 and "no source" in _lib.last_error()
    assert lib.dlsa_onehot_workspace_bytes(None, 10) == 0


def test_engine_refuses_cpu_tensors():
    import torch
    from dlsa_amd import engine
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        engine.gram(torch.zeros(4, 2, dtype=torch.float64))
    if not torch.cuda.is_available():
        import numpy as np
        import dlsa_amd
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            dlsa_amd.fit_linear_chunks([(0, np.zeros((4, 2)), np.zeros(4))], 2)


def test_workspace_queries_are_monotone_in_the_row_count():
    """A workspace sized for the largest partition of a fit is reused for every smaller row count (the other partitions, the
    subsample levels), so every *_workspace_bytes query must be non-decreasing in n.  The slab counts behind them are not
    (319489 rows: 504 slabs, 319488 rows: 512): the queries return monotone bounds.  Host-only calls, no GPU."""
    import numpy as np
    from dlsa_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(5)
    queries = [("dlsa_gram_workspace_bytes", lambda n, p: lib.dlsa_gram_workspace_bytes(ctypes.c_int64(n), p, 8)),
               ("dlsa_irls_pass_workspace_bytes", lambda n, p: lib.dlsa_irls_pass_workspace_bytes(ctypes.c_int64(n), p))]
    # (dlsa_irls_workspace_bytes is a multiple of these per-pass bounds: one slice per partition chain, four up to 8 GB partitions,
    #  one beyond -- it is asked for the largest partition of the call it serves, not reused across calls)
    for p in (5, 6, 37, 64, 100, 118, 260, 500):
        sizes = sorted(set([319488, 319489, 8192, 65535, 65536, 1 << 20] + [int(v) for v in rng.integers(1, 3_000_000, 60)]))
        for name, q in queries:
            vals = [q(n, p) for n in sizes]
            assert all(v > 0 for v in vals), (name, p)
            bad = [(sizes[i], sizes[i + 1], vals[i], vals[i + 1]) for i in range(len(vals) - 1) if vals[i] > vals[i + 1]]
            assert not bad, (name, p, bad[:3])
