"""GPU: the plan-driven fp64 Gram kernel (gram_plan.hip: generated per-wave tile plans on 1 / 2 / 4-CU groups, interleaved
tile pairs, 125 <= p <= 572) against an fp64 matmul over the whole width range -- every group size, tail-group count and
tile-count parity -- with and without weights, odd p, padded NaN pitches, ragged row counts."""
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def eng():
    assert torch.cuda.is_available()
    from dlsa_amd import engine
    return engine


def check(eng, n, p, seed, pad=0):
    gen = torch.Generator(device="cuda"); gen.manual_seed(seed)
    ld = p + (p & 1) + pad                                     # even row pitch (DMA path)
    buf = torch.full((n, ld), float("nan"), dtype=torch.float64, device="cuda")
    # distinct column scales catch a transposed / misplaced tile
    buf[:, :p] = torch.randn((n, p), dtype=torch.float64, device="cuda", generator=gen) * (
        1.0 + 0.01 * torch.arange(p, dtype=torch.float64, device="cuda"))
    X = buf[:, :p]
    w = torch.rand(n, dtype=torch.float64, device="cuda", generator=gen) * 0.25
    Xc = X.contiguous()
    for wt in (w, None):
        H = eng.gram(X, wt)
        assert torch.equal(H, H.T)
        ref = Xc.T @ (Xc if wt is None else Xc * wt[:, None])
        assert float((H - ref).abs().max()) < 1e-12 * float(ref.abs().max()), (p, wt is None)
        d = ref.diagonal().sqrt()                              # entry (i, j) on its own scale sqrt(H_ii H_jj)
        assert float(((H - ref).abs() / (d[:, None] * d[None, :])).max()) < 1e-12, (p, wt is None)


# every tile count 8 .. 35 with tail groups 0 .. 3 somewhere; the steps of 13 / 7 walk through all residues mod 16
@pytest.mark.parametrize("p", sorted(set(list(range(125, 573, 13)) + list(range(128, 573, 16)) +
                                         [260, 284, 285, 286, 300, 399, 400, 401, 480, 481, 496, 497, 500, 501, 504, 508, 509,
                                          512, 528, 556, 560, 564, 568, 570, 571, 572])))
def test_gram_plan_matches_fp64_matmul(eng, p):
    check(eng, 36000 + 3 * p + (p % 5), p, p, pad=(2 if p % 3 == 0 else 0))


@pytest.mark.parametrize("p,n", [(260, 32768), (260, 32769), (500, 65536 + 7), (500, 262144 + 1), (400, 100003), (300, 77777)])
def test_gram_plan_ragged_rows(eng, p, n):
    check(eng, n, p, n % 1000)


def test_gram_plan_linearity_and_accumulate(eng):
    n, p = 90000, 500
    X = torch.randn((n, p), dtype=torch.float64, device="cuda")
    w = torch.rand(n, dtype=torch.float64, device="cuda")
    H = eng.gram(X, w)
    cut = 45001
    H2 = eng.gram(X[:cut], w[:cut]) + eng.gram(X[cut:], w[cut:])
    assert float((H2 - H).abs().max()) < 1e-11 * float(H.abs().max())
    # the same call twice gives the same bits (static plans, fixed reduction order)
    assert torch.equal(H, eng.gram(X, w))


# ---------------------------------------------------------------------------------------------------------------------
# The pinned checker: oracle.gram (dlsa/models.py:130 restated) -- every instantiation of the metric's kernel, and one case
# each of the plan and narrow kernels.  dlsa_gram_last_kernel tells which kernel variant the launch really took.
# ---------------------------------------------------------------------------------------------------------------------
def _oracle_case(eng, n, p, seed, weighted, nan_pitch):
    import numpy as np
    from oracle import dlsa_oracle as orc
    rng = np.random.default_rng(seed)
    Xh = rng.standard_normal((n, p)) * (1.0 + 0.01 * np.arange(p))          # distinct column scales: a misplaced tile shows
    wh = rng.random(n) * 0.25 if weighted else None
    ld = p + (p & 1) + (2 if nan_pitch else 0)
    buf = torch.full((n, ld), float("nan"), dtype=torch.float64, device="cuda")      # whatever follows a row must not matter
    buf[:, :p] = torch.from_numpy(Xh).cuda()
    X = buf[:, :p]
    wd = torch.from_numpy(wh).cuda() if weighted else None
    ref = orc.gram(Xh, wh)
    d = np.sqrt(np.diag(ref))
    return X, wd, ref, d


def _against_oracle(H, ref, d, what):
    import numpy as np
    Hh = H.cpu().numpy()
    assert np.array_equal(Hh, Hh.T), what
    assert np.max(np.abs(Hh - ref)) < 1e-12 * np.max(np.abs(ref)), what
    assert np.max(np.abs(Hh - ref) / (d[:, None] * d[None, :])) < 1e-12, what      # entry (i, j) on its own scale


# p -> tail groups G of gram_cyclic_kernel<HASW, G>: 481..496 -> 0 (481 = the odd width whose pad column is loaded),
# 497..500 -> 1, 501..504 -> 2, 505..508 -> 3
@pytest.mark.parametrize("p,G", [(481, 0), (488, 0), (496, 0), (497, 1), (500, 1), (502, 2), (504, 2), (505, 3), (507, 3), (508, 3)])
@pytest.mark.parametrize("weighted", [True, False])
def test_gram_cyclic_every_instantiation_matches_oracle(eng, p, G, weighted):
    """gram_cyclic_kernel<{true,false},{0,1,2,3}> at n >= 65 536 (below that the plan kernel serves the width) against the
    pinned oracle, NaN-padded pitch on the even widths, and accumulate."""
    n = 65536 + 8 * p + (p % 7)
    X, wd, ref, d = _oracle_case(eng, n, p, 1000 + p, weighted, nan_pitch=(p % 2 == 0 and p % 4 != 0))
    H = eng.gram(X, wd)
    name, cycles = eng.gram_last_kernel(want_cycles=True)
    assert name == "gram_cyclic_kernel<%s,%d>" % ("true" if weighted else "false", G), name
    assert cycles > 0                                     # the clock probe of the launch (bench.py's shader_clock_GHz)
    _against_oracle(H, ref, d, (p, weighted))
    H2 = eng.gram(X, wd, out=H.clone(), accumulate=True)
    _against_oracle(H2 * 0.5, ref, d, (p, weighted, "accumulate"))


@pytest.mark.parametrize("p,n,kernel", [(260, 40000, "gram_plan_kernel<true,16,1>"), (400, 50000, "gram_plan_kernel<true,25,0>"),
                                        (566, 36000, "gram_plan_kernel<true,35,2>"),
                                        # round 3: the two shapes the plan kernel used to leave to the panel kernel
                                        (572, 36000, "gram_plan_kernel<true,35,3>"), (284, 36000, "gram_plan_kernel<true,18,0>"),
                                        (282, 40001, "gram_plan_kernel<true,18,0>"),
                                        (100, 30000, "gram_narrow_kernel<true,6,1>"), (50, 20000, "gram_narrow_kernel<true,3,1>"),
                                        (112, 20000, "gram_narrow_kernel<true,7,0>")])
def test_plan_and_narrow_kernels_match_oracle(eng, p, n, kernel):
    X, wd, ref, d = _oracle_case(eng, n, p, 2000 + p, True, nan_pitch=True)
    H = eng.gram(X, wd)
    name, cycles = eng.gram_last_kernel(want_cycles=True)
    assert name.startswith(kernel), name
    assert cycles > 0
    _against_oracle(H, ref, d, p)


def test_last_kernel_names_the_panel_kernel_too(eng):
    X = torch.randn((5000, 40), dtype=torch.float64, device="cuda")
    eng.gram(X)
    name, cycles = eng.gram_last_kernel(want_cycles=True)
    assert name.startswith("gram_kernel<double,false") and cycles == 0, name
    X32 = torch.randn((40000, 1024), dtype=torch.float32, device="cuda")
    eng.gram(X32)
    assert eng.gram_last_kernel()[0] == "gram_wide_f32_kernel<false>"
