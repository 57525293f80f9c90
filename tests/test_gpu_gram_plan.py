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
