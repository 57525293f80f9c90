"""GPU: the fp64 Gram at config 4's dense widths (125 <= p <= 284: the single-CU plans of gram_plan.hip) against an fp64
matmul over the whole range (every tile count / tail-group combination), with and without weights, odd p, padded NaN pitches."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def rel_inf(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return float(np.max(np.abs(a - b)) / max(1e-300, np.max(np.abs(b))))


@pytest.fixture(scope="module")
def eng():
    assert torch.cuda.is_available()
    from dlsa_amd import engine
    return engine


@pytest.mark.parametrize("p", list(range(125, 285, 7)) + [128, 144, 256, 260, 272, 276, 280, 284, 129, 259])
def test_gram_midwidth_matches_fp64_matmul(eng, p):
    n = 40000 + p
    gen = torch.Generator(device="cuda"); gen.manual_seed(p)
    ld = p + (p & 1) + (2 if p % 3 == 0 else 0)               # even row pitch (DMA path), sometimes with padding columns
    buf = torch.full((n, ld), float("nan"), dtype=torch.float64, device="cuda")
    buf[:, :p] = torch.randn((n, p), dtype=torch.float64, device="cuda", generator=gen)
    X = buf[:, :p]
    w = torch.rand(n, dtype=torch.float64, device="cuda", generator=gen) * 0.25
    for wt in (w, None):
        H = eng.gram(X, wt)
        assert torch.equal(H, H.T)
        Xw = X.contiguous() if wt is None else X * wt[:, None]
        ref = X.contiguous().T @ Xw
        assert float((H - ref).abs().max()) < 1e-12 * float(ref.abs().max()), (p, wt is None)


def test_gram_midwidth_asymmetric_columns_and_linearity(eng):
    """Distinct column scales catch a transposed / misplaced tile; two row blocks must add up."""
    n, p = 50000, 260
    gen = torch.Generator(device="cuda"); gen.manual_seed(260)
    X = torch.randn((n, p), dtype=torch.float64, device="cuda", generator=gen) * torch.arange(1, p + 1, dtype=torch.float64, device="cuda")
    w = torch.rand(n, dtype=torch.float64, device="cuda", generator=gen)
    H = eng.gram(X, w)
    ref = X.T @ (X * w[:, None])
    # entry (i, j) on ITS scale sqrt(H_ii H_jj) (an off-diagonal sum of n zero-mean products can cancel to nearly nothing, so
    # |ref_ij| itself is no scale): a misplaced tile is off by O(1) here, rounding by O(1e-14)
    d = ref.diagonal().sqrt()
    assert float(((H - ref).abs() / (d[:, None] * d[None, :])).max()) < 1e-12
    assert float((H - ref).abs().max()) < 1e-12 * float(ref.abs().max())
    cut = 33001
    H2 = eng.gram(X[:cut], w[:cut]) + eng.gram(X[cut:], w[cut:])          # the second block takes the panel kernel (n < 32768)
    assert float((H2 - H).abs().max()) < 1e-11 * float(H.abs().max())
