"""GPU: randomised shapes through `engine.gram` (fp64) against an fp64 matmul -- widths across every kernel's range (panel,
row-split, plan, cyclic), row counts around the kernels' thresholds, padded NaN row pitches (even and odd), weights on / off,
accumulate.  Seeded: the same 80 cases every run (bench/gram_fuzz.py runs more)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def test_gram_random_shapes_match_fp64_matmul():
    assert torch.cuda.is_available()
    from dlsa_amd import engine
    rng = np.random.default_rng(20260202)
    for c in range(80):
        p = int(rng.choice([rng.integers(1, 49), rng.integers(49, 113), rng.integers(113, 300), rng.integers(300, 481),
                            rng.integers(481, 509), rng.integers(509, 600)]))
        n = int(rng.choice([rng.integers(1, 9000), rng.integers(8192, 8192 + 64), rng.integers(32768, 32768 + 64),
                            rng.integers(65536, 65536 + 64), rng.integers(30000, 120000)]))
        ld = p + int(rng.choice([0, 0, 1, 2, 3, 6]))
        g = torch.Generator(device="cuda"); g.manual_seed(c)
        buf = torch.full((n, ld), float("nan"), dtype=torch.float64, device="cuda")
        buf[:, :p] = torch.randn((n, p), dtype=torch.float64, device="cuda", generator=g) * (
            1.0 + 0.01 * torch.arange(p, dtype=torch.float64, device="cuda"))
        X = buf[:, :p]
        w = torch.rand(n, dtype=torch.float64, device="cuda", generator=g) if rng.random() < 0.7 else None
        Xc = X.contiguous()
        ref = Xc.T @ (Xc if w is None else Xc * w[:, None])
        if rng.random() < 0.3:
            H0 = torch.randn((p, p), dtype=torch.float64, device="cuda", generator=g); H0 = H0 + H0.T
            H = H0.clone(); engine.gram(X, w, out=H, accumulate=True); ref = ref + H0
        else:
            H = engine.gram(X, w)
        d = (Xc.pow(2) if w is None else Xc.pow(2) * w[:, None]).sum(0).sqrt()
        scale = (d[:, None] * d[None, :]).clamp_min(1e-300) + ref.abs()       # entry (i, j) on its own scale
        assert float(((H - ref).abs() / scale).max()) < 1e-12, (c, n, p, ld, w is not None)
        assert torch.equal(H, H.T), (c, n, p, ld)
