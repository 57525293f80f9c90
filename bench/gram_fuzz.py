"""Randomised check of engine.gram (fp64) against an fp64 matmul: random widths, row counts around the kernels' thresholds,
row pitches with NaN padding, weights on / off, accumulate.  python bench/gram_fuzz.py cases seed"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from dlsa_amd import engine
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
worst = 0.0
for c in range(cases):
    p = int(rng.choice([rng.integers(1, 49), rng.integers(49, 113), rng.integers(113, 300), rng.integers(300, 481), rng.integers(481, 509), rng.integers(509, 600)]))
    n = int(rng.choice([rng.integers(1, 9000), rng.integers(8192, 8192 + 64), rng.integers(32768, 32768 + 64), rng.integers(65536, 65536 + 64), rng.integers(30000, 200000)]))
    pad = int(rng.choice([0, 0, 1, 2, 3, 6]))
    ld = p + pad
    g = torch.Generator(device="cuda"); g.manual_seed(c)
    buf = torch.full((n, ld), float("nan"), dtype=torch.float64, device="cuda")
    buf[:, :p] = torch.randn((n, p), dtype=torch.float64, device="cuda", generator=g) * (1.0 + 0.01 * torch.arange(p, dtype=torch.float64, device="cuda"))
    X = buf[:, :p]
    w = torch.rand(n, dtype=torch.float64, device="cuda", generator=g) if rng.random() < 0.7 else None
    Xc = X.contiguous()
    ref = Xc.T @ (Xc if w is None else Xc * w[:, None])
    if rng.random() < 0.3:
        H0 = torch.randn((p, p), dtype=torch.float64, device="cuda", generator=g); H0 = H0 + H0.T
        H = H0.clone(); engine.gram(X, w, out=H, accumulate=True); ref = ref + H0
    else:
        H = engine.gram(X, w)
    d = Xc.pow(2).sum(0).sqrt() if w is None else (Xc.pow(2) * w[:, None]).sum(0).sqrt()
    scale = (d[:, None] * d[None, :]).clamp_min(1e-300) + ref.abs()
    err = float(((H - ref).abs() / scale).max())
    worst = max(worst, err)
    sym = torch.equal(H, H.T) or w is not None and False
    if not (err < 1e-12) or not torch.equal(H, H.T):
        print("FUZZ FAIL case %d: n=%d p=%d ld=%d w=%s err=%.3e sym=%s" % (c, n, p, ld, w is not None, err, torch.equal(H, H.T)))
        sys.exit(1)
print("FUZZ ok: %d cases, worst scaled error %.3e" % (cases, worst))
