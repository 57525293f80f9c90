"""Randomised check of engine.gram in fp32 (panel + wide kernels) against an fp64 matmul of the same fp32 data."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from dlsa_amd import engine
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(5)
worst = 0.0
for c in range(cases):
    p = int(rng.choice([rng.integers(1, 300), rng.integers(300, 1100), 4 * rng.integers(190, 530), 256 * rng.integers(3, 9)]))
    n = int(rng.choice([rng.integers(1, 5000), rng.integers(20000, 90000)]))
    ld = p + int(rng.choice([0, 0, 1, 4, 8]))
    g = torch.Generator(device="cuda"); g.manual_seed(c)
    buf = torch.full((n, ld), float("nan"), dtype=torch.float32, device="cuda")
    buf[:, :p] = torch.randn((n, p), dtype=torch.float32, device="cuda", generator=g)
    X = buf[:, :p]
    w = torch.rand(n, dtype=torch.float32, device="cuda", generator=g) if rng.random() < 0.5 else None
    Xd = X.double()
    ref = Xd.T @ (Xd if w is None else Xd * w.double()[:, None])
    H = engine.gram(X, w).double()
    err = float((H - ref).abs().max() / ref.abs().max())
    worst = max(worst, err)
    if not err < 2e-5 or not torch.equal(H, H.T):
        print("FUZZ32 FAIL case %d n=%d p=%d ld=%d w=%s err=%.3e" % (c, n, p, ld, w is not None, err)); sys.exit(1)
print("FUZZ32 ok: %d cases, worst rel err %.3e" % (cases, worst))
