"""Time the LARS kernel (device path of lars_lsa) for a p x p logistic Hessian."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dlsa_amd import engine
for p in (50, 200, 500, 1000):
    rng = np.random.default_rng(p)
    n = 40 * p
    X = rng.random((n, p)) - 0.5
    S = torch.from_numpy(X.T @ ((rng.random(n) * 0.25)[:, None] * X)).cuda()
    b = torch.from_numpy(np.where(np.arange(p) < 0.4 * p, 1.0, 0.0) + 0.05 * rng.standard_normal(p)).cuda()
    engine.lars_path(S, b, False, float(n)); torch.cuda.synchronize()
    ts = []
    for typ in ("lar", "lasso"):
        reps = []
        for _ in range(4):
            t = time.perf_counter(); r = engine.lars_path(S, b, False, float(n), type=typ); torch.cuda.synchronize()
            reps.append((time.perf_counter() - t) * 1e3)
        ts.append((typ, min(reps), max(reps), r["beta"].shape[0] - 1))
    print("p=%d: " % p + "  ".join("%s %.2f ms (max %.2f; %d steps)" % t for t in ts))
