"""LARS lasso paths with drops on correlated designs: steps, drops and time (device) vs the 'lar' path."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dlsa_amd import engine
for p, rho in ((200, 0.98), (500, 0.98), (500, 0.995), (1000, 0.98)):
    rng = np.random.default_rng(p)
    n = 6 * p
    L = rng.standard_normal((3, p))
    X = np.sqrt(1 - rho) * rng.standard_normal((n, p)) + np.sqrt(rho) * (rng.standard_normal((n, 3)) @ L)
    S = torch.from_numpy(X.T @ ((rng.random(n) * 0.25)[:, None] * X)).cuda()
    b = torch.from_numpy(rng.standard_normal(p)).cuda()
    out = []
    for typ in ("lar", "lasso"):
        engine.lars_path(S, b, False, float(n), type=typ); torch.cuda.synchronize()
        t = time.perf_counter(); r = engine.lars_path(S, b, False, float(n), type=typ); torch.cuda.synchronize()
        out.append("%s %.2f ms, %d steps" % (typ, (time.perf_counter() - t) * 1e3, r["beta"].shape[0] - 1))
    print("p=%d rho=%.3f: %s" % (p, rho, "; ".join(out)))
