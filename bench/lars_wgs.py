"""LARS path time by number of workgroups: python bench/lars_wgs.py p wgs [wgs ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dlsa_amd import engine
p = int(sys.argv[1])
rng = np.random.default_rng(p)
n = 40 * p
X = rng.random((n, p)) - 0.5
S = torch.from_numpy(X.T @ ((rng.random(n) * 0.25)[:, None] * X)).cuda()
b = torch.from_numpy(np.where(np.arange(p) < 0.4 * p, 1.0, 0.0) + 0.05 * rng.standard_normal(p)).cuda()
ref = None
for w in sys.argv[2:]:
    with engine.kernel_options(lars_wgs=int(w), lars_q=0):
        engine.lars_path(S, b, False, float(n)); torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            t = time.perf_counter(); r = engine.lars_path(S, b, False, float(n)); torch.cuda.synchronize()
            ts.append((time.perf_counter() - t) * 1e3)
    beta = r["beta"].cpu().numpy()
    if ref is None: ref = beta
    print("p=%d wgs=%s: %.2f ms (%d steps) max|beta - first| %.2e" % (p, w, min(ts), beta.shape[0] - 1, np.abs(beta - ref).max() if beta.shape == ref.shape else -1))
