"""The LARS grid kernel (lars.hip, m > 1020) against its single-workgroup form on the same problems, and its time.
   python bench/lars_grid_check.py [p ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dlsa_amd import engine


def rel_inf(a, b):
    return float(np.max(np.abs(a - b)) / max(1e-300, np.max(np.abs(b))))


def problem(p, rho, seed):
    """(the generator of tests/lars_fuzz.py, restated: bench scripts do not import the oracle)"""
    rng = np.random.default_rng(seed)
    n = 6 * p + 4
    L = rng.standard_normal((3, p))
    X = np.sqrt(1 - rho) * rng.standard_normal((n, p)) + np.sqrt(rho) * (rng.standard_normal((n, 3)) @ L)
    S = X.T @ ((rng.random(n) * 0.25)[:, None] * X)
    b = rng.standard_normal(p)
    if rng.random() < 0.3:
        b[rng.random(p) < 0.5] *= 1e-3
    return S, b, n


for p in [int(v) for v in sys.argv[1:]] or [1100, 1536, 2000]:
    for intercept, typ in ((False, "lar"), (True, "lasso")):
        S, b, n = problem(p, 0.5, 777 + p)
        St, bt = torch.from_numpy(S).cuda(), torch.from_numpy(b).cuda()
        engine.lars_path(St, bt, intercept, float(n), type=typ); torch.cuda.synchronize()
        t = time.perf_counter(); r = engine.lars_path(St, bt, intercept, float(n), type=typ); torch.cuda.synchronize()
        ms = (time.perf_counter() - t) * 1e3
        with engine.kernel_options(lars_wgs=1):
            t = time.perf_counter(); r1 = engine.lars_path(St, bt, intercept, float(n), type=typ); torch.cuda.synchronize()
            ms1 = (time.perf_counter() - t) * 1e3
        assert r["beta"].shape == r1["beta"].shape
        e = max(rel_inf(r[k].cpu().numpy(), r1[k].cpu().numpy()) for k in ("beta", "AIC", "BIC"))
        print("p=%d %s intercept=%d: grid %.1f ms, one workgroup %.1f ms, %d steps, max relative difference %.1e" % (p, typ, intercept, ms, ms1, r["beta"].shape[0] - 1, e), flush=True)
        assert e < 1e-9
