"""The column-split LARS kernel (lars_c.hip, 1021 <= m <= 2044) against lars.hip's grid kernel on the same problems, and their times.
   python bench/lars_c_check.py [p ...]      LARS_C_WGS="a b c": also time these workgroup counts"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dlsa_amd import engine


def rel_inf(a, b):
    return float(np.max(np.abs(a - b)) / max(1e-300, np.max(np.abs(b))))


def problem(p, rho, seed):
    rng = np.random.default_rng(seed)
    n = 6 * p + 4
    L = rng.standard_normal((3, p))
    X = np.sqrt(1 - rho) * rng.standard_normal((n, p)) + np.sqrt(rho) * (rng.standard_normal((n, 3)) @ L)
    S = X.T @ ((rng.random(n) * 0.25)[:, None] * X)
    return S, rng.standard_normal(p), n


def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t = time.perf_counter(); r = fn(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
    return sorted(ts)[len(ts) // 2], r


for p in ([int(v) for v in sys.argv[1:]] or [1100, 1536, 2000]) if __name__ == "__main__" else []:
    for intercept, typ in ((False, "lar"), (True, "lasso")):
        S, b, n = problem(p, 0.5, 777 + p)
        St, bt = torch.from_numpy(S).cuda(), torch.from_numpy(b).cuda()
        ms, r = timed(lambda: engine.lars_path(St, bt, intercept, float(n), type=typ))
        with engine.kernel_options(lars_q=0):
            ms0, r0 = timed(lambda: engine.lars_path(St, bt, intercept, float(n), type=typ))
        assert r["beta"].shape == r0["beta"].shape, (r["beta"].shape, r0["beta"].shape)
        e = max(rel_inf(r[k].cpu().numpy(), r0[k].cpu().numpy()) for k in ("beta", "AIC", "BIC"))
        extra = ""
        for w in [int(v) for v in os.environ.get("LARS_C_WGS", "").split()]:
            with engine.kernel_options(lars_wgs=w):
                msw, rw = timed(lambda: engine.lars_path(St, bt, intercept, float(n), type=typ))
            assert rw["beta"].shape == r["beta"].shape and rel_inf(rw["beta"].cpu().numpy(), r["beta"].cpu().numpy()) < 1e-8
            extra += " | %d wgs %.1f ms" % (w, msw)
        print("p=%d %s intercept=%d: column split %.1f ms, lars.hip grid %.1f ms, %d steps, max relative difference %.1e%s" % (
            p, typ, intercept, ms, ms0, r["beta"].shape[0] - 1, e, extra), flush=True)
        assert e < 1e-7
