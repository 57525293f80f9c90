"""Randomised LARS / lasso paths: python tests/lars_fuzz.py [cases] [seed] [wide]      (wide = 1: widths 400 ... 1020; wide = 2: 1021 ... 2000)
(lives under tests/ because it checks against the oracle; run with 30 cases by tests/test_gpu_fuzz_smoke.py, with hundreds outside the suite)
Every case runs lars_q.hip (default build for its width, plus one forced build) and lars.hip (DLSA_LARS_Q=0) and compares the whole
path, beta0, AIC and BIC with the oracle's restatement of lsa.py:90-212."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dlsa_amd import engine
from oracle import dlsa_oracle as orc


def rel_inf(a, b):
    return float(np.max(np.abs(a - b)) / max(1e-300, np.max(np.abs(b))))


def problem(p, rho, seed):
    rng = np.random.default_rng(seed)
    n = 6 * p + 4
    L = rng.standard_normal((3, p))
    X = np.sqrt(1 - rho) * rng.standard_normal((n, p)) + np.sqrt(rho) * (rng.standard_normal((n, 3)) @ L)
    S = X.T @ ((rng.random(n) * 0.25)[:, None] * X)
    b = rng.standard_normal(p)
    if rng.random() < 0.3:
        b[rng.random(p) < 0.5] *= 1e-3
    return S, b, n


def run(S, b, intercept, n, typ, env):
    # (the switches go through dlsa_kernel_options: the library reads no environment variable for them)
    with engine.kernel_options(engine.kernel_options_from_env(env)):
        r = engine.lars_path(torch.from_numpy(S).cuda(), torch.from_numpy(b).cuda(), intercept, float(n), type=typ)
    return {k: r[k].cpu().numpy() for k in ("beta", "beta0", "AIC", "BIC")}


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 4242)
    worst = 0.0
    kinds = {}
    t0 = time.time()
    for c in range(cases):
        p = int(rng.choice([rng.integers(1, 30), rng.integers(30, 110), rng.integers(110, 260), rng.integers(260, 420)]))
        if len(sys.argv) > 3 and sys.argv[3] == "1":
            p = int(rng.choice([rng.integers(400, 520), rng.integers(520, 1021)]))
        if len(sys.argv) > 3 and sys.argv[3] == "2":
            p = int(rng.choice([rng.integers(1021, 1300), rng.integers(1300, 2001)]))
        intercept = bool(rng.random() < 0.4) and p > 1
        typ = "lasso" if rng.random() < 0.6 else "lar"
        rho = float(rng.choice([0.0, 0.5, 0.9, 0.98]))
        S, b, n = problem(p, rho, 31000 + c)
        # (DLSA_LARS_Q = 1: lars_q.hip up to 1020 variables -- the default hands m > 448 to lars_c.hip -- and lars_c.hip beyond, there with a
        # forced workgroup count)
        forced = {"DLSA_LARS_Q_THREADS": str(rng.choice([256, 512, 1024])), "DLSA_LARS_Q_LDS": str(rng.integers(0, 2)),
                  "DLSA_LARS_Q_WGS": str(rng.choice([1, 2, 3, 5, 8])), "DLSA_LARS_Q": "1"}
        if p > 1020: forced["DLSA_LARS_WGS"] = str(rng.choice([3, 16, 29, 64]))
        got = [run(S, b, intercept, n, typ, {}), run(S, b, intercept, n, typ, forced), run(S, b, intercept, n, typ, {"DLSA_LARS_Q": "0"})]
        if p - int(intercept) >= 64:       # the column-split kernel forced onto widths it does not take by default (it serves m >= 64), at a random workgroup count
            got.insert(2, run(S, b, intercept, n, typ, {"DLSA_LARS_Q": "2", "DLSA_LARS_WGS": str(rng.choice([2, 5, 16, 64]))}))
        ref = orc.lars_lsa(S, b, intercept, n, type=typ) if p <= 420 else got[-1]      # (beyond: lars.hip's R^{-1} form, a different method, is the reference)
        kinds["oracle" if p <= 420 else "kernels only"] = kinds.get("oracle" if p <= 420 else "kernels only", 0) + 1
        kinds["drops"] = kinds.get("drops", 0) + int(ref["beta"].shape[0] - 1 > p - int(intercept))
        for g in got:
            assert g["beta"].shape == ref["beta"].shape, (c, p, typ, intercept, rho, g["beta"].shape, ref["beta"].shape)
            e = max(rel_inf(g["beta"], ref["beta"]), rel_inf(g["AIC"], ref["AIC"]), rel_inf(g["BIC"], ref["BIC"]),
                    rel_inf(g["beta0"], ref["beta0"]) if intercept else 0.0)
            assert e < 1e-6, (c, p, typ, intercept, rho, e)
            worst = max(worst, e)
    print("LARS FUZZ ok: %d cases, worst relative difference %.2e, %s, %.0f s" % (cases, worst, kinds, time.time() - t0))


if __name__ == "__main__":
    main()
