"""Edge-case LSA problems on the wide LARS kernels: all-zero columns, exact ties of |Cvec| (several variables enter in one step,
lsa.py:130-149), tiny b0 entries, strongly correlated columns -- lars_c.hip (default) against lars.hip (lars_q = 0).
(EXACTLY duplicated columns are left out on purpose: there the reference algorithm itself is chaotic -- the duplicate's pivot r_pp^2 is
rounding noise ~1e-12 >> eps, so it is appended with 1 / r_pp ~ 1e6, and the numpy restatement of lsa.py runs to max_steps with
|beta| ~ 1e16 .. 1e41; every implementation, the oracle included, returns a different garbage path.  Checked in round 6, p = 200 / 300.)
   python bench/lars_degenerate_check.py [cases seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dlsa_amd import engine
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
worst, kinds = 0.0, {}
for c in range(cases):
    p = int(rng.choice([rng.integers(450, 700), rng.integers(700, 1021), rng.integers(1021, 1400)]))
    n = 4 * p
    X = rng.standard_normal((n, p))
    kind = str(rng.choice(["corr", "zerocol", "ties", "tinyb"]))
    if kind == "corr":
        for _ in range(int(rng.integers(1, 6))):
            a, b = rng.integers(0, p, 2)
            if a != b: X[:, b] = X[:, a] + 1e-2 * rng.standard_normal(n)          # correlation 0.99995: a small but honest pivot
    elif kind == "zerocol":
        X[:, rng.integers(0, p, 3)] = 0.0
    S = X.T @ ((rng.random(n) * 0.25 + 0.01)[:, None] * X)
    b = rng.standard_normal(p)
    if kind == "ties":
        S = np.eye(p) * 3.0; S[0, 1] = S[1, 0] = 0.5
        b = np.sign(rng.standard_normal(p)) * np.repeat(rng.random(p // 8 + 1) + 0.5, 8)[:p]      # groups of eight equal |Cvec|
    if kind == "tinyb":
        b[rng.random(p) < 0.3] *= 1e-12
    if kind == "zerocol":
        S[np.diag_indices(p)] += 0.0
    intercept = bool(rng.random() < 0.3) and kind != "ties"
    typ = "lasso" if rng.random() < 0.6 else "lar"
    St, bt = torch.from_numpy(S).cuda(), torch.from_numpy(b).cuda()
    try:
        r = engine.lars_path(St, bt, intercept, float(n), type=typ)
        with engine.kernel_options(lars_q=0):
            r0 = engine.lars_path(St, bt, intercept, float(n), type=typ)
    except Exception as e:
        print("case %d p=%d %s %s intercept=%d raised %r" % (c, p, kind, typ, intercept, e)); raise
    assert r["beta"].shape == r0["beta"].shape, (c, p, kind, typ, intercept, tuple(r["beta"].shape), tuple(r0["beta"].shape))
    A, B = r["beta"].cpu().numpy(), r0["beta"].cpu().numpy()
    ok = np.isfinite(A) == np.isfinite(B)
    assert ok.all(), (c, p, kind)
    m = np.isfinite(B)
    e = float(np.max(np.abs(A[m] - B[m])) / max(1e-300, np.max(np.abs(B[m]))))
    assert e < 1e-6, (c, p, kind, typ, intercept, e)
    worst = max(worst, e); kinds[kind] = kinds.get(kind, 0) + 1
print("LARS DEGENERATE ok: %d cases, worst relative difference %.2e, %s" % (cases, worst, kinds))
