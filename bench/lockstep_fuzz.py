"""Randomised check of the lock-step driver (csrc/irls_batch.hip): ragged partition sizes, many partitions, the pooled start and the
gradient-only passes on and off, both tolerances, contiguous and i % K partitions, with and without the intercept -- through the
size-independent properties of the MLE (bench/fit_fuzz.py): the score at the returned coef vanishes, Sig_inv is the Hessian AT it,
Sig_invMcoef = Sig_inv coef, loglik is the log-likelihood there.   python bench/lockstep_fuzz.py cases seed"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from dlsa_amd import engine
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
worst = dict(score=0.0, H=0.0, smc=0.0, ll=0.0)
seen = {}
for c in range(cases):
    icpt = bool(rng.random() < 0.4)
    p = int(2 * rng.integers(25, 56)) if icpt else int(rng.integers(49, 113))          # (with the intercept: even p, p + 1 <= 112)
    K = int(rng.choice([8, 9, 20, 64, 200, 600]))
    strided = bool(rng.random() < 0.35) and not (p & 1)
    budget = int(2.0e8 // p)                                                           # rows in all
    lo = 40 * (p + 1) + 100
    if strided:
        n = int(min(budget, K * rng.integers(lo, 30000))) + int(rng.integers(0, K))
        first, rows = list(range(K)), [(n - k + K - 1) // K for k in range(K)]
    else:
        kinds = rng.integers(0, 3, size=K)
        rows = [int(rng.integers(lo, lo + 1500)) if t == 0 else int(rng.integers(lo, 20000)) if t == 1 else int(rng.integers(20000, 70000)) for t in kinds]
        scale = min(1.0, budget / float(sum(rows)))
        rows = [max(lo, int(r * scale)) for r in rows]
        first = np.concatenate([[0], np.cumsum(rows)[:-1]]).astype(np.int64).tolist()
        n = int(sum(rows))
    if (p & 1) and not strided:
        pass                                                                           # odd widths: packed rows (engine.synth packs 49..120)
    X, y = engine.synth(int(rng.integers(1, 1 << 30)), 0, n, p, kind=engine.SYNTH_GAUSSIAN if rng.random() < 0.6 else engine.SYNTH_UNIFORM)
    tol = float(rng.choice([1e-10, 1e-13]))
    opt = dict(batched=True, small=False)
    v = rng.integers(0, 4)
    if v == 1: opt["pooled_start"] = False
    if v == 2: opt["grad_passes"] = int(rng.integers(0, 3))
    if v == 3: opt["subsample_div"] = int(rng.choice([0, 4, 8]))
    if os.environ.get("FUZZ_VERBOSE"):
        print("CASE %d: n=%d p=%d K=%d strided=%s icpt=%s tol=%g opt=%s rows %d..%d" % (c, n, p, K, strided, icpt, tol, opt, min(rows), max(rows)), flush=True)
    with engine.irls_options(**opt):
        r = engine.irls_fit_ex(X, y, first, rows, row_step=K if strided else 1, fit_intercept=icpt, tol=tol)
    path = engine.irls_last_fit_path()
    key = ("lock step" if path == 2 else "other driver %d" % path) + (" i%K" if strided else "") + (" icpt" if icpt else "")
    seen[key] = seen.get(key, 0) + 1
    assert r["status"] == [0] * K, ("status", c, n, p, K, key, opt, r["status"][:10], r["n_iter"][:10])
    pick = sorted(set([0, K - 1, int(np.argmin(rows)), int(np.argmax(rows))] + [int(v) for v in rng.integers(0, K, size=4)]))
    for k in pick:
        Xk = X[k::K][:rows[k]] if strided else X[first[k]:first[k] + rows[k]]
        yk = y[k::K][:rows[k]] if strided else y[first[k]:first[k] + rows[k]]
        A = torch.cat([torch.ones((Xk.shape[0], 1), dtype=torch.float64, device="cuda"), Xk], 1) if icpt else Xk.contiguous()
        b = r["coef"][k]
        eta = A @ b
        mu = torch.sigmoid(eta)
        score = A.T @ (yk - mu)
        w = mu * (1.0 - mu)
        H = A.T @ (A * w[:, None])
        d = H.diagonal().sqrt()
        es = float((score.abs() / (d * np.sqrt(A.shape[0]))).max())
        eH = float(((r["Sig_inv"][k] - H).abs() / (d[:, None] * d[None, :])).max())
        esm = float((r["Sig_invMcoef"][k] - r["Sig_inv"][k] @ b).abs().max() / float((r["Sig_inv"][k].abs() @ b.abs()).max() + 1e-300))
        sp = eta.clamp_min(0.0) + torch.log1p(torch.exp(-eta.abs()))
        ll = float((yk * eta - sp).sum())
        ell = abs(r["loglik"][k] - ll) / float(((yk * eta).abs() + sp).sum())
        for nm, val in (("score", es), ("H", eH), ("smc", esm), ("ll", ell)):
            worst[nm] = max(worst[nm], val)
        lim = 1e-10 if tol <= 1e-12 else 1e-8                     # (tol 1e-10 on the step leaves a score of that order)
        assert es < lim and eH < 1e-9 and esm < 1e-13 and ell < 1e-12, ("fit", c, n, p, K, k, key, opt, tol, es, eH, esm, ell, r["n_iter"][:8])
    del X, y
print("LOCK-STEP FUZZ ok: %d cases, worst %s, %s" % (cases, {k: "%.2e" % v for k, v in worst.items()}, seen))
