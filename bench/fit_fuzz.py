"""Randomised check of the per-partition fit (engine.irls_fit / irls_fit_ex: every driver path -- pooled small kernel,
subsample levels, frozen factor + secant pairs, fused Newton pass, peeked last iteration, strided partitions, implicit
intercept) through the size-independent properties of the MLE: the score at the returned coef vanishes, Sig_inv is the Hessian
AT the returned coef, Sig_invMcoef = Sig_inv coef, loglik is the log-likelihood there.  python bench/fit_fuzz.py cases seed"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from dlsa_amd import engine
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
worst = dict(score=0.0, H=0.0, smc=0.0, ll=0.0)
paths = {}
for c in range(cases):
    p = int(rng.choice([rng.integers(2, 30), rng.integers(30, 70), 2 * rng.integers(25, 61), rng.integers(60, 130), rng.integers(130, 320),
                        rng.integers(121, 513)]))        # (round 5: the wide pass's class up to 512 columns -- own-Hessian steps on warm partitions)
    K = int(rng.choice([1, 1, 2, 3, 7, 12, 40]))          # (>= 8 partitions of a fused-class width: the lock-step driver's territory)
    per = int(rng.choice([rng.integers(40 * p, 80 * p), rng.integers(200 * p, 400 * p), rng.integers(8192, 40000), rng.integers(100000, 400000)]))
    per = max(per, 40 * p)
    if K >= 12:
        per = min(per, 60000)
    n = per * K + int(rng.integers(0, 5))
    if n * p > 3e8:
        n = int(3e8 // p); per = n // K
    kind = engine.SYNTH_GAUSSIAN if rng.random() < 0.5 else engine.SYNTH_UNIFORM
    X, y = engine.synth(int(rng.integers(1, 1 << 30)), 0, n, p, kind=kind)
    strided = K > 1 and rng.random() < 0.5
    icpt = bool(rng.random() < 0.4)
    print_case = lambda: print("CASE %d: n=%d p=%d K=%d strided=%s icpt=%s kind=%d" % (c, n, p, K, strided, icpt, kind), flush=True)
    if os.environ.get("FIT_FUZZ_VERBOSE"):
        print_case()
    # the driver is the library's choice, or forced: lock step where it applies / the chained path
    force = rng.choice(["auto", "auto", "lock", "chains"])
    opt = {} if force == "auto" else (dict(batched=True, small=False) if force == "lock" else dict(batched=False))
    ctx = engine.irls_options(**opt)
    ctx.__enter__()
    if strided:
        first, rows = list(range(K)), [(n - k + K - 1) // K for k in range(K)]
        r = engine.irls_fit_ex(X, y, first, rows, row_step=K, fit_intercept=icpt)
        parts = [(X[k::K], y[k::K]) for k in range(K)]
    elif icpt:
        offs = [int(n * k / K) for k in range(K + 1)]
        r = engine.irls_fit_ex(X, y, offs[:-1], [offs[k + 1] - offs[k] for k in range(K)], row_step=1, fit_intercept=True)
        parts = [(X[offs[k]:offs[k + 1]], y[offs[k]:offs[k + 1]]) for k in range(K)]
    else:
        offs = [int(n * k / K) for k in range(K + 1)]
        r = engine.irls_fit(X, y, offs)
        parts = [(X[offs[k]:offs[k + 1]], y[offs[k]:offs[k + 1]]) for k in range(K)]
    ctx.__exit__(None, None, None)
    key = ("strided" if strided else "ranges") + ("+icpt" if icpt else "")
    paths[key] = paths.get(key, 0) + 1
    drv = ("chains", "small", "lock step")[engine.irls_last_fit_path()]
    paths["driver: " + drv] = paths.get("driver: " + drv, 0) + 1
    assert r["status"] == [0] * K, ("status", c, n, p, K, key, r["status"], r["n_iter"])
    for k, (Xk, yk) in enumerate(parts):
        A = torch.cat([torch.ones((Xk.shape[0], 1), dtype=torch.float64, device="cuda"), Xk], 1) if (icpt) else Xk.contiguous()
        b = r["coef"][k]
        eta = A @ b
        mu = torch.sigmoid(eta)
        score = A.T @ (yk - mu)
        w = mu * (1.0 - mu)
        H = A.T @ (A * w[:, None])
        d = H.diagonal().sqrt()
        # the MLE to 1e-13 in the step: the score is H * step, measured against sqrt(H_jj) * |beta-scale| ~ H_jj^(1/2) * n^(1/2)
        es = float((score.abs() / (d * np.sqrt(A.shape[0]))).max())
        eH = float(((r["Sig_inv"][k] - H).abs() / (d[:, None] * d[None, :])).max())
        esm = float((r["Sig_invMcoef"][k] - r["Sig_inv"][k] @ b).abs().max() / float((r["Sig_inv"][k].abs() @ b.abs()).max() + 1e-300))
        sp = eta.clamp_min(0.0) + torch.log1p(torch.exp(-eta.abs()))
        ll = float((yk * eta - sp).sum())
        ell = abs(r["loglik"][k] - ll) / float(((yk * eta).abs() + sp).sum())
        worst["score"] = max(worst["score"], es); worst["H"] = max(worst["H"], eH); worst["smc"] = max(worst["smc"], esm); worst["ll"] = max(worst["ll"], ell)
        assert es < 1e-10 and eH < 1e-9 and esm < 1e-13 and ell < 1e-12, ("fit", c, n, p, K, k, key, es, eH, esm, ell, r["n_iter"])
    del X, y, parts
print("FIT FUZZ ok: %d cases, worst %s, layouts %s" % (cases, {k: "%.2e" % v for k, v in worst.items()}, paths))
