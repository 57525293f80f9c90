"""Randomised check of the evaluation-side passes against fp64 torch arithmetic: engine.loglik (log-likelihood of c estimators in one
read, with and without the implicit intercept, strided row views), engine.xtv and engine.xtv_stats (fp64 / fp32 rows).
python bench/eval_fuzz.py cases seed"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from dlsa_amd import engine
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
worst = dict(loglik=0.0, xtv=0.0, stats=0.0)
for c in range(cases):
    p = int(rng.choice([rng.integers(1, 60), rng.integers(60, 300), rng.integers(300, 1100), rng.integers(1100, 2049)]))
    n = int(rng.choice([rng.integers(1, 3000), rng.integers(3000, 60000), rng.integers(60000, 250000)]))
    if n * p > 2e8:
        n = int(2e8 // p)
    step = int(rng.choice([1, 1, 2, 5]))
    pad = int(rng.choice([0, 0, 1, 3]))
    g = torch.Generator(device="cuda"); g.manual_seed(3000 + c)
    buf = torch.full((n * step, p + pad), float("nan"), dtype=torch.float64, device="cuda")
    Xv = buf[::step, :p]
    Xv.copy_(torch.randn((n, p), dtype=torch.float64, device="cuda", generator=g) * 0.4)
    y = (torch.rand(n, dtype=torch.float64, device="cuda", generator=g) < 0.45).double()
    icpt = bool(rng.random() < 0.5)
    ncol = int(rng.integers(1, 7))
    par = torch.randn((p + (1 if icpt else 0), ncol), dtype=torch.float64, device="cuda", generator=g) * float(rng.choice([0.05, 0.5, 3.0])) / np.sqrt(p)
    Xc = Xv.contiguous()
    A = torch.cat([torch.ones((n, 1), dtype=torch.float64, device="cuda"), Xc], 1) if icpt else Xc
    eta = A @ par
    sp = eta.clamp_min(0.0) + torch.log1p(torch.exp(-eta.abs()))
    ref = (y[:, None] * eta - sp).sum(0)
    scale = ((y[:, None] * eta).abs() + sp).sum(0) + 1.0
    out = engine.loglik(Xv, y, par, fit_intercept=icpt)
    e = float(((out - ref).abs() / scale).max())
    worst["loglik"] = max(worst["loglik"], e)
    assert e < 1e-12, ("loglik", c, n, p, step, pad, icpt, ncol, e)
    v = torch.randn(n, dtype=torch.float64, device="cuda", generator=g)
    gx, vv = engine.xtv(Xv, v)
    e = float((gx - Xc.T @ v).abs().max() / float((Xc.abs().T @ v.abs()).max() + 1e-300))
    e = max(e, abs(float(vv) - float(v @ v)) / float(v @ v + 1e-300))
    worst["xtv"] = max(worst["xtv"], e)
    assert e < 1e-12, ("xtv", c, n, p, step, pad, e)
    f32 = rng.random() < 0.4
    Xs, vs = (Xc.float(), v.float()) if f32 else (Xv, v)
    r = engine.xtv_stats(Xs, vs, want_colsum=True)
    gs, cs, st = r if isinstance(r, tuple) else (r["g"], r["colsum"], r["stats"])
    X64, v64 = Xs.double().contiguous() if f32 else Xc, vs.double()
    e = float((gs - X64.T @ v64).abs().max() / float((X64.abs().T @ v64.abs()).max() + 1e-300))
    e = max(e, float((cs - X64.sum(0)).abs().max() / float(X64.abs().sum(0).max() + 1e-300)))
    e = max(e, abs(float(st[0]) - float(v64 @ v64)) / float(v64 @ v64 + 1e-300), abs(float(st[1]) - float(v64.sum())) / float(v64.abs().sum() + 1e-300))
    worst["stats"] = max(worst["stats"], e)
    assert e < 1e-12, ("xtv_stats", c, n, p, f32, e)
    del buf, Xv, Xc, A, eta
print("EVAL FUZZ ok: %d cases, worst %s" % (cases, {k: "%.2e" % v for k, v in worst.items()}))
