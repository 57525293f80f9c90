"""Randomised check of the linear map step (fit_linear_partitions: strided / range partitions, implicit intercept, fp64 / fp32
rows, any width up to 2100) against fp64 torch arithmetic on the same rows.  python bench/linear_fuzz.py cases seed"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import dlsa_amd
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
worst = dict(H=0.0, g=0.0, coef=0.0)
for c in range(cases):
    f32 = rng.random() < 0.4
    p = int(rng.choice([rng.integers(1, 40), rng.integers(40, 130), rng.integers(130, 600), rng.integers(760, 1100) if f32 else rng.integers(400, 600), 4 * rng.integers(192, 513) if f32 else rng.integers(2, 64)]))
    K = int(rng.choice([1, 2, 3, 5]))
    per = int(rng.choice([rng.integers(max(2 * p, 8), 6 * p + 16), rng.integers(3000, 20000), rng.integers(20000, 90000)]))
    per = max(per, 2 * p + 4)
    if per * K * p > 1.5e8:
        per = int(1.5e8 // (K * p)); per = max(per, p + 8)
    n = per * K + int(rng.integers(0, 4))
    g = torch.Generator(device="cuda"); g.manual_seed(5000 + c)
    dt = torch.float32 if f32 else torch.float64
    X = (torch.rand((n, p), dtype=torch.float64, device="cuda", generator=g) - 0.5).to(dt)
    beta = torch.zeros(p, dtype=torch.float64, device="cuda"); beta[: max(1, int(0.4 * p))] = 1.0
    icpt = bool(rng.random() < 0.5)
    y = (X.double() @ beta + (0.3 if icpt else 0.0) + torch.randn(n, dtype=torch.float64, device="cuda", generator=g)).to(dt)
    strided = K > 1 and rng.random() < 0.5
    if os.environ.get("FUZZ_VERBOSE"):
        import time as _t
        print("CASE %d n=%d p=%d K=%d f32=%s icpt=%s strided=%s t=%.1f" % (c, n, p, K, f32, icpt, strided, _t.time() % 1000), flush=True)
    if strided:
        mb = dlsa_amd.fit_linear_partitions(X, y, partition_num=K, fit_intercept=icpt)
        parts = [(X[k::K], y[k::K]) for k in range(K)]
    else:
        offs = [int(n * k / K) for k in range(K + 1)]
        mb = dlsa_amd.fit_linear_partitions(X, y, part_offsets=offs, fit_intercept=icpt)
        parts = [(X[offs[k]:offs[k + 1]], y[offs[k]:offs[k + 1]]) for k in range(K)]
    tolH, tolg = (1e-5, 2e-5) if f32 else (1e-12, 1e-11)          # fp32 rows: fp32 products summed in fp32 inside a slab (~sqrt(rows) eps)
    for k, (Xk, yk) in enumerate(parts):
        A = Xk.double()
        if icpt:
            A = torch.cat([torch.ones((A.shape[0], 1), dtype=torch.float64, device="cuda"), A], 1)
        H = A.T @ A
        gk = A.T @ yk.double()
        d = H.diagonal().sqrt()
        eH = float(((mb.Sig_inv[k] - H).abs() / (d[:, None] * d[None, :])).max())
        eg = float((mb.Sig_invMcoef[k] - gk).abs().max() / float((A.abs().T @ yk.double().abs()).max()))
        worst["H"] = max(worst["H"], eH) if not f32 else worst["H"]; worst["g"] = max(worst["g"], eg) if not f32 else worst["g"]
        assert eH < tolH and eg < tolg, ("linear", c, n, p, K, k, strided, icpt, f32, eH, eg)
        if mb.status[k] == 0 and not f32 and A.shape[0] > 3 * A.shape[1]:
            ref = torch.linalg.lstsq(A, yk.double()[:, None]).solution[:, 0]
            ec = float((mb.coef[k] - ref).abs().max() / ref.abs().max())
            worst["coef"] = max(worst["coef"], ec)
            assert ec < 1e-7, ("coef", c, n, p, K, k, ec)
    del X, y, parts, mb
print("LINEAR FUZZ ok: %d cases, worst (fp64 cases) %s" % (cases, {k: "%.2e" % v for k, v in worst.items()}))
