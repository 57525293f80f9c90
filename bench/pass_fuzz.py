"""Randomised check of the Newton pass entries (engine.irls_pass: fused kernel where eligible; engine.logit_pass: ring /
register kernels) against fp64 torch arithmetic: random widths 2..600, row counts around the kernels' thresholds, row pitches with
NaN padding, label offsets, beta scales up to |eta| ~ 40.  python bench/pass_fuzz.py cases seed"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from dlsa_amd import engine
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
worst = dict(w=0.0, g=0.0, ll=0.0, H=0.0)
kinds = {}
for c in range(cases):
    p = int(rng.choice([rng.integers(2, 49), 2 * rng.integers(25, 61), rng.integers(49, 121), rng.integers(121, 300), rng.integers(300, 600)]))
    n = int(rng.choice([rng.integers(1, 9000), rng.integers(8192, 8192 + 70), rng.integers(30000, 140000), rng.integers(65536, 65536 + 64)]))
    pad = int(rng.choice([0, 0, 1, 2, 4]))
    ld = p + pad
    g = torch.Generator(device="cuda"); g.manual_seed(1000 + c)
    buf = torch.full((n + 1, ld), float("nan"), dtype=torch.float64, device="cuda")
    off = int(rng.integers(0, 2))                                   # rows start 0 or 1 row into the allocation
    X = buf[off:off + n, :p]
    X.copy_(torch.randn((n, p), dtype=torch.float64, device="cuda", generator=g) * 0.3)
    ybuf = torch.empty(n + 1, dtype=torch.float64, device="cuda")
    y = ybuf[off:off + n]
    y.copy_((torch.rand(n, dtype=torch.float64, device="cuda", generator=g) < 0.4).double())
    beta = torch.randn(p, dtype=torch.float64, device="cuda", generator=g) * float(rng.choice([0.0, 0.1, 1.0, 8.0])) / np.sqrt(p) * 3.0
    Xc = X.contiguous()
    eta = Xc @ beta
    mu = torch.sigmoid(eta)
    w_ref = torch.exp(-eta.abs()) / (1.0 + torch.exp(-eta.abs())) ** 2
    g_ref = Xc.T @ (y - mu)
    sp = eta.clamp_min(0.0) + torch.log1p(torch.exp(-eta.abs()))          # (torch's softplus switches to x above 20: 2e-9 off)
    ll_ref = float((y * eta - sp).sum())
    ll_scale = float(((y * eta).abs() + sp).sum()) + 1.0      # the sum cancels: errors relative to its terms
    H_ref = Xc.T @ (Xc * w_ref[:, None])
    gs = float((Xc.abs().sum(0)).max()) + 1e-300
    if rng.random() < 0.5:
        H, gg, ll, w = engine.irls_pass(X, y, beta, want_w=bool(rng.random() < 0.5))
        name = engine.gram_last_kernel()[0]
        d = (Xc.pow(2) * w_ref[:, None]).sum(0).sqrt()
        eH = float(((H - H_ref).abs() / ((d[:, None] * d[None, :]).clamp_min(1e-300) + H_ref.abs())).max())
        worst["H"] = max(worst["H"], eH)
        assert eH < 1e-11 and torch.equal(H, H.T), ("H", c, n, p, ld, off, eH, name)
    else:
        w, gg, ll = engine.logit_pass(X, y, beta)
        name = engine.gram_last_kernel()[0] if (49 <= p <= 120 and p % 2 == 0 and n >= 8192) else "logit_kernel"
    kinds[name.split("<")[0]] = kinds.get(name.split("<")[0], 0) + 1
    eg = float((gg - g_ref).abs().max()) / gs
    el = abs(float(ll) - ll_ref) / ll_scale
    worst["g"] = max(worst["g"], eg); worst["ll"] = max(worst["ll"], el)
    assert eg < 1e-12 and el < 1e-12, ("g/ll", c, n, p, ld, off, eg, el, name)
    if w is not None:
        ew = float(((w - w_ref).abs() / w_ref.clamp_min(1e-300)).max())
        worst["w"] = max(worst["w"], ew)
        assert ew < 1e-11, ("w", c, n, p, ld, off, ew, name)
print("PASS FUZZ ok: %d cases, worst errors %s, kernels %s" % (cases, {k: "%.2e" % v for k, v in worst.items()}, kinds))
