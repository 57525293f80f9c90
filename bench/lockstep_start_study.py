"""Study (torch fp64, not the product path): Newton step sizes of lock-step partitions from different starts -- what would cut a full-row pass?
   python bench/lockstep_start_study.py [K nk p]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dlsa_amd import engine
K, nk, p = (int(float(v)) for v in (sys.argv[1:4] if len(sys.argv) >= 4 else (1000, 20000, 100)))
X, y = engine.synth(20260101, 0, K * nk, p, kind=engine.SYNTH_GAUSSIAN)

def newton(Xk, yk, b, tol=1e-10, it=12, Hfix=None):
    seq = []
    for _ in range(it):
        eta = Xk @ b
        mu = torch.sigmoid(eta)
        g = Xk.T @ (yk - mu)
        H = (Xk * (mu * (1 - mu))[:, None]).T @ Xk if Hfix is None else Hfix
        d = torch.linalg.solve(H, g)
        r = float(d.abs().max() / max(1.0, float(b.abs().max())))
        seq.append(r)
        if r <= tol: break
        b = b + d
    return b, seq

each = max(256, ((max(8 * nk, 1000 * p) + K - 1) // K + 31) // 32 * 32)
idx = torch.cat([torch.arange(k * nk, k * nk + each, device=X.device) for k in range(K)])
bp, seq = newton(X[idx], y[idx], torch.zeros(p, dtype=torch.float64, device=X.device), tol=3e-2)
print("pooled fit on %d rows: steps %s" % (idx.numel(), ["%.1e" % v for v in seq]))
eta = X[idx] @ bp; mu = torch.sigmoid(eta); Hp = (X[idx] * (mu * (1 - mu))[:, None]).T @ X[idx] * (nk / idx.numel())
import collections
cnt = collections.Counter(); cnt2 = collections.Counter(); cnt3 = collections.Counter()
for k in range(0, K, max(1, K // 40)):
    Xk, yk = X[k * nk:(k + 1) * nk], y[k * nk:(k + 1) * nk]
    _, s1 = newton(Xk, yk, bp.clone())
    # a gradient-only first step with the pooled Hessian, then Newton
    g = Xk.T @ (yk - torch.sigmoid(Xk @ bp)); b1 = bp + torch.linalg.solve(Hp, g)
    _, s2 = newton(Xk, yk, b1)
    # two gradient-only steps
    g = Xk.T @ (yk - torch.sigmoid(Xk @ b1)); b2 = b1 + torch.linalg.solve(Hp, g)
    _, s3 = newton(Xk, yk, b2)
    cnt[len(s1)] += 1; cnt2[len(s2)] += 1; cnt3[len(s3)] += 1
    if k < 3 * max(1, K // 40):
        print("partition %d: pooled start %s | + 1 pooled-H step %s | + 2 %s" % (k, ["%.1e" % v for v in s1], ["%.1e" % v for v in s2], ["%.1e" % v for v in s3]))
print("full passes needed from the pooled start: %s;  after one gradient-only step: %s;  after two: %s" % (dict(cnt), dict(cnt2), dict(cnt3)))
