"""One structured (raw numerics + codes) IRLS fit on the airline-shaped design, for rocprofv3 --stats:
python bench/onehot_one.py rows K"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dlsa_amd import engine
n, K = int(sys.argv[1]), int(sys.argv[2])
g = torch.Generator(device="cuda"); g.manual_seed(7)
num = torch.randn((n, 7), dtype=torch.float64, device="cuda", generator=g) * 3.0 + 1.5
levels = (11, 6, 20, 110, 110)
codes = torch.empty((n, 5), dtype=torch.int32, device="cuda")
for fi, L in enumerate(levels):
    pr = 1.0 / torch.arange(1, L + 1, dtype=torch.float64, device="cuda")
    codes[:, fi] = torch.multinomial(pr / pr.sum(), n, replacement=True, generator=g).int()
p = 8 + sum(L - 1 for L in levels)
level_col, pos = [], 8
for L in levels:
    level_col += [-1] + list(range(pos, pos + L - 1)); pos += L - 1
plan = engine.OnehotPlan(p, [0] + [1] * 7, [0] + list(range(7)), [0.0] + [1.5] * 7, [1.0] + [3.0] * 7, list(range(8)), list(levels), level_col)
beta = torch.randn(p, dtype=torch.float64, device="cuda", generator=g) * 0.15
w, _, _ = engine.onehot_logit_pass(plan, num, codes, torch.zeros(n, dtype=torch.float64, device="cuda"), beta, want_g=False, want_loglik=False)
eta = torch.log(1.0 / (0.5 - torch.sqrt((0.25 - w).clamp_min(0))) - 1.0).abs()   # |eta| from w = mu(1-mu); sign irrelevant for a benchmark
y = (torch.rand(n, dtype=torch.float64, device="cuda", generator=g) < torch.sigmoid(eta)).double()
offs = [int(n * k / K) for k in range(K + 1)]
engine.onehot_irls_fit(plan, num, codes, y, offs); torch.cuda.synchronize()
t = time.perf_counter(); r = engine.onehot_irls_fit(plan, num, codes, y, offs); torch.cuda.synchronize(); dt = time.perf_counter() - t
print("onehot_irls_fit rows=%d p=%d K=%d: %.4f s iters %s status %s" % (n, p, K, dt, r["n_iter"][:4], set(r["status"])))
