"""Ordered (bit-reproducible) against unordered LDS accumulation of the structured one-hot passes: timing and run-to-run bits.
python bench/onehot_order.py rows"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dlsa_amd import engine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 14000000
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
y = (torch.rand(n, dtype=torch.float64, device="cuda", generator=g) < 0.4).double()

def timed(fn, reps=5):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); out = fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return sorted(ts)[len(ts) // 2], out

for mode in ("1", "0", "1", "0"):
  with engine.kernel_options(onehot_ordered=int(mode)):
    tl, (w, gvec, ll) = timed(lambda: engine.onehot_logit_pass(plan, num, codes, y, beta))
    tg, H = timed(lambda: engine.onehot_gram(plan, num, codes, w))
    same_g = all(torch.equal(gvec, engine.onehot_logit_pass(plan, num, codes, y, beta)[1]) for _ in range(4))
    same_H = all(torch.equal(H, engine.onehot_gram(plan, num, codes, w)) for _ in range(4))
    print("ORDERED=%s rows=%d p=%d: logit %.3f ms  gram %.3f ms  g bit-identical over 5 runs: %s  H: %s" % (mode, n, p, tl, tg, same_g, same_H))
