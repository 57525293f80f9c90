"""Host vs device time of repeated LARS paths at one p (looks for outliers)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dlsa_amd import engine
p = int(sys.argv[1]) if len(sys.argv) > 1 else 500
rng = np.random.default_rng(p)
n = 40 * p
X = rng.random((n, p)) - 0.5
S = torch.from_numpy(X.T @ ((rng.random(n) * 0.25)[:, None] * X)).cuda()
b = torch.from_numpy(np.where(np.arange(p) < 0.4 * p, 1.0, 0.0) + 0.05 * rng.standard_normal(p)).cuda()
engine.lars_path(S, b, False, float(n)); torch.cuda.synchronize()
for rep in range(12):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t = time.perf_counter(); e0.record()
    engine.lars_path(S, b, False, float(n)); e1.record(); torch.cuda.synchronize()
    print("rep %d host %.2f ms device %.2f ms" % (rep, (time.perf_counter() - t) * 1e3, e0.elapsed_time(e1)))
    if rep == 5: time.sleep(0.5)
