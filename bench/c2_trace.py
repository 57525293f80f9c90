"""Config 2 (1e7 x 100, one partition): the fit's wall time and its iteration trace (DLSA_IRLS_TRACE)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dlsa_amd import engine
n, p = 10_000_000, 100
X, y = engine.synth(20260101, 0, n, p, kind=engine.SYNTH_GAUSSIAN)
offs = [0, n]
for _ in range(3):
    r = engine.irls_fit(X, y, offs); torch.cuda.synchronize()
ts = []
for _ in range(7):
    t = time.perf_counter(); r = engine.irls_fit(X, y, offs); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
print("fit %.3f ms (min %.3f)" % (sorted(ts)[3] * 1e3, min(ts) * 1e3), list(r["n_iter"]), file=sys.stderr)
os.environ["DLSA_IRLS_TRACE"] = "1"
r = engine.irls_fit(X, y, offs); torch.cuda.synchronize()
