"""Config 3's shard as ONE partition (2.5e7 x 500): the driver's trace and the wall time.  python bench/c3_k1_trace.py [n] [p]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dlsa_amd import engine
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 25_000_000
p = int(sys.argv[2]) if len(sys.argv) > 2 else 500
X, y = engine.synth(20260101, 0, n, p, kind=engine.SYNTH_GAUSSIAN)
r = engine.irls_fit(X, y, [0, n]); torch.cuda.synchronize()
ts = []
for _ in range(3):
    t = time.perf_counter(); r = engine.irls_fit(X, y, [0, n]); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
print("fit %.4f s (%s)  n_iter %s" % (min(ts), ["%.4f" % v for v in ts], r["n_iter"]))
with engine.irls_options(trace=True):
    engine.irls_fit(X, y, [0, n]); torch.cuda.synchronize()
