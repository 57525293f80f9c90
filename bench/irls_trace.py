"""Iteration trace (DLSA_IRLS_TRACE) and wall time of one map fit: python bench/irls_trace.py n p [K]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dlsa_amd import engine
n, p = int(float(sys.argv[1])), int(sys.argv[2]); K = int(sys.argv[3]) if len(sys.argv) > 3 else 1
X, y = engine.synth(20260101, 0, n, p, kind=engine.SYNTH_GAUSSIAN)
offs = [int(n * k / K) for k in range(K + 1)]
engine.irls_fit(X, y, offs); torch.cuda.synchronize()
for rep in range(3):
    t = time.perf_counter(); r = engine.irls_fit(X, y, offs); torch.cuda.synchronize()
    print("fit %.4f s  iters %s status %s" % (time.perf_counter() - t, r["n_iter"][:4], r["status"][:4]), flush=True)
os.environ["DLSA_IRLS_TRACE"] = "1"
engine.irls_fit(X, y, offs)
