"""One IRLS fit (for rocprofv3 --stats): python bench/irls_one.py rows p K"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dlsa_amd import engine
rows, p, K = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
X, y = engine.synth(20260101, 0, rows, p, kind=engine.SYNTH_GAUSSIAN)
offs = [int(rows * k / K) for k in range(K + 1)]
engine.irls_fit(X, y, offs); torch.cuda.synchronize()
t = time.perf_counter(); r = engine.irls_fit(X, y, offs); torch.cuda.synchronize(); dt = time.perf_counter() - t
print("irls_fit rows=%d p=%d K=%d: %.4f s iters %s" % (rows, p, K, dt, r["n_iter"][:4]))
