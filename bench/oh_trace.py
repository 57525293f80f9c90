"""Structured (one-hot) map fit of the airline-shaped shard: wall time, and under rocprofv3 the kernel list of ONE fit.
python bench/oh_trace.py [rows] [K]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dlsa_amd import engine
import surrogates
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 14_000_000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 14
d = surrogates.airline_shaped(n, dense=False)
offs = [int(n * k / K) for k in range(K + 1)]
engine.onehot_irls_fit(d["plan"], d["num"], d["codes"], d["y"], offs); torch.cuda.synchronize()
for _ in range(3):
    t = time.perf_counter(); r = engine.onehot_irls_fit(d["plan"], d["num"], d["codes"], d["y"], offs); torch.cuda.synchronize()
    print("structured fit %.4f s  iters %s  status ok %s" % (time.perf_counter() - t, r["n_iter"], all(s == 0 for s in r["status"])), flush=True)
