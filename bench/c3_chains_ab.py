"""Config 3's shard (25 x 1e6 x 500): the map fit against the number of partition chains.  python bench/c3_chains_ab.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dlsa_amd import engine
K, nk, p = 25, 1_000_000, 500
X, y = engine.synth(20260101, 0, K * nk, p, kind=engine.SYNTH_GAUSSIAN)
offs = [i * nk for i in range(K + 1)]
for ch in (None, 3, 4, 5, 6, 8):
    with engine.irls_options(chains=ch):
        engine.irls_fit(X, y, offs); torch.cuda.synchronize()
        ts = []
        for _ in range(4):
            t = time.perf_counter(); r = engine.irls_fit(X, y, offs); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
    print("chains %s: %.4f s (%s)" % (ch, min(ts), ["%.4f" % v for v in ts]), flush=True)
