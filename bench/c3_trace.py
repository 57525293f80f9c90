import os, sys, time
sys.path.insert(0, "/root/repo")
import torch
from dlsa_amd import engine
n, p, K = 25_000_000, 500, 25
X, y = engine.synth(20260101, 0, n, p, kind=engine.SYNTH_GAUSSIAN)
offs = [i * (n // K) for i in range(K + 1)]
r = engine.irls_fit(X, y, offs); torch.cuda.synchronize()
t = time.perf_counter(); r = engine.irls_fit(X, y, offs); torch.cuda.synchronize(); print("fit %.4f s" % (time.perf_counter() - t), list(r["n_iter"]), file=sys.stderr)
os.environ["DLSA_IRLS_TRACE"] = "1"
r = engine.irls_fit(X, y, offs); torch.cuda.synchronize()
