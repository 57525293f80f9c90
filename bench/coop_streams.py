"""Does a cooperative launch cost later streams their overlap?  (Why dlsa_kernel_options.cooperative is opt-in.)
Fits a chained problem (four partition chains, each on its own stream created at the first such fit) in a FRESH process
  a) without any cooperative launch before it;
  b) after one cooperative launch (a LARS path at p = 300 with cooperative = 1).
usage: coop_streams.py a|b"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dlsa_amd import engine
mode = sys.argv[1] if len(sys.argv) > 1 else "a"
if mode == "b":
    rng = np.random.default_rng(3)
    p = 300; n = 40 * p
    X = rng.random((n, p)) - 0.5
    S = torch.from_numpy(X.T @ ((rng.random(n) * 0.25)[:, None] * X)).cuda()
    b = torch.from_numpy(np.where(np.arange(p) < 0.4 * p, 1.0, 0.0) + 0.05 * rng.standard_normal(p)).cuda()
    with engine.kernel_options(cooperative=True):
        t = time.perf_counter(); r = engine.lars_path(S, b, False, float(n)); torch.cuda.synchronize()
    print("cooperative LARS path first: %.2f ms, %d steps" % ((time.perf_counter() - t) * 1e3, r["beta"].shape[0] - 1), flush=True)
K, nk, p = 16, 250000, 260
X, y = engine.synth(20260101, 0, K * nk, p, kind=engine.SYNTH_GAUSSIAN)
offs = [i * nk for i in range(K + 1)]
for chains in (4, 1):
    with engine.irls_options(chains=chains, own_hessian=False):
        engine.irls_fit(X, y, offs); torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            t = time.perf_counter(); engine.irls_fit(X, y, offs); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
    print("mode %s: %d partitions of %d x %d on %d chain(s): %.2f ms" % (mode, K, nk, p, chains, min(ts) * 1e3), flush=True)
