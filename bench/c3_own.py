"""Config 3's per-GPU shard (25 partitions of 1e6 x 500) fitted with and without the partition's own reduced-precision Hessian
(dlsa_irls_options.own_hessian): wall time, iterations, and the two results against each other.  usage: c3_own.py [K] [rows per partition] [p]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dlsa_amd import _lib
if os.environ.get("DLSA_AB_LIB"):          # same-box A/B of a build variant
    _lib.LIB_PATH = os.path.abspath(os.environ["DLSA_AB_LIB"])
from dlsa_amd import engine
K = int(sys.argv[1]) if len(sys.argv) > 1 else 25
nk = int(float(sys.argv[2])) if len(sys.argv) > 2 else 1_000_000
p = int(sys.argv[3]) if len(sys.argv) > 3 else 500
n = K * nk
X, y = engine.synth(20260101, 0, n, p, kind=engine.SYNTH_GAUSSIAN)
offs = [i * nk for i in range(K + 1)]
res = {}
for own in (False, True):
    with engine.irls_options(own_hessian=own):
        r = engine.irls_fit(X, y, offs); torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            t = time.perf_counter(); r = engine.irls_fit(X, y, offs); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
    res[own] = r
    print("own_hessian=%d  fit %.4f s (min of 3: %s)  n_iter %s  status %s" % (own, min(ts), ["%.4f" % v for v in ts], list(r["n_iter"]), set(int(v) for v in r["status"])), flush=True)
a, b = res[False], res[True]
rel = lambda u, v: float((u - v).abs().max() / v.abs().max())
print("coef %.2e  Sig_inv %.2e  Sig_invMcoef %.2e  (own vs not, relative l-inf)" % (rel(b["coef"], a["coef"]), rel(b["Sig_inv"], a["Sig_inv"]), rel(b["Sig_invMcoef"], a["Sig_invMcoef"])), flush=True)
if os.environ.get("C3_TRACE"):
    with engine.irls_options(own_hessian=True, trace=True, chains=1):
        engine.irls_fit(X[: 3 * nk], y[: 3 * nk], offs[:4]); torch.cuda.synchronize()
