"""Config 4's structured fit (raw numerics + level codes, 14 partitions of 1e6 rows, p = 260): wall time, iteration trace, and (under
rocprofv3 --kernel-trace --stats) the kernels it is made of."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dlsa_amd import engine
from surrogates import airline_shaped as make
n, K = 14_000_000, 14
c = make(n, 7, dense=False)
num, codes, plan, y = c["num"], c["codes"], c["plan"], c["y"]
offs = [int(n * k / K) for k in range(K + 1)]
for _ in range(3):
    r = engine.onehot_irls_fit(plan, num, codes, y, offs); torch.cuda.synchronize()
ts = []
for _ in range(7):
    t = time.perf_counter(); r = engine.onehot_irls_fit(plan, num, codes, y, offs); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
print("fit %.3f ms (min %.3f)" % (sorted(ts)[3] * 1e3, min(ts) * 1e3), list(r["n_iter"]), file=sys.stderr)
if os.environ.get("C4_TRACE"):
    os.environ["DLSA_IRLS_TRACE"] = "1"
    os.environ["DLSA_IRLS_CHAINS"] = "1"
    r = engine.onehot_irls_fit(plan, num, codes, y, offs); torch.cuda.synchronize()
