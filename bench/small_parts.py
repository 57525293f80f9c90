"""Config 1 (20 partitions of 5 000 x 50, the reference's simulated_pdf defaults): map step wall time, one launch vs host-driven."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dlsa_amd
from dlsa_amd import engine
K, nk, p = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (20, 5000, 50)))
X, y = engine.synth(20260101, 0, K * nk, p)
offs = [k * nk for k in range(K + 1)]
for mode in ("1", "0"):
    os.environ["DLSA_IRLS_SMALL"] = mode
    for icpt in (False, True):
        ts = []
        for _ in range(5):
            torch.cuda.synchronize(); t = time.perf_counter()
            mb = dlsa_amd.fit_logistic_partitions(X, y, part_offsets=offs, fit_intercept=icpt)
            torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
        print("SMALL=%s K=%d n_k=%d p=%d intercept=%s: map step %.3f ms (min %.3f)  iters %s" % (mode, K, nk, p, icpt, sorted(ts)[2] * 1e3, min(ts) * 1e3, mb.n_iter[:3]))
