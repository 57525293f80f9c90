"""LARS path kernels side by side: python bench/lars_ab.py p...   (DLSA_LARS_Q=0: lars.hip's R^{-1} form; default: lars_q.hip for m <= 400)
Prints min / median wall time of lars_path (host call incl. the step-count read-back) for 'lar' and 'lasso'."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dlsa_amd import _lib
if os.environ.get("DLSA_LIB"): _lib.LIB_PATH = os.path.abspath(os.environ["DLSA_LIB"])
from dlsa_amd import engine

for p in [int(v) for v in sys.argv[1:]] or [50, 100, 260]:
    rng = np.random.default_rng(p)
    n = 40 * p
    X = rng.random((n, p)) - 0.5
    S = torch.from_numpy(X.T @ ((rng.random(n) * 0.25)[:, None] * X)).cuda()
    b = torch.from_numpy(np.where(np.arange(p) < 0.4 * p, 1.0, 0.0) + 0.05 * rng.standard_normal(p)).cuda()
    out = []
    for q in ("0", "1"):
        with engine.kernel_options(lars_q=int(q)):
            for typ in ("lar", "lasso"):
                engine.lars_path(S, b, False, float(n), type=typ); torch.cuda.synchronize()
                reps = []
                for _ in range(9):
                    t = time.perf_counter(); r = engine.lars_path(S, b, False, float(n), type=typ); torch.cuda.synchronize()
                    reps.append((time.perf_counter() - t) * 1e3)
                reps.sort()
                out.append("Q=%s %s %.3f / %.3f ms (%d steps)" % (q, typ, reps[0], reps[len(reps) // 2], r["beta"].shape[0] - 1))
    print("p=%d: " % p + "  ".join(out), flush=True)
