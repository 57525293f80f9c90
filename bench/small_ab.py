"""The one-launch kernel for many small partitions (csrc/irls_small.hip): python bench/small_ab.py [K nk p] ...
DLSA_LIB=<path> picks another build of the library (e.g. -DDLSA_SM_THREADS=512)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dlsa_amd import _lib
if os.environ.get("DLSA_LIB"): _lib.LIB_PATH = os.path.abspath(os.environ["DLSA_LIB"])
import torch
import dlsa_amd
from dlsa_amd import engine

shapes = [tuple(int(float(v)) for v in sys.argv[i:i + 3]) for i in range(1, len(sys.argv) - 2, 3)] or \
    [(20, 5000, 50), (200, 5000, 50), (100, 20000, 64), (20, 50000, 50), (256, 2000, 20)]
for K, nk, p in shapes:
    X, y = engine.synth(20260101, 0, K * nk, p, kind=engine.SYNTH_UNIFORM if hasattr(engine, "SYNTH_UNIFORM") else engine.SYNTH_GAUSSIAN)
    offs = [k * nk for k in range(K + 1)]
    ts = []
    for _ in range(7):
        torch.cuda.synchronize(); t = time.perf_counter()
        mb = dlsa_amd.fit_logistic_partitions(X, y, part_offsets=offs, batched=False)      # DLSA_IRLS_SMALL=2 forces the one-launch kernel, =0 the chains
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
    print("K=%4d n_k=%6d p=%3d: %.3f ms (min %.3f), path %d, iters %s, ok %s" % (K, nk, p, sorted(ts)[3] * 1e3, min(ts) * 1e3, engine.irls_last_fit_path(),
                                                                                mb.n_iter[:3], all(s == 0 for s in mb.status)), flush=True)
