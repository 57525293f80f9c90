"""Lock step: full-row Newton iterations and wall time against the number of gradient-only passes (DLSA_IRLS_GRAD_PASSES).
   python bench/grad_passes_ab.py [K nk p]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import dlsa_amd
from dlsa_amd import engine
K, nk, p = (int(float(v)) for v in (sys.argv[1:4] if len(sys.argv) >= 4 else (1000, 20000, 100)))
X, y = engine.synth(20260101, 0, K * nk, p, kind=engine.SYNTH_GAUSSIAN)
offs = [k * nk for k in range(K + 1)]
for gp in (0, 1, 2, 3, 4):
    os.environ["DLSA_IRLS_GRAD_PASSES"] = str(gp)
    ts = []
    for _ in range(4):
        torch.cuda.synchronize(); t = time.perf_counter()
        mb = dlsa_amd.fit_logistic_partitions(X, y, part_offsets=offs, batched=True, small=False)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
    it = np.asarray(mb.n_iter)
    print("K %d nk %d p %d  grad passes %d: %.1f ms  Newton iterations min %d max %d mean %.2f" % (K, nk, p, gp, sorted(ts)[1] * 1e3, it.min(), it.max(), it.mean()), flush=True)
