"""Quick timing of the fused logit pass (HIP events)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dlsa_amd import engine
shapes = [(20_000_000, 50), (10_000_000, 100), (10_000_000, 128), (10_000_000, 250), (25_000_000, 500), (5_000_000, 1000), (5_000_000, 2000)]
if len(sys.argv) > 2:
    shapes = [(int(float(sys.argv[1])), int(sys.argv[2]))]          # one shape: logit_quick.py rows p
for (rows, p) in shapes:
    X, y = engine.synth(1, 0, rows, p, kind=engine.SYNTH_GAUSSIAN)
    beta = torch.zeros(p, dtype=torch.float64, device="cuda"); beta[: int(0.4 * p)] = 1.0
    engine.logit_pass(X, y, beta); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); engine.logit_pass(X, y, beta); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    ms = sorted(ts)[2]
    print("logit rows=%d p=%d: %.3f ms  %.3g rows/s  %.2f TB/s" % (rows, p, ms, rows / ms * 1e3, rows * 8 * (p + 2) / ms * 1e-9))
    del X, y
