"""Driver for profiling the fused Newton pass and the ring logit pass: python bench/fused_driver.py rows p [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dlsa_amd import engine
rows, p = int(float(sys.argv[1])), int(sys.argv[2]); reps = int(sys.argv[3]) if len(sys.argv) > 3 else 4
X, y = engine.synth(20260101, 0, rows, p, kind=engine.SYNTH_GAUSSIAN)
beta = torch.zeros(p, dtype=torch.float64, device="cuda"); beta[: int(0.4 * p)] = 1.0
for _ in range(reps):
    engine.irls_pass(X, y, beta)
    engine.logit_pass(X, y, beta)
torch.cuda.synchronize()
print("done", engine.gram_last_kernel())
