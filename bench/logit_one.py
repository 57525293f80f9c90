"""One shape of the logit pass, a few launches (for rocprofv3 PMC passes): python bench/logit_one.py rows p"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dlsa_amd import engine
rows, p = int(sys.argv[1]), int(sys.argv[2])
X, y = engine.synth(1, 0, rows, p, kind=engine.SYNTH_GAUSSIAN)
beta = torch.zeros(p, dtype=torch.float64, device="cuda"); beta[: int(0.4 * p)] = 1.0
for _ in range(3):
    engine.logit_pass(X, y, beta)
torch.cuda.synchronize()
