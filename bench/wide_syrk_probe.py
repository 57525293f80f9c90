"""Six wide Newton passes at 1e6 x 500 (for a kernel trace of the logit-image / syrk / reduce launches).  DLSA_AB_LIB = a variant library."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dlsa_amd import _lib
if os.environ.get("DLSA_AB_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["DLSA_AB_LIB"])
from dlsa_amd import engine
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000
p = int(sys.argv[2]) if len(sys.argv) > 2 else 500
X, y = engine.synth(20260101, 0, n, p, kind=engine.SYNTH_GAUSSIAN)
beta = torch.zeros(p, dtype=torch.float64, device="cuda"); beta[: int(0.4 * p)] = 0.9
for _ in range(6):
    engine.newton_wide_pass(X, y, beta)
torch.cuda.synchronize()
