"""Experiment: end-time skew of the four workgroups of each slab group in the cyclic Gram kernel (needs the -DDLSA_CYC_TIMESTAMPS build)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dlsa_amd import engine
rows, p = int(sys.argv[1]), int(sys.argv[2])
X, _ = engine.synth(1, 0, rows, p, kind=engine.SYNTH_GAUSSIAN, labels=False)
w = torch.rand(rows, dtype=torch.float64, device="cuda") * 0.25
for rep in range(2):
    engine.gram(X, w); torch.cuda.synchronize()
ws = list(engine._ws_cache.values())[0].view(torch.float64)
PP = 512
t = torch.stack([ws[(s + 1) * PP * PP - 1 - m] for s in range(64) for m in range(4)]).view(64, 4).cpu()
t = t - t.min()
print("p=%d end-time (100 MHz ticks) per group: spread within group  min %.0f  median %.0f  max %.0f ; overall span %.0f ticks = %.3f ms" % (
    p, (t.max(1).values - t.min(1).values).min(), (t.max(1).values - t.min(1).values).median(), (t.max(1).values - t.min(1).values).max(), t.max(), t.max() / 1e5))
print(" first groups:", t[:4].tolist())
