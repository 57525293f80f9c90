"""Does the bf16 image of the wide Newton pass stay on chip when the pass is taken in ROW BLOCKS (image written by the logit kernel and
read back by the syrk while it is still in the memory-side cache)?  python bench/wide_blocked_probe.py [rows p]
For each block size: HIP-event time of newton_wide_pass over all blocks (one stream, the same workspace reused) and of the plain logit
pass over the same blocks."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dlsa_amd import engine
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 4_000_000
p = int(sys.argv[2]) if len(sys.argv) > 2 else 500
X, y = engine.synth(11, 0, n, p, kind=engine.SYNTH_GAUSSIAN)
beta = torch.zeros(p, dtype=torch.float64, device="cuda"); beta[: p // 3] = 0.5
def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts) // 2]
for B in (n, 1_000_000, 500_000, 250_000, 125_000, 62_500):
    if B > n: continue
    blocks = [(i, min(n, i + B)) for i in range(0, n, B)]
    def wide():
        for lo, hi in blocks: engine.newton_wide_pass(X[lo:hi], y[lo:hi], beta)
    def plain():
        for lo, hi in blocks: engine.logit_pass(X[lo:hi], y[lo:hi], beta, want_w=True)
    tw, tp = timed(wide), timed(plain)
    print("rows %d p %d in blocks of %8d (%3d blocks, image %.0f MB each): wide pass %.3f ms per 1e6 rows, plain logit pass %.3f ms per 1e6 rows, H~ costs %.3f" % (
        n, p, B, len(blocks), B * ((p + 31) // 32) * 64 / 1e6, tw / n * 1e6, tp / n * 1e6, (tw - tp) / n * 1e6), flush=True)
