"""Gram timing at a given row pitch: bench/gram_pitch.py rows p ld reps  (X = the first p columns of an [rows, ld] matrix)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dlsa_amd import engine
_ko = engine.kernel_options(engine.kernel_options_from_env()); _ko.__enter__()      # DLSA_GRAM_DBG etc. from the shell: applied by the host layer (the library reads no environment variable for them)

rows, p, ld, reps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
Xf, _ = engine.synth(1, 0, rows, ld, kind=engine.SYNTH_GAUSSIAN, labels=False)
X = Xf[:, :p]
w = torch.rand(rows, dtype=torch.float64, device="cuda") * 0.25
H = torch.empty(p, p, dtype=torch.float64, device="cuda")
engine.gram(X, w, out=H); torch.cuda.synchronize()
ts = []
for _ in range(reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); engine.gram(X, w, out=H); e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
ms = sorted(ts)[len(ts) // 2]
print("PITCH dbg=%s rows=%d p=%d ld=%d: median %.3f ms  %.4g rows/s  %.2f TF(alg)" % (
    os.environ.get("DLSA_GRAM_DBG", "0"), rows, p, ld, ms, rows / ms * 1e3, rows * (p * (p + 1) + p) / ms * 1e-9))
