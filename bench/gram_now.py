"""Gram timing with and without weights: python bench/gram_now.py rows p [f32]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dlsa_amd import engine
rows, p = int(sys.argv[1]), int(sys.argv[2])
dt = torch.float32 if len(sys.argv) > 3 and sys.argv[3] == "f32" else torch.float64
X, _ = engine.synth(1, 0, rows, p, kind=engine.SYNTH_GAUSSIAN, labels=False, dtype=dt)
w = torch.rand(rows, dtype=dt, device="cuda") * 0.25
H = torch.empty(p, p, dtype=dt, device="cuda")
for name, ww in (("weighted", w), ("unweighted", None)):
    engine.gram(X, ww, out=H); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); engine.gram(X, ww, out=H); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    ms = sorted(ts)[2]
    print("gram %s rows=%d p=%d %s: %.3f ms  %.2f TF(alg)" % (name, rows, p, str(dt)[6:], ms, rows * (p * (p + 1) + p) / ms * 1e-9))
