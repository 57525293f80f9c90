"""fp32 wide Gram timing: python bench/gram_time_f32.py rows p [reps] [lib.so]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dlsa_amd import _lib
if len(sys.argv) > 4:
    _lib.LIB_PATH = os.path.abspath(sys.argv[4])
from dlsa_amd import engine
rows, p = int(float(sys.argv[1])), int(sys.argv[2]); reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
X, _ = engine.synth(20260101, 0, rows, p, kind=engine.SYNTH_GAUSSIAN, labels=False, dtype=torch.float32)
H = torch.empty(p, p, dtype=torch.float32, device="cuda")
engine.gram(X, None, out=H); torch.cuda.synchronize()
ref = None
if rows <= 200000:
    ref = X.double().T @ X.double()
    print("relerr %.2e" % float((H.double() - ref).abs().max() / ref.abs().max()))
ts = []
for _ in range(reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); engine.gram(X, None, out=H); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
ms = sorted(ts)[len(ts) // 2]
print("GRAMF32 p=%d rows=%.1e  median %.3f ms  %.2f TF  %s [%s]" % (p, rows, ms, rows * p * (p + 1) / ms * 1e-9, engine.gram_last_kernel()[0], os.path.basename(_lib.LIB_PATH)), flush=True)
