"""Weighted Gram timing with the dispatched kernel and the sustained clock: python bench/gram_time.py rows p [reps] [lib.so]
(lib.so: load that build of the library instead of dlsa_amd/libdlsa_hip.so -- same-box A/B of build variants)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dlsa_amd import _lib
if len(sys.argv) > 4:
    _lib.LIB_PATH = os.path.abspath(sys.argv[4])
from dlsa_amd import engine
rows, p = int(float(sys.argv[1])), int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 7
X, _ = engine.synth(20260101, 0, rows, p, kind=engine.SYNTH_GAUSSIAN, labels=False)
w = torch.rand(rows, dtype=torch.float64, device="cuda") * 0.25
H = torch.empty(p, p, dtype=torch.float64, device="cuda")
for _ in range(2):
    engine.gram(X, w, out=H)
torch.cuda.synchronize()
ts, cl = [], []
for _ in range(reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); engine.gram(X, w, out=H); e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
    name, cyc = engine.gram_last_kernel(want_cycles=True)
    cl.append(cyc / (ts[-1] * 1e-3) / 1e9)
k = sorted(range(reps), key=lambda i: ts[i])[reps // 2]
print("GRAM p=%d rows=%.1e  median %.3f ms  min %.3f  %.2f TF  %.3f GHz  %s  [%s]" % (
    p, rows, ts[k], min(ts), rows * (p * (p + 1) + p) / ts[k] * 1e-9, cl[k], name, os.path.basename(_lib.LIB_PATH)), flush=True)
