"""Per-workgroup timeline of the fused Newton pass (a -DFP_TIMELINE=1 build: bench/build_variant.sh tl irls_pass.hip -DFP_TIMELINE=1):
start / prologue end / loop end / kernel end of every workgroup in s_memrealtime ticks (100 MHz), read back from the free slots of
the g partials.   DLSA_AB_LIB=build/var/libdlsa_tl.so python bench/fused_timeline.py [n] [p]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from dlsa_amd import _lib
if os.environ.get("DLSA_AB_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["DLSA_AB_LIB"])
from dlsa_amd import engine

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
p = int(sys.argv[2]) if len(sys.argv) > 2 else 100
X, y = engine.synth(20260101, 0, n, p, kind=engine.SYNTH_GAUSSIAN)
beta = torch.zeros(p, dtype=torch.float64, device="cuda"); beta[: int(0.4 * p)] = 1.0
for _ in range(3):
    engine.irls_pass(X, y, beta)
torch.cuda.synchronize()
lib = _lib.load()
ws = engine._workspace(lib.dlsa_irls_pass_workspace_bytes(n, p), X.device)
nt, g = p // 16, (p % 16 + 3) // 4
if g == 4:
    nt, g = nt + 1, 0
ntc = nt + (1 if g else 0)
GP, PP = 16 * ntc + 8, ((p + 15) // 16 * 16 + 63) // 64 * 64
nslab = min(256, max(1, n // 2048))
part = (nslab * PP * PP * 8 + 255) // 256 * 256
gp = ws[part: part + nslab * GP * 8].view(torch.int64).view(nslab, GP).cpu().numpy()
tl = gp[:, 16 * ntc + 1: 16 * ntc + 8].astype(np.int64)
t0 = tl[:, 0].min()
us = (tl - t0) / 100.0
start, pro, loop, end = us[:, 0], us[:, 1], us[:, 2], us[:, 3]
wl = np.concatenate([us[:, 2:3], us[:, 4:7]], axis=1)         # the four waves' loop ends
print("workgroups %d   kernel span %.1f us" % (nslab, end.max()))
print("start      min %.1f  max %.1f" % (start.min(), start.max()))
print("prologue   mean %.1f us  max %.1f" % ((pro - start).mean(), (pro - start).max()))
print("loop (w0)  mean %.1f us  min %.1f  max %.1f" % ((loop - pro).mean(), (loop - pro).min(), (loop - pro).max()))
print("wave skew at the loop's end (max - min over the four waves)  mean %.1f us  max %.1f" % ((wl.max(1) - wl.min(1)).mean(), (wl.max(1) - wl.min(1)).max()))
print("epilogue   mean %.1f us  max %.1f   (from the LAST wave's loop end: mean %.1f)" % ((end - loop).mean(), (end - loop).max(), (end - wl.max(1)).mean()))
print("end        min %.1f  mean %.1f  max %.1f" % (end.min(), end.mean(), end.max()))
q = np.percentile(end, [5, 25, 50, 75, 95])
print("end percentiles 5/25/50/75/95: " + " ".join("%.1f" % v for v in q))
xcd = np.arange(nslab) % 8
print("mean end per XCD (workgroup id mod 8): " + " ".join("%.1f" % end[xcd == k].mean() for k in range(8)))
