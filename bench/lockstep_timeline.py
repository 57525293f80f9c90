"""Per-workgroup timeline of the LAST fused pass of a lock-step fit (a -DFP_TIMELINE=1 build of irls_pass.hip and irls_batch.hip:
bench/build_lockstep_tl.sh): start / prologue end / loop end / end of every workgroup in s_memrealtime ticks (100 MHz).
   DLSA_AB_LIB=build/var/libdlsa_lstl.so python bench/lockstep_timeline.py [K nk p]"""
import os, sys, struct
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from dlsa_amd import _lib
if os.environ.get("DLSA_AB_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["DLSA_AB_LIB"])
import dlsa_amd
from dlsa_amd import engine

K, nk, p = (int(float(v)) for v in (sys.argv[1:4] if len(sys.argv) >= 4 else (1000, 20000, 100)))
X, y = engine.synth(20260101, 0, K * nk, p, kind=engine.SYNTH_GAUSSIAN)
offs = [k * nk for k in range(K + 1)]
dump = "/tmp/lockstep_tl.bin"
os.environ["DLSA_TL_DUMP"] = dump
for _ in range(3):
    mb = dlsa_amd.fit_logistic_partitions(X, y, part_offsets=offs, batched=True, small=False)
torch.cuda.synchronize()
raw = open(dump, "rb").read()
nslab, GP, pp, _ = struct.unpack("4i", raw[:16])
gp = np.frombuffer(raw[16:], dtype=np.int64).reshape(nslab, GP)
nt, g = pp // 16, (pp % 16 + 3) // 4
if g == 4:
    nt, g = nt + 1, 0
ntc = nt + (1 if g else 0)
tl = gp[:, 16 * ntc + 1: 16 * ntc + 8]
t0 = tl[:, 0].min()
us = (tl - t0) / 100.0
start, pro, loop, end = us[:, 0], us[:, 1], us[:, 2], us[:, 3]
wl = np.concatenate([us[:, 2:3], us[:, 4:7]], axis=1)
print("workgroups %d   kernel span %.1f us" % (nslab, end.max()))
print("prologue   mean %.1f us  max %.1f" % ((pro - start).mean(), (pro - start).max()))
print("loop (w0)  mean %.1f us  min %.1f  max %.1f" % ((loop - pro).mean(), (loop - pro).min(), (loop - pro).max()))
print("wave skew at the loop's end  mean %.1f us  max %.1f" % ((wl.max(1) - wl.min(1)).mean(), (wl.max(1) - wl.min(1)).max()))
print("epilogue   mean %.1f us  max %.1f   (from the LAST wave's loop end: mean %.1f)" % ((end - loop).mean(), (end - loop).max(), (end - wl.max(1)).mean()))
order = np.argsort(start)
rounds = [order[i:i + 256] for i in range(0, nslab, 256)]
for i, r in enumerate(rounds):
    print("round %d: %4d workgroups  start %.1f .. %.1f  end %.1f .. %.1f   loop mean %.1f" % (i, len(r), start[r].min(), start[r].max(), end[r].min(), end[r].max(), (loop - pro)[r].mean()))
busy = (end - start).sum() / (256 * end.max())
print("CU busy fraction (sum of workgroup spans / 256 x kernel span): %.3f;  loops only: %.3f" % (busy, (wl.max(1) - pro).sum() / (256 * end.max())))
