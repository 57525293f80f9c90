"""Fused Newton pass (dlsa_irls_pass_f64) against the logit pass + the Gram pass it replaces, per width."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dlsa_amd import _lib
if os.environ.get("DLSA_AB_LIB"):          # same-box A/B of a build variant (bench/build_variant.sh)
    _lib.LIB_PATH = os.path.abspath(os.environ["DLSA_AB_LIB"])
from dlsa_amd import engine

def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
for p in [int(v) for v in (sys.argv[2:] or ["50", "64", "80", "100", "112"])]:
    X, y = engine.synth(20260101, 0, n, p, kind=engine.SYNTH_GAUSSIAN)
    beta = torch.zeros(p, dtype=torch.float64, device="cuda"); beta[: int(0.4 * p)] = 1.0
    w, _, _ = engine.logit_pass(X, y, beta)
    tl = t(lambda: engine.logit_pass(X, y, beta))
    if p > 120:                                     # logit pass only (the fused pass covers 49..120)
        print("p=%4d n=%.0e  logit %.3f ms  %.2f TB/s  %s" % (p, n, tl, n * p * 8 / tl * 1e-9, os.path.basename(_lib.LIB_PATH)), flush=True)
        del X, y, w
        continue
    tg = t(lambda: engine.gram(X, w))
    tf = t(lambda: engine.irls_pass(X, y, beta))
    name, cyc = engine.gram_last_kernel(want_cycles=True)
    tf1 = t(lambda: engine.irls_pass(X, y, beta), reps=1)
    name, cyc = engine.gram_last_kernel(want_cycles=True)
    print("p=%4d n=%.0e  logit %.3f ms  gram %.3f ms  (sum %.3f)  fused %.3f ms (%.2fx of the pair; %.2f GHz)  %s" % (
        p, n, tl, tg, tl + tg, tf, tf / (tl + tg), cyc / (tf1 * 1e-3) / 1e9, name + " " + os.path.basename(_lib.LIB_PATH)), flush=True)
    del X, y, w
