"""Time of one factor + inverse + H^-1 + solve of a p x p SPD system through a one-row IRLS... (engine has no direct entry):
uses irls_fit on a tiny partition count to time the fit's fixed costs.  python bench/chol_quick.py p"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dlsa_amd import _lib
if len(sys.argv) > 2:
    _lib.LIB_PATH = os.path.abspath(sys.argv[2])
from dlsa_amd import engine
_ko = engine.kernel_options(engine.kernel_options_from_env()); _ko.__enter__()      # DLSA_GRAM_DBG etc. from the shell: applied by the host layer (the library reads no environment variable for them)
p = int(sys.argv[1]) if len(sys.argv) > 1 else 100
n = 20000
X, y = engine.synth(1, 0, n, p, kind=engine.SYNTH_GAUSSIAN)
offs = [0, n]
engine.irls_fit(X, y, offs); torch.cuda.synchronize()
ts = []
for _ in range(20):
    t = time.perf_counter(); r = engine.irls_fit(X, y, offs); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
print("p=%d n=%d fit %.3f ms (median of 20), iters %s, CHOL_SMALL=%s" % (p, n, sorted(ts)[10] * 1e3, r["n_iter"], os.environ.get("DLSA_CHOL_SMALL", "1")))
