import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dlsa_amd import engine
X, y = engine.synth(20260101, 0, 4000000, 500, kind=engine.SYNTH_GAUSSIAN)
offs = [0, 2000000, 4000000]
a = engine.irls_fit(X, y, offs)
os.environ["DLSA_IRLS_PREDICT"] = "0"
b = engine.irls_fit(X, y, offs)
rc = float((a["coef"] - b["coef"]).abs().max() / b["coef"].abs().max())
rh = float((a["Sig_inv"] - b["Sig_inv"]).abs().max() / b["Sig_inv"].abs().max())
print("predict vs confirm: coef rel %.2e  Sig_inv rel %.2e  iters %s vs %s" % (rc, rh, a["n_iter"], b["n_iter"]))
