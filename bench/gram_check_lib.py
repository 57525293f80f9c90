"""Numeric check of a library build's Gram against an fp64 matmul: python bench/gram_check_lib.py lib.so p rows"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dlsa_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
from dlsa_amd import engine
p, rows = int(sys.argv[2]), int(float(sys.argv[3]))
X = torch.randn((rows, p), dtype=torch.float64, device="cuda") * (1.0 + 0.01 * torch.arange(p, dtype=torch.float64, device="cuda"))
w = torch.rand(rows, dtype=torch.float64, device="cuda")
for ww in (w, None):
    H = engine.gram(X, ww)
    ref = X.T @ (X if ww is None else X * ww[:, None])
    print("CHECK p=%d rows=%d %s: relerr %.2e sym %s  %s" % (p, rows, "weighted" if ww is not None else "unweighted",
          float((H - ref).abs().max() / ref.abs().max()), bool(torch.equal(H, H.T)), engine.gram_last_kernel()[0]), flush=True)
