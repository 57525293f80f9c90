"""Quick numeric check of engine.gram against an fp64 matmul for a list of widths: python bench/gram_check.py rows p p p ..."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dlsa_amd import engine
rows = int(sys.argv[1])
for p in map(int, sys.argv[2:]):
    g = torch.Generator(device="cuda"); g.manual_seed(p)
    X = torch.randn((rows, p), dtype=torch.float64, device="cuda", generator=g)
    w = torch.rand(rows, dtype=torch.float64, device="cuda", generator=g)
    for wt in (w, None):
        H = engine.gram(X, wt)
        ref = X.T @ (X if wt is None else X * wt[:, None])
        print("CHECK p=%d rows=%d w=%s: rel err %.3e  H00 %.6g ref00 %.6g" % (p, rows, wt is not None, float((H - ref).abs().max() / ref.abs().max()), float(H[0, 0]), float(ref[0, 0])))
