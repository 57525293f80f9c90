"""LARS path phase timer: python bench/lars_prof.py lib.so p...  (lib built with -DDLSA_LARS_PROF prints the per-phase microseconds)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dlsa_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
import torch
from dlsa_amd import engine
for p in [int(v) for v in sys.argv[2:]]:
    g = torch.Generator(device="cuda").manual_seed(p)
    A = torch.randn((6 * p, p), dtype=torch.float64, device="cuda", generator=g)
    S = A.T @ (A * torch.rand(6 * p, 1, dtype=torch.float64, device="cuda", generator=g) * 0.25)
    b = torch.randn(p, dtype=torch.float64, device="cuda", generator=g)
    engine.lars_path(S, b, False, 6.0 * p); torch.cuda.synchronize()
    t = time.perf_counter(); r = engine.lars_path(S, b, False, 6.0 * p); torch.cuda.synchronize()
    print("p=%d lars_path %.3f ms, %d steps" % (p, (time.perf_counter() - t) * 1e3, r["beta"].shape[0] - 1), flush=True)
