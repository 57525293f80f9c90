"""Quick timing of dlsa_design_f64 (HIP events, preallocated output) on the airline-shaped column plan."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dlsa_amd import engine

def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 14_000_000
    levels = tuple(int(v) for v in sys.argv[2].split(",")) if len(sys.argv) > 2 else (11, 6, 20, 110, 110)
    q = int(sys.argv[3]) if len(sys.argv) > 3 else 7
    num = torch.randn((n, q), dtype=torch.float64, device="cuda")
    codes = torch.stack([torch.randint(0, L, (n,), device="cuda", dtype=torch.int32) for L in levels], 1).contiguous()
    kind, src, level, shift, scale = [0], [0], [0], [0.0], [1.0]
    for j in range(q):
        kind.append(1); src.append(j); level.append(0); shift.append(1.5); scale.append(3.0)
    for fi, L in enumerate(levels):
        for lv in range(1, L):
            kind.append(2); src.append(fi); level.append(lv); shift.append(0.0); scale.append(1.0)
    d = lambda a, t: torch.tensor(a, dtype=t, device="cuda")
    spec = (d(kind, torch.int32), d(src, torch.int32), d(level, torch.int32), d(shift, torch.float64), d(scale, torch.float64))
    p = len(kind)
    X = torch.empty((n, p), dtype=torch.float64, device="cuda")
    engine.design(num, codes, *spec, out=X); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); engine.design(num, codes, *spec, out=X); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ms = sorted(ts)[2]
    print("design n=%d p=%d q=%d f=%d: %.3f ms  write %.0f GB/s  (+read %.0f GB/s)" % (
        n, p, q, len(levels), ms, n * p * 8 / ms * 1e-6, n * (q * 8 + len(levels) * 4) / ms * 1e-6))
    Y = torch.empty_like(X)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    Y.zero_(); torch.cuda.synchronize()
    e0.record(); Y.zero_(); e1.record(); torch.cuda.synchronize()
    print("memset of the same size: %.3f ms  %.0f GB/s" % (e0.elapsed_time(e1), n * p * 8 / e0.elapsed_time(e1) * 1e-6))

main()
