"""PCIe-inclusive rate of the frame-level operator boundary: host fp64 rows -> HBM (pageable and pinned)."""
import time, torch, numpy as np
n, p = 2_000_000, 500
x = torch.from_numpy(np.random.default_rng(0).random((n, p)))
for name, src in (("pageable", x), ("pinned", x.pin_memory())):
    src.cuda(); torch.cuda.synchronize()
    t = time.perf_counter(); d = src.cuda(non_blocking=True); torch.cuda.synchronize(); dt = time.perf_counter() - t
    print("H2D %s: %.1f GB/s -> %.3g rows/s at p=%d (Gram kernel: 2.46e8 rows/s)" % (name, n * p * 8 / dt / 1e9, n / dt, p))
