"""Quick A/B timing of the Gram kernel (HIP events), for kernel tuning on the GPU box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dlsa_amd import engine
_ko = engine.kernel_options(engine.kernel_options_from_env()); _ko.__enter__()      # DLSA_GRAM_DBG etc. from the shell: applied by the host layer (the library reads no environment variable for them)

def main():
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 5_000_000
    p = int(sys.argv[2]) if len(sys.argv) > 2 else 500
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    dt = torch.float64 if (len(sys.argv) <= 4 or sys.argv[4] == "f64") else torch.float32
    X, _ = engine.synth(1, 0, rows, p, kind=engine.SYNTH_GAUSSIAN, labels=False, dtype=dt)
    w = torch.rand(rows, dtype=dt, device="cuda") * 0.25
    H = torch.empty(p, p, dtype=dt, device="cuda")
    engine.gram(X, w, out=H); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); engine.gram(X, w, out=H); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ms = sorted(ts)[len(ts) // 2]
    fl = p * (p + 1) + p
    print("DBG=%s rows=%d p=%d %s: median %.3f ms  min %.3f  %.4g rows/s  %.2f TF(alg)  %.0f GB/s(alg)" % (
        os.environ.get("DLSA_GRAM_DBG", "0"), rows, p, str(dt)[6:], ms, min(ts), rows / ms * 1e3, rows * fl / ms * 1e-9,
        rows * (p + 1) * X.element_size() / ms * 1e-6))

main()
