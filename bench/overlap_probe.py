"""Do the generator's kernels run NEXT TO the wide fp32 Gram kernel?  Times the Gram of one chunk, the generation of one chunk, and both
on two streams.   python bench/overlap_probe.py [rows] [p]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dlsa_amd import engine

rows = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1 << 22
p = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
dt = torch.float32
X1 = engine.empty_rows(rows, p, dt, "cuda"); X2 = engine.empty_rows(rows, p, dt, "cuda")
y2 = torch.empty(rows, dtype=dt, device="cuda")
engine.synth(1, 0, rows, p, kind=engine.SYNTH_GAUSSIAN, labels=False, dtype=dt, out=X1)
H = torch.zeros((p, p), dtype=torch.float64, device="cuda")
side = torch.cuda.Stream()

def gram():
    engine.gram_acc64(X1, None, out=H, accumulate=True)

def gen():
    engine.synth(2, 0, rows, p, kind=engine.SYNTH_GAUSSIAN, labels=False, dtype=dt, out=X2)
    engine.synth_response(2, 0, X2, out=y2)

def wall(fn, reps=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3

def both(order):
    def f():
        side.wait_stream(torch.cuda.current_stream())
        if order == 0:
            with torch.cuda.stream(side):
                gen()
            gram()
        else:
            gram()
            with torch.cuda.stream(side):
                gen()
        torch.cuda.current_stream().wait_stream(side)
    return f

tg, ts = wall(gram), wall(gen)
print("gram %.1f ms   generation %.1f ms   sum %.1f" % (tg, ts, tg + ts))
print("both, generation enqueued first: %.1f ms" % wall(both(0)))
print("both, gram enqueued first:       %.1f ms" % wall(both(1)))
