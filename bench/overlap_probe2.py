"""Stream concurrency sanity: two half-GPU kernels on two streams."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
print({k: v for k, v in os.environ.items() if any(s in k.upper() for s in ("HIP", "AMD", "HSA", "ROC", "GPU"))})
a = torch.randn(64, 1 << 20, device="cuda"); b = torch.randn(64, 1 << 20, device="cuda")
side = torch.cuda.Stream()
def wall(fn, reps=5):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
# a latency-bound small-grid kernel: one row of 1M elements -> few workgroups busy for a while
def k1(x):
    for _ in range(20): x[0].sin_()
def one(): k1(a)
def two():
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side): k1(b)
    k1(a)
    torch.cuda.current_stream().wait_stream(side)
print("one stream x1: %.2f ms   two streams (each the same work): %.2f ms" % (wall(one), wall(two)))
