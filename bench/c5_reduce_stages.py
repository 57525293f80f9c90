"""Config 5 (capped: 2.4e7 x 2000 fp32, K = 8): the stages after the map -- dlsa_mapred (sum + WLS) and dlsa (LARS + BIC) -- timed one by one.
   DLSA_AB_LIB=... python bench/c5_reduce_stages.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dlsa_amd import _lib
if os.environ.get("DLSA_AB_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["DLSA_AB_LIB"])
import dlsa_amd
from dlsa_amd import engine
n, p, K = int(float(os.environ.get("C5_ROWS", "6e6"))), 2000, 8
X, _ = engine.synth(20260101, 0, n, p, kind=engine.SYNTH_GAUSSIAN, labels=False, dtype=torch.float32)
beta = torch.zeros(p, dtype=torch.float32, device="cuda"); beta[: int(0.4 * p)] = 1.0
y = torch.randn(n, dtype=torch.float32, device="cuda")
for r in range(0, n, 1_000_000):
    y[r:r + 1_000_000] += X[r:r + 1_000_000] @ beta
offs = [int(n * k / K) for k in range(K + 1)]
mb = dlsa_amd.fit_linear_partitions(X, y, part_offsets=offs)
def timed(f, reps=5):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t = time.perf_counter(); r = f(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
    return sorted(ts)[len(ts) // 2] * 1e3, r
t1, o = timed(lambda: dlsa_amd.dlsa_mapred(mb))
t2, d = timed(lambda: dlsa_amd.dlsa(o.iloc[:, 2:], o["beta_byOLS"], n))
Sig = torch.from_numpy(o.iloc[:, 2:].to_numpy()).cuda()
b = torch.from_numpy(o["beta_byOLS"].to_numpy()).cuda()
t3, _ = timed(lambda: engine.lars_path(Sig, b, False, float(n)))
print("dlsa_mapred %.1f ms | dlsa (LARS + selection, DataFrames) %.1f ms | engine.lars_path alone %.1f ms" % (t1, t2, t3), flush=True)
