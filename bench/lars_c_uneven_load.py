"""lars_c.hip under UNEVEN load: the path at p = 1101 (lasso, drops, intercept) repeated while a second stream keeps the chip busy with Gram
launches of another shape -- every workgroup's reads of the rows other workgroups wrote must still be fresh.  Each run must equal the idle
run bit for bit (the column-split kernel ran) or to 1e-9 (a barrier timed out behind the other stream's workgroups and lars.hip's
single-workgroup kernel reran the path); python bench/lars_c_uneven_load.py [runs]"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dlsa_amd import engine, _lib
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from lars_c_check import problem
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 20
S, b, n = problem(1101, 0.97, 11)
Sd, bd = torch.from_numpy(S).cuda(), torch.from_numpy(b).cuda()
ref = engine.lars_path(Sd, bd, True, float(n), type="lasso")
torch.cuda.synchronize()
X, _ = engine.synth(3, 0, 1_500_000, 260, kind=engine.SYNTH_GAUSSIAN, labels=False)
w = torch.rand(X.shape[0], dtype=torch.float64, device="cuda")
stop = False
launched = [0]
def hammer():
    s2 = torch.cuda.Stream()
    with torch.cuda.stream(s2):
        while not stop:
            for _ in range(4): engine.gram(X, w)
            launched[0] += 4
            s2.synchronize()
t = threading.Thread(target=hammer); t.start()
time.sleep(0.2)
aborts0 = _lib.load().dlsa_lars_grid_barrier_timeout(0.0)
same, close, worst, ts = 0, 0, 0.0, []
for i in range(runs):
    t0 = time.perf_counter()
    r = engine.lars_path(Sd, bd, True, float(n), type="lasso")
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    assert r["beta"].shape == ref["beta"].shape, (i, r["beta"].shape)
    if torch.equal(r["beta"], ref["beta"]) and torch.equal(r["BIC"], ref["BIC"]): same += 1
    else:
        e = float((r["beta"] - ref["beta"]).abs().max() / ref["beta"].abs().max()); worst = max(worst, e); close += 1
        assert e < 1e-9, (i, e)
stop = True; t.join()
aborts = _lib.load().dlsa_lars_grid_barrier_timeout(0.0) - aborts0
print("UNEVEN LOAD ok: %d runs beside %d Gram launches on a second stream: %d bit-identical to the idle run, %d within %.1e (%d launches gave up at a barrier and were rerun on one workgroup); "
      "ms per path min %.1f median %.1f max %.1f" % (runs, launched[0], same, close, worst, aborts, min(ts), sorted(ts)[len(ts) // 2], max(ts)))
