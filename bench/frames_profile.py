"""Where the frame-level reduce stage spends its time at config 1 / config 2 widths: python bench/frames_profile.py [p K]"""
import cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dlsa_amd
from dlsa_amd import engine
p, K = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (50, 20)
n = K * 5000
X, y = engine.synth(20260101, 0, n, p, kind=engine.SYNTH_GAUSSIAN)
offs = [k * 5000 for k in range(K + 1)]


def whole():
    mb = dlsa_amd.fit_logistic_partitions(X, y, part_offsets=offs)
    out = dlsa_amd.dlsa_mapred(mb)
    sel = dlsa_amd.dlsa(out.iloc[:, 2:], out["beta_byOLS"], n)
    return mb, out, sel


for _ in range(5):
    whole()
torch.cuda.synchronize()
stages = {"fit": 0.0, "mapred": 0.0, "dlsa": 0.0}
R = 50
for _ in range(R):
    t0 = time.perf_counter(); mb = dlsa_amd.fit_logistic_partitions(X, y, part_offsets=offs); torch.cuda.synchronize()
    t1 = time.perf_counter(); out = dlsa_amd.dlsa_mapred(mb); torch.cuda.synchronize()
    t2 = time.perf_counter(); sel = dlsa_amd.dlsa(out.iloc[:, 2:], out["beta_byOLS"], n); torch.cuda.synchronize()
    t3 = time.perf_counter()
    stages["fit"] += t1 - t0; stages["mapred"] += t2 - t1; stages["dlsa"] += t3 - t2
print("p=%d K=%d per call (ms):" % (p, K), {k: round(v / R * 1e3, 3) for k, v in stages.items()})
pr = cProfile.Profile()
pr.enable()
for _ in range(R):
    whole()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28)
print(s.getvalue()[:6000])
