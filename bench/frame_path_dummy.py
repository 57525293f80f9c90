"""The frame-level operator on an airline-shaped chunk (7 numeric columns + 5 string factors, the reference's dummy path,
models.py:56-104): wall time of logistic_model(sample_df, dummy_info=..., ...) and where the host time goes.
python bench/frame_path_dummy.py [rows]"""
import os, sys, time, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pandas as pd
import torch
import dlsa_amd
from dlsa_amd import dummies

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1000000
rng = np.random.default_rng(3)
levels = (11, 6, 20, 110, 110)
cols = {"partition_id": np.zeros(n, np.int64)}
num = rng.normal(size=(n, 7)) * 3 + 1.5
for j in range(7):
    cols["num%d" % j] = num[:, j].copy()
codes = []
for fi, L in enumerate(levels):
    pr = 1.0 / np.arange(1, L + 1); pr /= pr.sum()
    c = rng.choice(L, size=n, p=pr)
    codes.append(c)
    cols["fac%d" % fi] = np.array(["L%03d" % v for v in range(L)])[c]
eta = 0.2 * num[:, 0] - 0.1 * num[:, 1] + 0.3 * (codes[0] == 1) - 0.2 * (codes[3] == 2)
cols["label"] = (rng.random(n) < 1 / (1 + np.exp(-eta))).astype(np.int64)
df = pd.DataFrame(cols)
facs = ["fac%d" % i for i in range(5)]
dummy_info = dummies.select_dummy_factors(dummies.dummy_factors_counts(df, facs), keep_top=[1, 1, 0.8, 0.9, 0.9], replace_with="000_OTHERS")
baseline = [sorted(dummy_info["factor_selected_names"][f])[0] for f in facs]
data_info = pd.DataFrame({c: [0.0, float(df[c].mean()), float(df[c].std())] for c in ["num%d" % j for j in range(7)]})

def call():
    return dlsa_amd.logistic_model(df, "label", fit_intercept=True, dummy_info=dummy_info, dummy_factors_baseline=baseline, data_info=data_info)

out = call(); torch.cuda.synchronize()
ts = []
for _ in range(3):
    t = time.perf_counter(); out = call(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
print("airline-shaped frame n=%d: logistic_model %.1f ms per call, block %s" % (n, min(ts) * 1e3, out.shape), flush=True)
pr = cProfile.Profile(); pr.enable(); call(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(22); print(s.getvalue()[:4200])
