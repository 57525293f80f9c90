"""The frame-level operator (the reference's pandas UDF boundary): wall time of logistic_model(sample_df, ...) by stage.
python bench/frame_path.py"""
import os, sys, time, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pandas as pd
import torch
import dlsa_amd

def frame(n, p, seed=1, columnar=False):
    rng = np.random.default_rng(seed)
    X = rng.random((n, p)) - 0.5
    beta = np.zeros(p); beta[: int(0.4 * p)] = 1.0
    y = (rng.random(n) < 1 / (1 + np.exp(-X @ beta))).astype(np.int64)
    if columnar:      # the layout Arrow / read_csv hand over: column-major blocks
        df = pd.DataFrame({"x%d" % i: np.ascontiguousarray(X[:, i]) for i in range(p)})
    else:
        df = pd.DataFrame(X, columns=["x%d" % i for i in range(p)])
    df.insert(0, "label", y); df.insert(0, "partition_id", 0)
    return df

for n, p, columnar in [(5000, 50, False), (100000, 100, False), (1000000, 100, False), (1000000, 100, True), (200000, 500, False), (200000, 500, True)]:
    df = frame(n, p, columnar=columnar)
    dlsa_amd.logistic_model(df, "label"); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t = time.perf_counter(); out = dlsa_amd.logistic_model(df, "label"); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
    Xd = torch.from_numpy(df.iloc[:, 2:].to_numpy()).cuda(); yd = torch.from_numpy(df["label"].to_numpy().astype(np.float64)).cuda()
    t = time.perf_counter(); mb = dlsa_amd.fit_logistic_partitions(Xd, yd, partition_num=1); torch.cuda.synchronize(); t_fit = time.perf_counter() - t
    t = time.perf_counter(); mb = dlsa_amd.fit_logistic_partitions(Xd, yd, partition_num=1); torch.cuda.synchronize(); t_fit = time.perf_counter() - t
    t = time.perf_counter(); a = df.iloc[:, 2:].to_numpy(); t_np = time.perf_counter() - t
    t = time.perf_counter(); torch.from_numpy(np.ascontiguousarray(a)).cuda(); torch.cuda.synchronize(); t_h2d = time.perf_counter() - t
    print(("columnar " if columnar else "row-major ") + "n=%d p=%d  logistic_model(frame) %.2f ms  | tensor fit %.2f ms, frame -> numpy %.2f ms, host -> HBM %.2f ms (%.1f GB/s)" % (
        n, p, min(ts) * 1e3, t_fit * 1e3, t_np * 1e3, t_h2d * 1e3, a.nbytes / t_h2d / 1e9), flush=True)
    if (n, p) == (1000000, 100) and columnar:
        pr = cProfile.Profile(); pr.enable()
        for _ in range(5):
            dlsa_amd.logistic_model(df, "label")
        pr.disable(); s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(14); print(s.getvalue()[:2600])
