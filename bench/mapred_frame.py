"""dlsa_mapred on a stacked pandas frame (K * p rows x (3 + p) columns, the layout the map step emits): wall time per call.
python bench/mapred_frame.py [p] [K]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pandas as pd
import torch
import dlsa_amd

p = int(sys.argv[1]) if len(sys.argv) > 1 else 500
K = int(sys.argv[2]) if len(sys.argv) > 2 else 200
rng = np.random.default_rng(0)
A = rng.standard_normal((4 * p, p)); S = A.T @ A
names = ["x%d" % i for i in range(p)]
blocks = []
for k in range(K):
    Sk = S * (1 + 0.01 * k); ck = rng.standard_normal(p) * 0.1 + 1.0
    b = pd.DataFrame(np.column_stack([np.arange(p), ck, Sk @ ck, Sk]), columns=["par_id", "coef", "Sig_invMcoef"] + names)
    b["par_id"] = b["par_id"].astype(np.int64)
    blocks.append(b)
stacked = pd.concat(blocks, ignore_index=True)
dlsa_amd.dlsa_mapred(stacked, num_partitions=K); torch.cuda.synchronize()
ts = []
for _ in range(3):
    t = time.perf_counter(); out = dlsa_amd.dlsa_mapred(stacked, num_partitions=K); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
print("dlsa_mapred(stacked frame) p=%d K=%d (%d rows, %.0f MB): %.1f ms per call" % (p, K, len(stacked), stacked.memory_usage().sum() / 1e6, min(ts) * 1e3))
