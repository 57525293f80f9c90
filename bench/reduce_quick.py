"""Reduce side by stage: block sum, WLS solve, LARS path, frame building.  python bench/reduce_quick.py p [K]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import dlsa_amd
from dlsa_amd import engine

p = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 8
g = torch.Generator(device="cuda").manual_seed(1)
A = torch.randn((4 * p, p), dtype=torch.float64, device="cuda", generator=g)
S1 = A.T @ A
sig = torch.stack([S1 * (1.0 + 0.01 * k) for k in range(K)])
theta = torch.zeros(p, dtype=torch.float64, device="cuda"); theta[: int(0.4 * p)] = 1.0
coef = torch.stack([theta + 0.01 * torch.randn(p, dtype=torch.float64, device="cuda", generator=g) for _ in range(K)])
smc = torch.stack([sig[k] @ coef[k] for k in range(K)])

def wall(fn, reps=3):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t = time.perf_counter(); r = fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
    return min(ts), r

t_sum, msg = wall(lambda: engine.sum_blocks(coef, smc, sig))
S = msg[: p * p].view(p, p); v = msg[p * p: p * p + p]
t_wls, (th, _rank) = wall(lambda: engine.wls_solve(S, v))
t_lars, path = wall(lambda: engine.lars_path(S, th, False, 4 * p * K))
names = ["x%d" % i for i in range(p)]
mb = dlsa_amd.MappedBlocks(coef, smc, sig, names, [0] * K, [1] * K, [0.0] * K, sample_size=4 * p * K)
t_red, out = wall(lambda: dlsa_amd.dlsa_mapred(mb))
t_dlsa, sel = wall(lambda: dlsa_amd.dlsa(out.iloc[:, 2:], out["beta_byOLS"], sample_size=4 * p * K, fit_intercept=False))
print("p=%d K=%d  sum_blocks %.2f ms  wls_solve %.2f ms  lars_lsa %.2f ms | dlsa_mapred (frames) %.2f ms  dlsa (frames) %.2f ms" % (
    p, K, t_sum * 1e3, t_wls * 1e3, t_lars * 1e3, t_red * 1e3, t_dlsa * 1e3))
