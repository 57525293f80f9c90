"""Randomised check of the reduce side (engine.sum_blocks + engine.wls_solve: Cholesky for an SPD sum, minimum-norm least squares
for a singular one, dlsa.py:30-49) against numpy's lstsq on the host.  python bench/reduce_fuzz.py cases seed"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from dlsa_amd import engine
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
worst = dict(sum=0.0, spd=0.0, deficient=0.0)
kinds = dict(spd=0, zero_column=0, duplicate_column=0)
for c in range(cases):
    p = int(rng.choice([rng.integers(1, 20), rng.integers(20, 130), rng.integers(130, 420)]))
    K = int(rng.integers(1, 9))
    kind = str(rng.choice(["spd", "spd", "zero_column", "duplicate_column"])) if p > 2 else "spd"
    kinds[kind] += 1
    sig, coef = [], []
    A0 = rng.standard_normal((4 * p + 8, p))
    if kind == "zero_column":
        A0[:, rng.integers(0, p)] = 0.0                       # a dummy level present in no partition (models.py:84-91)
    elif kind == "duplicate_column":
        i, j = rng.choice(p, 2, replace=False); A0[:, j] = A0[:, i]
    for k in range(K):
        Ak = A0 * (1.0 + 0.05 * rng.standard_normal((1, p)) * (kind == "spd")) + 0.1 * rng.standard_normal(A0.shape) * (kind == "spd")
        sig.append(Ak.T @ Ak); coef.append(rng.standard_normal(p))
    sig = np.stack(sig); coef = np.stack(coef); smc = np.einsum("kij,kj->ki", sig, coef)
    dsig, dcoef, dsmc = (torch.from_numpy(a).cuda() for a in (sig, coef, smc))
    msg = engine.sum_blocks(dcoef, dsmc, dsig).cpu().numpy()
    S, v, cs = sig.sum(0), smc.sum(0), coef.sum(0)
    e = max(np.abs(msg[: p * p].reshape(p, p) - S).max() / np.abs(S).max(), np.abs(msg[p * p: p * p + p] - v).max() / (np.abs(v).max() + 1e-300),
            np.abs(msg[p * p + p:] - cs).max() / (np.abs(cs).max() + 1e-300))
    worst["sum"] = max(worst["sum"], e)
    assert e < 1e-13, ("sum", c, p, K, e)
    theta, rank = engine.wls_solve(torch.from_numpy(S).cuda(), torch.from_numpy(v).cuda())
    ref, _, rk, _ = np.linalg.lstsq(S, v, rcond=None)
    e = np.abs(theta.cpu().numpy() - ref).max() / (np.abs(ref).max() + 1e-300)
    if kind == "spd":
        worst["spd"] = max(worst["spd"], e)
        assert rank == p and e < 1e-9, ("spd", c, p, K, rank, e)
    else:
        worst["deficient"] = max(worst["deficient"], e)
        assert rank == rk == p - 1 and e < 1e-8, (kind, c, p, K, rank, rk, e)
print("REDUCE FUZZ ok: %d cases %s, worst %s" % (cases, kinds, {k: "%.2e" % v for k, v in worst.items()}))
