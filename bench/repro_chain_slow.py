import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from dlsa_amd import engine
import surrogates
def oh(tag):
    n, K = 14_000_000, 14
    d = surrogates.airline_shaped(n, dense=False)
    offs = [int(n * k / K) for k in range(K + 1)]
    engine.onehot_irls_fit(d["plan"], d["num"], d["codes"], d["y"], offs); torch.cuda.synchronize()
    t = time.perf_counter(); r = engine.onehot_irls_fit(d["plan"], d["num"], d["codes"], d["y"], offs); torch.cuda.synchronize()
    print(tag, "structured fit %.4f s" % (time.perf_counter() - t), r["n_iter"], flush=True)
step = sys.argv[1]
if len(sys.argv) < 3:
    oh("fresh process:")
if step == "alloc":
    big = torch.empty(int(100e9), dtype=torch.uint8, device="cuda"); del big
elif step == "fit_small_k1":
    X, y = engine.synth(1, 0, 2_000_000, 500); engine.irls_fit(X, y, [0, 2_000_000]); del X, y
elif step == "fit_big_k1":
    X, y = engine.synth(1, 0, 25_000_000, 500); engine.irls_fit(X, y, [0, 25_000_000]); del X, y
elif step == "lars500":
    g = torch.Generator(device="cuda").manual_seed(1); A = torch.randn((3000, 500), dtype=torch.float64, device="cuda", generator=g)
    engine.lars_path(A.T @ A, torch.randn(500, dtype=torch.float64, device="cuda", generator=g), False, 3000.0)
elif step == "lars100":
    g = torch.Generator(device="cuda").manual_seed(1); A = torch.randn((600, 100), dtype=torch.float64, device="cuda", generator=g)
    engine.lars_path(A.T @ A, torch.randn(100, dtype=torch.float64, device="cuda", generator=g), False, 600.0)
elif step == "gram_big":
    X, y = engine.synth(1, 0, 25_000_000, 500); engine.gram(X, None); del X, y
torch.cuda.synchronize()
oh("after %s:" % step)
torch.cuda.empty_cache(); engine.release_workspaces() if hasattr(engine, "release_workspaces") else None
oh("after empty_cache + release_workspaces:")
