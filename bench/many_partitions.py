"""Many partitions of a narrow design: the lock-step driver (csrc/irls_batch.hip) against the chained host-driven path.
   python bench/many_partitions.py [K nk p] ...   (default: the shapes of VERDICT r3 'missing 5' and a few around them)
   MP_ICPT=1: fit_intercept (the reference's driver always does, logistic_dlsa.py:79);  MP_STRIDED=1: partition_id = i % K (models.py:33)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dlsa_amd import _lib
if os.environ.get("DLSA_AB_LIB"):          # same-box A/B of a build variant (bench/build_variant.sh)
    _lib.LIB_PATH = os.path.abspath(os.environ["DLSA_AB_LIB"])
import dlsa_amd
from dlsa_amd import engine

shapes = [tuple(int(float(v)) for v in sys.argv[i:i + 3]) for i in range(1, len(sys.argv) - 2, 3)] or \
    [(1000, 20000, 100), (200, 100000, 100), (100, 20000, 64), (50, 400000, 100), (10, 1000000, 100)]
for K, nk, p in shapes:
    X, y = engine.synth(20260101, 0, K * nk, p, kind=engine.SYNTH_GAUSSIAN)
    offs = [k * nk for k in range(K + 1)]
    icpt, strided = os.environ.get("MP_ICPT", "0") != "0", os.environ.get("MP_STRIDED", "0") != "0"
    part = dict(partition_num=K) if strided else dict(part_offsets=offs)
    res = {}
    dlsa_amd.fit_logistic_partitions(X, y, fit_intercept=icpt, **part)          # (untimed: the first calls on freshly written rows run up to 2x slower)
    for name, opt in (("auto", {}), ("lock step", dict(batched=True, small=False)), ("own start", dict(batched=True, small=False, pooled_start=False)),
                      ("chains", dict(batched=False, small=False))):
        ts = []
        for _ in range(3):
            torch.cuda.synchronize(); t = time.perf_counter()
            mb = dlsa_amd.fit_logistic_partitions(X, y, fit_intercept=icpt, **part, **opt)
            torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
        if os.environ.get("MP_VERBOSE"): print("   %s: %s ms" % (name, ["%.1f" % (v * 1e3) for v in ts]), flush=True)
        res[name] = (sorted(ts)[1], engine.irls_last_fit_path(), mb.n_iter[:2], all(s == 0 for s in mb.status), mb)
    err = float((res["lock step"][4].coef - res["chains"][4].coef).abs().max())
    print("%s%sK=%5d n_k=%8d p=%3d (%.1f GB): auto %.1f ms (path %d) | lock step %.1f ms iters %s (own-subsample start %.1f ms iters %s) | chains %.1f ms iters %s | ok %s %s | max coef diff %.1e" % (
        "intercept " if icpt else "", "i%K " if strided else "", K, nk, p, K * nk * p * 8 / 1e9, res["auto"][0] * 1e3, res["auto"][1], res["lock step"][0] * 1e3, res["lock step"][2],
        res["own start"][0] * 1e3, res["own start"][2], res["chains"][0] * 1e3, res["chains"][2], res["lock step"][3], res["chains"][3], err), flush=True)
    del X, y, res
