"""Randomised check of the frame-level dummy path (logistic_model with dummy_info / baselines / data_info, models.py:56-104):
the structured fit on raw numerics + level codes against the dense fit of the design kernel's matrix on the same frame, including
chunks that lack a selected level (the all-zero block) and folded (dropped) levels.  python bench/dummy_frame_fuzz.py cases seed"""
import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pandas as pd
import torch
import dlsa_amd
from dlsa_amd import dummies
from dlsa_amd.design import DesignSpec
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
stats = dict(fitted=0, zero_block=0, dense_only=0)
worst = 0.0
plan_of = DesignSpec.onehot_plan
for c in range(cases):
    n = int(rng.choice([rng.integers(3000, 9000), rng.integers(9000, 60000)]))
    q = int(rng.integers(1, 6)); f = int(rng.integers(1, 4))
    cols = {"partition_id": np.zeros(n, np.int64)}
    num = rng.normal(size=(n, q)) * rng.uniform(0.5, 3, q) + rng.normal(size=q)
    for j in range(q):
        cols["num%d" % j] = num[:, j].copy()
    eta = 0.4 * num[:, 0] / num[:, 0].std()
    facs = []
    for t in range(f):
        L = int(rng.integers(3, 25))
        pr = 1.0 / np.arange(1, L + 1) ** rng.uniform(0.5, 1.5); pr /= pr.sum()
        code = rng.choice(L, size=n, p=pr)
        labels = np.array([("lv%02d" % v) if rng.random() < 0.8 else str(v) for v in range(L)])
        kind = int(rng.integers(0, 3))
        if kind == 0: cols["fac%d" % t] = labels[code]
        elif kind == 1: cols["fac%d" % t] = code.astype(np.int64)                  # integer-coded factor
        else: cols["fac%d" % t] = pd.Categorical(labels[code])
        eta = eta + 0.3 * (code == 1) - 0.2 * (code == 2)
        facs.append("fac%d" % t)
    cols["label"] = (rng.random(n) < 1 / (1 + np.exp(-eta))).astype(np.int64)
    df = pd.DataFrame(cols)
    keep = [float(rng.choice([1.0, 0.95, 0.85, 0.7])) for _ in facs]
    info = dummies.select_dummy_factors(dummies.dummy_factors_counts(df, facs), keep_top=keep, replace_with="000_OTHERS")
    baseline = [sorted(info["factor_selected_names"][fc])[0] for fc in facs]
    data_info = pd.DataFrame({cn: [0.0, float(df[cn].mean()), float(df[cn].std())] for cn in ["num%d" % j for j in range(q)]}) if rng.random() < 0.6 else []
    icpt = bool(rng.random() < 0.7)
    chunk = df.iloc[: int(rng.choice([n, n, max(400, n // 8)]))]                    # a small chunk may lack a selected level
    kw = dict(fit_intercept=icpt, dummy_info=info, dummy_factors_baseline=baseline, data_info=data_info)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        DesignSpec.onehot_plan = plan_of
        a = dlsa_amd.logistic_model(chunk, "label", **kw)
        structured_used = plan_of(DesignSpec.from_reference(list(chunk.columns), "label", icpt, info, baseline, data_info)) is not None
        DesignSpec.onehot_plan = lambda self: None                                   # force the dense design-kernel path
        b = dlsa_amd.logistic_model(chunk, "label", **kw)
        DesignSpec.onehot_plan = plan_of
    assert list(a.columns) == list(b.columns) and a.shape == b.shape, (c, list(a.columns)[:6], list(b.columns)[:6])
    A, B = a.iloc[:, 1:].to_numpy(), b.iloc[:, 1:].to_numpy()
    if not B.any():
        stats["zero_block"] += 1
        assert not A.any() and not a["par_id"].to_numpy().any(), ("zero block", c)
        continue
    if not structured_used:
        stats["dense_only"] += 1
    d = np.sqrt(np.abs(np.diag(B[:, 2:])))
    if (d == 0).any() or not np.isfinite(B).all() or max(np.abs(A[:, 0]).max(), np.abs(B[:, 0]).max()) > 12.0:
        stats["separated"] = stats.get("separated", 0) + 1
        continue                                                                    # a degenerate chunk (separated / empty level): both report it
    scale = np.column_stack([np.full(len(d), np.abs(B[:, 0]).max()), np.full(len(d), np.abs(B[:, 1]).max()), d[:, None] * d[None, :]])
    err = float((np.abs(A - B) / scale).max())
    worst = max(worst, err); stats["fitted"] += 1
    if not err < 1e-8:
        i, j = np.unravel_index(np.argmax(np.abs(A - B) / scale), A.shape)
        print("MISMATCH case %d n=%d chunk=%d q=%d f=%d icpt=%s structured=%s err=%.3e at (%d,%d): %r vs %r" % (c, n, len(chunk), q, f, icpt, structured_used, err, i, j, A[i, j], B[i, j]))
        print(" columns", list(a.columns)); print(" coef structured", A[:, 0]); print(" coef dense     ", B[:, 0])
        print(" level counts in chunk", {fc: chunk[fc].astype(str).value_counts().to_dict() for fc in facs})
        print(" dummy_info", {k: info[k] for k in ("factor_selected", "factor_dropped")}, "baseline", baseline, "keep", keep)
        sys.exit(1)
print("DUMMY FRAME FUZZ ok: %d cases %s, worst scaled difference structured vs dense %.2e" % (cases, stats, worst))
