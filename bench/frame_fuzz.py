"""Randomised check of the frame-level operator's upload paths (DesignSpec.numeric_to_device: per-column upload of columnar frames,
one strided window of frames built from a row-major array, the host fallback for mixed dtypes / scattered columns):
logistic_model(frame) must give the block of the tensor fit on the same values.  python bench/frame_fuzz.py cases seed"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pandas as pd
import torch
import dlsa_amd
from dlsa_amd import engine
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 150
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
layouts = {}
worst = 0.0
for c in range(cases):
    p = int(rng.integers(1, 40))
    n = int(rng.choice([rng.integers(60 * p + 50, 200 * p + 200), rng.integers(3000, 12000), rng.integers(70000, 120000)]))   # below / above the 2^16-element fast paths
    X = rng.random((n, p)) - 0.5
    beta = np.zeros(p); beta[: max(1, int(0.4 * p))] = 1.0
    y = (rng.random(n) < 1 / (1 + np.exp(-X @ beta))).astype(np.int64)
    names = ["x%d" % i for i in range(p)]
    layout = str(rng.choice(["row_major", "columnar", "scattered", "mixed_dtypes", "row_major_parent_with_label"]))
    layouts[layout] = layouts.get(layout, 0) + 1
    Xe = X
    if layout == "row_major":
        df = pd.DataFrame(X, columns=names); df.insert(0, "label", y); df.insert(0, "partition_id", 0)
    elif layout == "columnar":
        df = pd.DataFrame({"partition_id": 0, "label": y, **{c_: np.ascontiguousarray(X[:, i]) for i, c_ in enumerate(names)}})
    elif layout == "scattered":
        df = pd.DataFrame({"partition_id": 0, "label": y, **{c_: np.ascontiguousarray(X[:, i]) for i, c_ in enumerate(names)}})
        order = list(rng.permutation(names)); cut = int(rng.integers(0, p + 1))
        df = df[["partition_id"] + order[:cut] + ["label"] + order[cut:]]
    elif layout == "mixed_dtypes":
        cols = {"partition_id": 0, "label": y}
        Xe = X.copy()
        for i, c_ in enumerate(names):
            t = int(rng.integers(0, 3))
            if t == 0: cols[c_] = np.ascontiguousarray(X[:, i])
            elif t == 1: v = X[:, i].astype(np.float32); cols[c_] = v; Xe[:, i] = v.astype(np.float64)
            else: v = np.round(X[:, i] * 4).astype(np.int64); cols[c_] = v; Xe[:, i] = v.astype(np.float64)
        df = pd.DataFrame(cols)
    else:      # one float parent holding partition_id, label and the features (the reference's simulate_logistic frame)
        df = pd.DataFrame(np.column_stack([np.zeros(n), y.astype(np.float64), X]), columns=["partition_id", "label"] + names)
    icpt = bool(rng.random() < 0.5)
    out = dlsa_amd.logistic_model(df, "label", fit_intercept=icpt)
    feat = [c_ for c_ in df.columns if c_ not in ("partition_id", "label")]
    Xd = torch.from_numpy(np.ascontiguousarray(Xe[:, [names.index(c_) for c_ in feat]])).cuda()
    yd = torch.from_numpy(y.astype(np.float64)).cuda()
    r = engine.irls_fit_ex(Xd, yd, [0], [n], row_step=1, fit_intercept=icpt)
    assert list(out.columns) == ["par_id", "coef", "Sig_invMcoef"] + (["intercept"] if icpt else []) + feat, (c, layout)
    ref = np.column_stack([r["coef"][0].cpu().numpy(), r["Sig_invMcoef"][0].cpu().numpy(), r["Sig_inv"][0].cpu().numpy()])
    got = out.iloc[:, 1:].to_numpy()
    d = np.sqrt(np.abs(np.diag(ref[:, 2:])))
    scale = np.column_stack([np.full(len(d), np.abs(ref[:, 0]).max()), np.full(len(d), np.abs(ref[:, 1]).max()), d[:, None] * d[None, :]])
    err = float((np.abs(got - ref) / scale).max())       # (the frame path fits the design kernel's matrix, explicit ones column included: same values, other kernels)
    worst = max(worst, err)
    assert r["status"] == [0] and err < 1e-10, ("block", c, layout, n, p, icpt, err)
print("FRAME FUZZ ok: %d cases, layouts %s: worst scaled difference to the tensor fit %.2e" % (cases, layouts, worst))
