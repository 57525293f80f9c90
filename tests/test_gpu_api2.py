"""GPU: round-2 additions to the reference-named API surface -- logistic_model_eval_sdf (dlsa/model_eval.py:10-42), the
CSV -> level codes -> device shard ingestion (projects/logistic_dlsa.py:218-237) feeding the map step, and the coef.csv
table of the driver (projects/results/plot_coef.py:43-51)."""
import os
import subprocess
import sys
import warnings

import numpy as np
import pandas as pd
import pytest

from f4_fixture import load_f4

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOL_MLE = 1e-10


def rel_inf(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return float(np.max(np.abs(a - b)) / max(1e-300, np.max(np.abs(b))))


@pytest.fixture(scope="module")
def api():
    assert torch.cuda.is_available()
    import dlsa_amd
    return dlsa_amd


def test_logistic_model_eval_sdf_sums_partitions_to_the_reference_total(api):
    """model_eval.py:10-42: per-partition logistic_model_eval, summed.  The reference's total over the whole F4 chunk must
    come out whatever the partitioning."""
    z, df, dummy_info, baseline, data_info = load_f4()
    coef = z["coef_mle"]
    par = pd.DataFrame({"beta_byOLS": coef, "beta_half": 0.5 * coef, "beta_zero": 0.0 * coef})
    for K in (1, 3):
        d = df.copy()
        d["partition_id"] = np.arange(len(d)) % K
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")                  # a chunk that lacks a level warns and still evaluates (models.py:187-194)
            out = api.logistic_model_eval_sdf(d, par, True, "label", dummy_info, baseline, data_info)
        assert list(out.columns) == list(par.columns) and out.shape == (1, 3)
        assert rel_inf(out.to_numpy().ravel(), z["eval_loglik"]) < TOL_MLE


def test_loglik_partitions_tensor_path(api):
    from oracle import dlsa_oracle as orc
    X, y = orc.synth_logistic(31, 0, 9000, 12)
    par = np.random.default_rng(1).normal(size=(13, 10)) * 0.3          # 10 columns: two passes of <= 8
    got = api.loglik_partitions(torch.from_numpy(X).cuda(), torch.from_numpy(y).cuda(), par, fit_intercept=True).cpu().numpy()
    want = orc.logistic_loglik(np.column_stack([np.ones(9000), X]), y, par)
    assert rel_inf(got, want) < 1e-12


def test_csv_ingestion_feeds_the_map_step(api, tmp_path):
    """CSV -> select / dropna / binarise -> level selection -> level codes on the device -> fit_logistic_design, against
    logistic_model on the same partitions' frames (the reference-faithful path, pinned by F4)."""
    from dlsa_amd import dummies, ingest
    rng = np.random.default_rng(8)
    n = 9000
    raw = pd.DataFrame({"Distance": rng.normal(700, 300, n), "DepTime": rng.uniform(0, 2400, n),
                        "UniqueCarrier": rng.choice(["AA", "UA", "DL", "WN", "HP", "TW"], n, p=[.35, .25, .2, .12, .05, .03]),
                        "DayOfWeek": rng.integers(1, 8, n), "FlightNum": rng.integers(1, 999, n)})
    eta = 0.5 * (raw["Distance"] - 700) / 300 - 0.4 * (raw["UniqueCarrier"] == "UA") + 0.3 * (raw["DayOfWeek"] == 5)
    raw["ArrDelay"] = np.where(rng.random(n) < 1 / (1 + np.exp(-eta)), rng.uniform(1, 60, n), -rng.uniform(0, 30, n))
    raw.loc[rng.choice(n, 40, replace=False), "DepTime"] = np.nan
    path = str(tmp_path / "airline.csv")
    raw.to_csv(path, index=False, na_rep="NA")
    usecols, dummy_cols = ["Distance", "DepTime", "UniqueCarrier", "DayOfWeek"], ["UniqueCarrier", "DayOfWeek"]
    pdf = ingest.read_csv_frame(path, usecols, "ArrDelay", dummy_columns=dummy_cols)
    assert len(pdf) == n - 40
    dummy_info = dummies.select_dummy_factors(dummies.dummy_factors_counts(pdf, dummy_cols), [0.93, 1], "000_OTHERS")
    assert dummy_info["factor_dropped"]["UniqueCarrier"] == ["HP", "TW"]
    baseline = ["UniqueCarrier_000_OTHERS", "DayOfWeek_" + str(dummy_info["factor_selected"]["DayOfWeek"][0])]
    data_info = ingest.data_info_from_frame(pdf, ["Distance", "DepTime"])
    sh = ingest.shard_from_frame(pdf, "ArrDelay", dummy_info, baseline, data_info, True, sample_size_per_partition=3000)
    K = sh["partition_num"]
    assert K == 3 and sh["partitions"] == [0, 1, 2] and int(sh["part_offsets"][-1]) == len(pdf) and not sh["unknown_levels"]
    mb = api.fit_logistic_design(sh["num"], sh["codes"], sh["y"], sh["spec"], part_offsets=sh["part_offsets"])
    assert mb.status == [0] * K
    pid = np.arange(len(pdf)) % K
    for k in range(K):
        frame = pdf[pid == k].reset_index(drop=True)
        out = api.logistic_model(frame, "ArrDelay", fit_intercept=True, dummy_info=dummy_info,
                                 dummy_factors_baseline=baseline, data_info=data_info)
        assert list(out.columns[3:]) == sh["spec"].names
        assert rel_inf(mb.coef[k].cpu().numpy(), out["coef"].to_numpy()) < TOL_MLE
        assert rel_inf(mb.Sig_inv[k].cpu().numpy(), out.iloc[:, 3:].to_numpy()) < TOL_MLE
    # two ranks: each owns its partitions, together they cover the frame
    a = ingest.shard_from_frame(pdf, "ArrDelay", dummy_info, baseline, data_info, True, 3000, world=2, rank=0)
    b = ingest.shard_from_frame(pdf, "ArrDelay", dummy_info, baseline, data_info, True, 3000, world=2, rank=1)
    assert a["partitions"] == [0, 2] and b["partitions"] == [1]
    assert int(a["part_offsets"][-1]) + int(b["part_offsets"][-1]) == len(pdf)


def test_driver_writes_the_coef_csv(tmp_path):
    out = str(tmp_path / "coef.csv")
    pr = subprocess.run([sys.executable, os.path.join(ROOT, "projects", "logistic_dlsa.py"), "--sample-size", "20000", "--p", "12",
                         "--partition-num", "4", "--fit-intercept", "--coef-csv", out], stdout=subprocess.PIPE,
                        stderr=subprocess.PIPE, text=True, timeout=600)
    assert pr.returncode == 0, pr.stderr[-2000:]
    tab = pd.read_csv(out, index_col="Var")
    assert list(tab.columns) == ["MLE", "DLSA_AIC", "DLSA_BIC", "WLSE", "ONE_SHOT"]
    assert list(tab.index) == ["intercept"] + ["x%d" % i for i in range(12)]
    assert np.isfinite(tab.to_numpy()).all()
    # the global MLE and the WLS estimate of 4 partitions of a well-specified model agree to O(1/n)
    assert float(np.max(np.abs(tab["MLE"] - tab["WLSE"]))) < 0.05


def test_cabi_rccl_allreduce_single_rank_communicator():
    """dlsa_allreduce_f64: the one-round reduce through the C ABI on an RCCL communicator the library opened
    itself.  The test box has one GPU, so the communicator has one rank (RCCL refuses two ranks on a device): the message
    must come back unchanged, on the caller's stream, and the library must have resolved RCCL at run time."""
    from dlsa_amd import engine
    uid = engine.RcclComm.unique_id()
    assert len(uid) == 128 and any(uid)
    comm = engine.RcclComm(1, uid, 0)
    p = 500
    msg = torch.randn(p * p + 2 * p, dtype=torch.float64, device="cuda")
    ref = msg.clone()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        out = comm.allreduce(msg)
        out = comm.allreduce(out)
    s.synchronize()
    assert out.data_ptr() == msg.data_ptr() and torch.equal(msg, ref)
    with pytest.raises(TypeError):
        comm.allreduce(msg.float())
    comm.close()


def test_logistic_model_same_block_for_columnar_and_row_major_frames(api):
    """The frame-level operator takes the numeric columns as pandas holds them: a frame built column by column (the Arrow /
    read_csv layout: uploaded per column, transposed in HBM), one built from a row-major array, and one with the feature
    columns scattered between other columns must give the same block, bit for bit."""
    import numpy as np
    import pandas as pd
    rng = np.random.default_rng(12)
    n, p = 3000, 24                                  # n * p >= 2^16: the per-column upload path
    X = rng.random((n, p)) - 0.5
    y = (rng.random(n) < 1 / (1 + np.exp(-X[:, :8].sum(1)))).astype(np.int64)
    names = ["x%d" % i for i in range(p)]
    row_major = pd.DataFrame(X, columns=names); row_major.insert(0, "label", y); row_major.insert(0, "partition_id", 0)
    columnar = pd.DataFrame({"partition_id": 0, "label": y, **{c: np.ascontiguousarray(X[:, i]) for i, c in enumerate(names)}})
    scattered = columnar[["partition_id"] + names[:5] + ["label"] + names[5:]]
    outs = [api.logistic_model(df, "label", fit_intercept=True) for df in (row_major, columnar, scattered)]
    assert list(outs[0].columns) == ["par_id", "coef", "Sig_invMcoef", "intercept"] + names
    for o in outs[1:]:
        assert list(o.columns) == list(outs[0].columns)
        assert np.array_equal(o.to_numpy(), outs[0].to_numpy())


def test_single_numeric_column_frames_and_column_views(api):
    """A size-1 dimension is 'contiguous' with any stride (numpy and torch keep the parent's pitch on it): a one-feature frame and a
    one-column view of a wider device matrix must both go through (found by bench/frame_fuzz.py: the row-major check refused them)."""
    import numpy as np
    import pandas as pd
    from dlsa_amd import engine
    rng = np.random.default_rng(4)
    n = 70000                                        # above the 2^16-element threshold of the per-column upload
    x = rng.random(n) - 0.5
    y = (rng.random(n) < 1 / (1 + np.exp(-2 * x))).astype(np.int64)
    wide = np.column_stack([np.zeros(n), y.astype(np.float64), x])               # one row-major parent: the feature is a strided [n, 1] view
    frames = [pd.DataFrame({"partition_id": 0, "label": y, "x0": x}), pd.DataFrame(wide, columns=["partition_id", "label", "x0"])]
    outs = [api.logistic_model(df, "label", fit_intercept=True) for df in frames]
    assert np.array_equal(outs[0].to_numpy(), outs[1].to_numpy())
    Xw = torch.from_numpy(np.column_stack([x, rng.random(n), rng.random(n)])).cuda()
    yd = torch.from_numpy(y.astype(np.float64)).cuda()
    r_view = engine.irls_fit_ex(Xw[:, 0:1], yd, [0], [n], row_step=1, fit_intercept=True)       # pitch 3, one column
    r_col = engine.irls_fit_ex(torch.from_numpy(x[:, None].copy()).cuda(), yd, [0], [n], row_step=1, fit_intercept=True)
    assert r_view["status"] == [0] and torch.equal(r_view["coef"], r_col["coef"]) and torch.equal(r_view["Sig_inv"], r_col["Sig_inv"])
    assert np.allclose(outs[0]["coef"].to_numpy(), r_col["coef"][0].cpu().numpy(), rtol=1e-10, atol=0)


def test_driver_real_data_mode_from_csv(api, tmp_path):
    """projects/logistic_dlsa.py --csv: the reference's real-data branch (logistic_dlsa.py:100-175, 218-237) -- CSV -> select /
    dropna / binarise -> dummy levels (created and pickled, then LOADED on the second run) -> standardisation table -> level codes ->
    structured fit per partition -> dlsa_mapred -> dlsa -> evaluation -> pickle [Sig_inv_beta, out_dlsa, out_par, out_model_eval,
    out_time].  Checked against the library calls made directly on the same frame."""
    import pickle
    from dlsa_amd import dummies, ingest
    rng = np.random.default_rng(21)
    n = 12000
    raw = pd.DataFrame({"Distance": rng.normal(700, 300, n), "DepTime": rng.uniform(0, 2400, n),
                        "UniqueCarrier": rng.choice(["AA", "UA", "DL", "WN", "HP", "TW"], n, p=[.35, .25, .2, .12, .05, .03]),
                        "DayOfWeek": rng.integers(1, 8, n), "FlightNum": rng.integers(1, 999, n)})
    eta = 0.5 * (raw["Distance"] - 700) / 300 - 0.4 * (raw["UniqueCarrier"] == "UA") + 0.3 * (raw["DayOfWeek"] == 5)
    raw["ArrDelay"] = np.where(rng.random(n) < 1 / (1 + np.exp(-eta)), rng.uniform(1, 60, n), -rng.uniform(0, 30, n))
    raw.loc[rng.choice(n, 50, replace=False), "Distance"] = np.nan
    path = str(tmp_path / "air.csv")
    raw.to_csv(path, index=False, na_rep="NA")
    usecols, dcols = ["Distance", "DepTime", "UniqueCarrier", "DayOfWeek"], ["UniqueCarrier", "DayOfWeek"]
    common = [sys.executable, os.path.join(ROOT, "projects", "logistic_dlsa.py"), "--csv", path, "--y-name", "ArrDelay",
              "--usecols-x", ",".join(usecols), "--dummy-columns", ",".join(dcols), "--dummy-keep-top", "0.93,1", "--fit-intercept",
              "--sample-size-per-partition", "4000", "--dummy-info", str(tmp_path / "dummy_info.pkl"), "--data-info", str(tmp_path / "data_info.csv")]
    outs = []
    for run in range(2):                                  # second run: dummy_info / data_info are loaded from the files the first wrote
        save = str(tmp_path / ("res%d.pkl" % run))
        pr = subprocess.run(common + ["--save", save], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
        assert pr.returncode == 0, pr.stderr[-2000:]
        with open(save, "rb") as f:
            outs.append(pickle.load(f))
    assert os.path.exists(tmp_path / "dummy_info.pkl") and os.path.exists(tmp_path / "data_info.csv")
    Sig_inv_beta, out_dlsa, out_par, out_model_eval, out_time = outs[0]
    for a, b in zip(outs[0][:4], outs[1][:4]):
        assert np.array_equal(a.to_numpy(), b.to_numpy())
    # the same steps through the library
    pdf = ingest.read_csv_frame(path, usecols, "ArrDelay", dummy_columns=dcols)
    info = dummies.select_dummy_factors(dummies.dummy_factors_counts(pdf, dcols), [0.93, 1], "000_OTHERS")
    baseline = ["UniqueCarrier_000_OTHERS", sorted(info["factor_selected_names"]["DayOfWeek"])[0]]
    data_info = ingest.data_info_from_frame(pdf, ["Distance", "DepTime"])
    sh = ingest.shard_from_frame(pdf, "ArrDelay", info, baseline, data_info, True, sample_size_per_partition=4000)
    K = sh["partition_num"]
    assert K == 3 and int(out_time["partition_num"][0]) == 3 and int(out_time["sample_size"][0]) == len(pdf) == n - 50
    ref = api.dlsa_mapred(api.fit_logistic_design(sh["num"], sh["codes"], sh["y"], sh["spec"], part_offsets=sh["part_offsets"]), num_partitions=K)
    assert list(Sig_inv_beta.columns) == ["beta_byOLS", "beta_byONESHOT"] + sh["spec"].names and sh["spec"].names[0] == "intercept"
    assert rel_inf(Sig_inv_beta.to_numpy(), ref.to_numpy()) < 1e-12
    assert list(out_par.columns) == ["beta_byAIC", "beta_byBIC", "beta_byOLS", "beta_byONESHOT"]
    # evaluation: the log-likelihood of each estimator over all rows, against the frame-level evaluation of the same frame
    pdf.insert(0, "partition_id", np.arange(len(pdf)) % K)
    ev = api.logistic_model_eval_sdf(pdf, out_par, True, "ArrDelay", info, baseline, data_info)
    assert rel_inf(out_model_eval.to_numpy(), ev.to_numpy()) < 1e-10


def test_the_ctypes_stub_printed_in_integration_md_runs_as_written():
    """INTEGRATION.md section 3 shows the ctypes binding a reference maintainer would add; the block is executed here as printed
    (only the library path is made absolute) and its fit compared with the oracle."""
    import re
    from oracle import dlsa_oracle as orc
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    m = re.search(r"```python\nimport ctypes, numpy as np, torch\n(.*?)```", text, re.S)
    assert m, "the stub moved"
    code = "import ctypes, numpy as np, torch\n" + m.group(1)
    code = code.replace('ctypes.CDLL("libdlsa_hip.so")', 'ctypes.CDLL(%r)' % os.path.join(ROOT, "dlsa_amd", "libdlsa_hip.so"))
    ns = {}
    exec(compile(code, "INTEGRATION.md", "exec"), ns)
    rng = np.random.default_rng(0)
    n, p = 20000, 12
    X = rng.random((n, p)) - 0.5
    beta = np.zeros(p); beta[:5] = 1.0
    y = (rng.random(n) < 1 / (1 + np.exp(-X @ beta))).astype(np.float64)
    coef, smc, sig = ns["fit_block"](X, y)
    c, s2, S = orc.logistic_model_block(X, y)
    assert rel_inf(coef, c) < 1e-10 and rel_inf(sig, S) < 1e-10 and rel_inf(smc, s2) < 1e-10
