"""GPU tests of the reference-compatible Python surface (dlsa_amd.models / .dlsa / .lsa):
same call sequence, column names and results as the reference's own run (golden fixtures)."""
import glob
import os
import warnings

import numpy as np
import pandas as pd
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

TOL_MLE = 1e-10


def rel_inf(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return float(np.max(np.abs(a - b)) / max(1e-300, np.max(np.abs(b))))


@pytest.fixture(scope="module")
def api():
    assert torch.cuda.is_available()
    import dlsa_amd
    return dlsa_amd


@pytest.fixture(scope="module")
def orc():
    from oracle import dlsa_oracle
    return dlsa_oracle


def _frame(X, y, K, names):
    pid = np.arange(X.shape[0]) % K
    return pd.DataFrame(np.column_stack([pid, y, X]), columns=["partition_id", "label"] + names)


F1 = sorted(os.path.basename(f)[3:-8] for f in glob.glob(os.path.join(GOLDEN, "F1_*_mle.npz")))


@pytest.mark.parametrize("name", F1)
def test_reference_call_sequence_matches_golden(api, orc, name):
    """logistic_model per partition -> dlsa_mapred -> dlsa, exactly as projects/logistic_dlsa.py
    drives them (:305-316, :337, :341-344), against the reference's own outputs."""
    z = np.load(os.path.join(GOLDEN, "F1_%s_mle.npz" % name))
    if name.startswith("synth"):
        X, y = orc.synth_logistic(int(z["seed"]), 0, int(z["n"]), int(z["p"]))
    else:
        g = np.load(os.path.join(GOLDEN, "games_expand_input.npz"))
        X, y = g["X"].astype(float), g["y"].astype(float)
    K, icpt = int(z["K"]), bool(z["fit_intercept"])
    names = ["x%d" % i for i in range(X.shape[1])]
    df = _frame(X, y, K, names)
    outs = [api.logistic_model(df[df.partition_id == k].reset_index(drop=True), "label", fit_intercept=icpt)
            for k in range(K)]
    assert list(outs[0].columns) == list(z["columns"])
    for k in range(K):
        assert list(outs[k]["par_id"]) == list(range(outs[k].shape[0]))
        assert rel_inf(outs[k]["coef"], z["coef"][k]) < TOL_MLE
        assert rel_inf(outs[k].iloc[:, 3:], z["Sig_inv"][k]) < TOL_MLE
        assert rel_inf(outs[k]["Sig_invMcoef"], z["Sig_invMcoef"][k]) < TOL_MLE
    mapped = pd.concat(outs, ignore_index=True)
    mr = api.dlsa_mapred(mapped)
    assert list(mr.columns) == list(z["mapred_columns"])
    assert rel_inf(mr["beta_byOLS"], z["beta_byOLS"]) < TOL_MLE
    assert rel_inf(mr["beta_byONESHOT"], z["beta_byONESHOT"]) < TOL_MLE
    assert rel_inf(mr.iloc[:, 2:], z["Sig_inv_sum"]) < TOL_MLE
    out = api.dlsa(Sig_inv_=mr.iloc[:, 2:], beta_=mr["beta_byOLS"], sample_size=X.shape[0], fit_intercept=icpt)
    assert list(out.columns) == ["beta_byAIC", "beta_byBIC"]
    by_aic, by_bic, _ = orc.dlsa(z["Sig_inv_sum"], z["beta_byOLS"], X.shape[0], fit_intercept=icpt)
    assert rel_inf(out["beta_byAIC"], by_aic) < 1e-8
    assert rel_inf(out["beta_byBIC"], by_bic) < 1e-8
    if icpt:
        # the beta_0 convention (the rpy2 line dlsa.py:97: beta0[idx] + b0[0]; the shipped Python crashes there, D2-D5): at the
        # unshrunk END of the path the selected vector must be the WLS estimate itself, intercept included -- whatever the
        # criterion picked -- and the reported intercept moves with the path by exactly -Sigma_00^-1 Sigma_0. (beta - beta_WLS)
        # (lsa.py:98-104: the intercept is profiled out of the quadratic)
        S, b = mr.iloc[:, 2:].to_numpy(), mr["beta_byOLS"].to_numpy()
        path = api.lars_lsa(S, b, True, X.shape[0])
        beta, beta0 = np.asarray(path["beta"]), np.asarray(path["beta0"])
        assert rel_inf(beta[-1], b[1:]) < 1e-8 and abs(beta0[-1]) < 1e-8 * max(1.0, abs(b[0]))
        prof = -(beta - b[1:][None, :]) @ S[0, 1:] / S[0, 0]
        assert np.max(np.abs(beta0 - prof)) < 1e-8 * max(1.0, np.max(np.abs(prof)))
        ib = int(np.argmin(np.asarray(path["BIC"])))
        assert abs(float(out["beta_byBIC"].iloc[0]) - (beta0[ib] + b[0])) < 1e-10 * max(1.0, abs(b[0]))


def test_tensor_fast_path_equals_frame_path(api, orc):
    n, p, K = 12000, 30, 6
    X, y = orc.synth_logistic(5, 0, n, p)
    Xd, yd = torch.from_numpy(X).cuda(), torch.from_numpy(y).cuda()
    for icpt in (False, True):
        mb = api.fit_logistic_partitions(Xd, yd, partition_num=K, fit_intercept=icpt)
        assert mb.status == [0] * K
        mr = api.dlsa_mapred(mb)
        frame = mb.to_frame()
        assert frame.shape == (K * (p + icpt), 3 + p + icpt)
        mr2 = api.dlsa_mapred(frame)
        assert rel_inf(mr2.to_numpy(), mr.to_numpy()) < 1e-12
        parts = orc.partition_rows(n, K)
        blocks = [orc.logistic_model_block(X[q], y[q], icpt) for q in parts]
        ols, oneshot, S = orc.dlsa_mapred_blocks([b[0] for b in blocks], [b[1] for b in blocks], [b[2] for b in blocks])
        assert rel_inf(mr["beta_byOLS"], ols) < TOL_MLE
        assert rel_inf(mr["beta_byONESHOT"], oneshot) < TOL_MLE


def test_lars_lsa_signature_and_golden(api):
    z = np.load(os.path.join(GOLDEN, "F3_lars_p50_lasso.npz"))
    r = api.lars_lsa(np.matrix(z["Sigma"]), z["b"], False, int(z["n"]), type="lasso")
    assert set(r) == {"AIC", "BIC", "beta", "beta0"}
    assert rel_inf(r["beta"], z["beta"]) < 1e-8 and rel_inf(r["BIC"], z["BIC"]) < 1e-8


def test_dummy_path_standardise_and_zero_block(api, orc):
    """models.py:56-104: level folding, one-hot, baseline drop, canonical column order,
    standardisation; :84-91 zero block when a level is missing from the chunk."""
    rng = np.random.default_rng(2)
    n = 4000
    df = pd.DataFrame({"partition_id": np.zeros(n), "label": 0.0,
                       "dist": rng.normal(5.0, 2.0, n), "dep": rng.normal(-1.0, 0.5, n),
                       "carrier": rng.choice(["AA", "BB", "CC", "ZZ"], n, p=[0.5, 0.3, 0.15, 0.05]),
                       "dow": rng.choice(["1", "2", "3"], n)})
    eta = 0.4 * (df["dist"] - 5) / 2 - 0.3 * (df["carrier"] == "BB") + 0.5 * (df["dow"] == "3")
    df["label"] = (rng.random(n) < 1 / (1 + np.exp(-eta))).astype(float)
    dummy_info = {"factor_selected": {"carrier": ["AA", "BB", "CC"], "dow": ["1", "2", "3"]},
                  "factor_dropped": {"carrier": ["ZZ"], "dow": []},
                  "factor_selected_names": {"carrier": ["carrier_AA", "carrier_BB", "carrier_CC", "carrier_000_OTHERS"],
                                            "dow": ["dow_1", "dow_2", "dow_3"]}}
    baseline = ["carrier_000_OTHERS", "dow_1"]
    data_info = df[["dist", "dep"]].describe()
    data_info = pd.DataFrame({c: [str(n), data_info[c]["mean"], data_info[c]["std"]] for c in ["dist", "dep"]})
    out = api.logistic_model(df, "label", fit_intercept=True, dummy_info=dummy_info,
                             dummy_factors_baseline=baseline, data_info=data_info)
    want = ["par_id", "coef", "Sig_invMcoef", "intercept", "dep", "dist",
            "carrier_AA", "carrier_BB", "carrier_CC", "dow_2", "dow_3"]
    assert list(out.columns) == want
    # oracle on the same design matrix
    Xo = np.column_stack([(df["dep"] - float(data_info["dep"][1])) / float(data_info["dep"][2]),
                          (df["dist"] - float(data_info["dist"][1])) / float(data_info["dist"][2]),
                          df["carrier"] == "AA", df["carrier"] == "BB", df["carrier"] == "CC",
                          df["dow"] == "2", df["dow"] == "3"]).astype(float)
    c, smc, sig = orc.logistic_model_block(Xo, df["label"].to_numpy(), True)
    assert rel_inf(out["coef"], c) < TOL_MLE and rel_inf(out.iloc[:, 3:], sig) < TOL_MLE
    # a chunk without any "CC" carrier -> all-zero block + warning
    sub = df[df["carrier"] != "CC"].reset_index(drop=True)
    with warnings.catch_warnings(record=True) as wlist:
        warnings.simplefilter("always")
        zero = api.logistic_model(sub, "label", fit_intercept=True, dummy_info=dummy_info,
                                  dummy_factors_baseline=baseline, data_info=data_info)
    assert any("missing in this data chunk" in str(w.message) for w in wlist)
    assert zero.shape == (8, 11) and float(np.abs(zero.to_numpy()).max()) == 0.0
    assert list(zero.columns) == want


def test_simulate_and_eval(api, orc):
    df = api.simulate_logistic(5000, 8, "systematic", 4, seed=77)
    assert list(df.columns) == ["partition_id", "label"] + ["x%d" % i for i in range(8)]
    assert df["partition_id"].tolist()[:6] == [0, 1, 2, 3, 0, 1]
    Xo, yo = orc.synth_logistic(77, 0, 5000, 8)
    assert np.array_equal(df.iloc[:, 2:].to_numpy(), Xo)
    with pytest.raises(Exception):
        api.simulate_logistic(10, 2, "random", 2)
    par = pd.DataFrame({"beta_byOLS": orc.true_beta(8), "beta_byONESHOT": np.zeros(8)})
    ev = api.logistic_model_eval(df, "label", par)
    ll = orc.logistic_loglik(df.iloc[:, 2:].to_numpy(), df["label"].to_numpy(), par.to_numpy())
    assert list(ev.columns) == ["beta_byOLS", "beta_byONESHOT"]
    assert rel_inf(ev.to_numpy()[0], ll) < 1e-12


def test_mapred_rejects_empty(api):
    empty = pd.DataFrame(columns=["par_id", "coef", "Sig_invMcoef", "x0", "x1"])
    with pytest.raises(Exception, match="Zero-length"):
        api.dlsa_mapred(empty)


class _FakeSparkFrame:
    """The slice of pyspark.sql.DataFrame that dlsa_mapred touches (dlsa.py:30-34,52), backed by pandas; counts what travels."""

    def __init__(self, pdf, partitions):
        self._pdf, self.columns, self.collected_rows = pdf, list(pdf.columns), []
        self.rdd = type("rdd", (), {"getNumPartitions": staticmethod(lambda: partitions)})()

    def groupby(self, key):
        outer = self

        class Grouped:
            def sum(self, *cols):
                g = outer._pdf.groupby(key, as_index=False)[list(cols)].sum()
                g.columns = [key] + ["sum(%s)" % c for c in cols]
                g = g.sample(frac=1.0, random_state=1)           # Spark returns the groups in no particular order
                child = _FakeSparkFrame(g, 1)
                child.collected_rows = outer.collected_rows
                return child
        return Grouped()

    def toPandas(self):
        self.collected_rows.append(len(self._pdf))
        return self._pdf.copy()


def test_mapred_on_a_spark_frame_sums_on_the_spark_side(api, orc):
    """A Spark DataFrame goes through the reference's own calls -- groupby('par_id').sum(*columns[1:]).toPandas() -- so p rows
    are collected, not K * p; the result equals the pandas route on the stacked frame, one-shot mean divided by
    rdd.getNumPartitions() (dlsa.py:30-34,51-52)."""
    X, y = orc.synth_logistic(3, 0, 9000, 12, orc.SYNTH_UNIFORM)
    frames = []
    for k in range(3):
        df = pd.DataFrame(X[k::3], columns=["x%d" % i for i in range(12)])
        df.insert(0, "label", y[k::3]); df.insert(0, "partition_id", k)
        frames.append(api.logistic_model(df, "label"))
    stacked = pd.concat(frames, ignore_index=True)
    sdf = _FakeSparkFrame(stacked, 3)
    out_spark = api.dlsa_mapred(sdf)
    out_pandas = api.dlsa_mapred(stacked, num_partitions=3)
    assert sdf.collected_rows == [12]                         # the grouped sums only
    assert list(out_spark.columns) == list(out_pandas.columns)
    assert rel_inf(out_spark.to_numpy(), out_pandas.to_numpy()) < 1e-12
    empty = _FakeSparkFrame(stacked.iloc[:0], 3)
    with pytest.raises(Exception, match="Zero-length"):
        api.dlsa_mapred(empty)


def test_driver_script_pickle_layout(api, orc, tmp_path):
    """projects/logistic_dlsa.py counterpart: call order map -> dlsa_mapred -> dlsa -> eval and the
    reference's result pickle [Sig_inv_beta, out_dlsa, out_par, out_model_eval, out_time] (:411)."""
    import pickle
    import subprocess
    import sys
    from conftest import ROOT
    pkl = str(tmp_path / "out.pkl")
    n, K, p = 20000, 4, 10
    subprocess.check_call([sys.executable, os.path.join(ROOT, "projects", "logistic_dlsa.py"), "--sample-size", str(n),
                           "--partition-num", str(K), "--p", str(p), "--seed", "123", "--save", pkl])
    Sig_inv_beta, out_dlsa, out_par, out_model_eval, out_time = pickle.load(open(pkl, "rb"))
    assert list(Sig_inv_beta.columns) == ["beta_byOLS", "beta_byONESHOT"] + ["x%d" % i for i in range(p)]
    assert list(out_par.columns) == ["beta_byAIC", "beta_byBIC", "beta_byOLS", "beta_byONESHOT"]
    assert list(out_model_eval.columns) == list(out_par.columns)
    for col in ("sample_size", "sample_size_per_partition", "n_par", "partition_num", "memsize_total",
                "time_repartition", "time_mapred", "time_dlsa", "time_model_fit", "time_model_eval"):
        assert col in out_time.columns
    X, y = orc.synth_logistic(123, 0, n, p)
    parts = orc.partition_rows(n, K)
    blocks = [orc.logistic_model_block(X[q], y[q]) for q in parts]
    ols, oneshot, S = orc.dlsa_mapred_blocks([b[0] for b in blocks], [b[1] for b in blocks], [b[2] for b in blocks])
    assert rel_inf(Sig_inv_beta["beta_byOLS"], ols) < TOL_MLE
    assert rel_inf(Sig_inv_beta["beta_byONESHOT"], oneshot) < TOL_MLE
    ll = orc.logistic_loglik(X, y, out_par.to_numpy())
    assert rel_inf(out_model_eval.to_numpy()[0], ll) < 1e-11


def test_linear_model_blocks_combine_to_global_ols(api, orc):
    """N3: linear-model map step; the WLS combine of the OLS blocks is the global OLS estimate."""
    rng = np.random.default_rng(4)
    n, p, K = 30000, 40, 6
    X = orc.synth_features(9, 0, n, p)
    beta = orc.true_beta(p)
    y = 0.5 + X @ beta + rng.standard_normal(n)
    Xd, yd = torch.from_numpy(X).cuda(), torch.from_numpy(y).cuda()
    mb = api.fit_linear_partitions(Xd, yd, partition_num=K, fit_intercept=True)
    assert mb.status == [0] * K
    mr = api.dlsa_mapred(mb)
    Xi = np.column_stack([np.ones(n), X])
    ols = np.linalg.lstsq(Xi, y, rcond=None)[0]
    assert rel_inf(mr["beta_byOLS"], ols) < 1e-10
    assert rel_inf(mr.iloc[:, 2:], Xi.T @ Xi) < 1e-12
    parts = orc.partition_rows(n, K)
    c0 = np.linalg.lstsq(Xi[parts[0]], y[parts[0]], rcond=None)[0]
    assert rel_inf(mb.coef[0].cpu().numpy(), c0) < 1e-10
    rss0 = float(np.sum((y[parts[0]] - Xi[parts[0]] @ c0) ** 2))
    assert abs(mb.loglik[0] - rss0) < 1e-8 * rss0
    df = pd.DataFrame(np.column_stack([np.zeros(n), y, X]), columns=["partition_id", "label"] + ["x%d" % i for i in range(p)])
    out = api.linear_model(df, "label", fit_intercept=True)
    assert list(out.columns)[:4] == ["par_id", "coef", "Sig_invMcoef", "intercept"]
    assert rel_inf(out["coef"], ols) < 1e-10
    sel = api.dlsa(mr.iloc[:, 2:], mr["beta_byOLS"], n, fit_intercept=True)
    assert sel.shape == (p + 1, 2)


def test_loglik_single_pass_many_columns(api, orc):
    from dlsa_amd import engine
    rng = np.random.default_rng(12)
    for (n, p, c) in [(3000, 7, 1), (5000, 130, 3), (4000, 500, 4), (3000, 300, 7), (1500, 1100, 8), (2000, 1500, 5)]:
        X, y = orc.synth_logistic(21, 0, n, p)
        par = rng.standard_normal((p, c)) * (2.0 / np.sqrt(p))
        out = engine.loglik(torch.from_numpy(X).cuda(), torch.from_numpy(y).cuda(), torch.from_numpy(par).cuda()).cpu().numpy()
        assert rel_inf(out, orc.logistic_loglik(X, y, par)) < 1e-12


def test_xtv(api, orc):
    from dlsa_amd import engine
    rng = np.random.default_rng(3)
    for (n, p) in [(1, 1), (1000, 5), (4099, 129), (3000, 500), (700, 1000)]:
        X = rng.random((n, p)) - 0.5
        v = rng.standard_normal(n)
        g, vv = engine.xtv(torch.from_numpy(X).cuda(), torch.from_numpy(v).cuda())
        assert rel_inf(g.cpu().numpy(), X.T @ v) < 1e-12
        assert abs(vv.item() - v @ v) <= 1e-12 * (v @ v)


def test_linear_model_fp32_rows(api, orc):
    """Config-5 shape in miniature: fp32 rows, wide p, fp32 Gram + fp64-accumulated X'y."""
    from dlsa_amd import engine
    rng = np.random.default_rng(6)
    n, p, K = 20000, 300, 2
    X = (rng.random((n, p)) - 0.5).astype(np.float32)
    y = (X.astype(np.float64) @ orc.true_beta(p) + rng.standard_normal(n)).astype(np.float32)
    Xd, yd = torch.from_numpy(X).cuda(), torch.from_numpy(y).cuda()
    g, vv = engine.xtv(Xd, yd)
    assert rel_inf(g.cpu().numpy(), X.astype(np.float64).T @ y.astype(np.float64)) < 1e-6
    assert abs(vv.item() - float(y.astype(np.float64) @ y.astype(np.float64))) < 1e-6 * vv.item()
    mb = api.fit_linear_partitions(Xd, yd, partition_num=K)
    mr = api.dlsa_mapred(mb)
    ols = np.linalg.lstsq(X.astype(np.float64), y.astype(np.float64), rcond=None)[0]
    assert rel_inf(mr["beta_byOLS"], ols) < 2e-3          # fp32 Gram accumulation
