"""Pin the CPU oracle (oracle/dlsa_oracle.py) against golden vectors produced by RUNNING
the reference (oracle/make_golden.py, committed under tests/golden/).  CPU-only."""
import glob
import os

import numpy as np
import pytest

from oracle import dlsa_oracle as orc
from conftest import GOLDEN
from golden_inputs import lars_case

TOL_MLE = 1e-10      # oracle vs reference driven to the exact MLE (tol=1e-15 shim)
TOL_SHIPPED = 2e-2   # oracle vs reference as shipped (sklearn tol=1e-4): sanity tier only


def rel_inf(a, b):
    return np.max(np.abs(a - b)) / max(1e-300, np.max(np.abs(b)))


def _inputs(z, name):
    if name.startswith("synth"):
        return orc.synth_logistic(int(z["seed"]), 0, int(z["n"]), int(z["p"]), orc.SYNTH_UNIFORM)
    g = np.load(os.path.join(GOLDEN, "games_expand_input.npz"))
    return g["X"].astype(np.float64), g["y"].astype(np.float64)


F1 = sorted(os.path.basename(f)[3:-8] for f in glob.glob(os.path.join(GOLDEN, "F1_*_mle.npz")))


@pytest.mark.parametrize("name", F1)
def test_map_step_matches_reference_mle(name):
    z = np.load(os.path.join(GOLDEN, "F1_%s_mle.npz" % name))
    X, y = _inputs(z, name)
    K, icpt = int(z["K"]), bool(z["fit_intercept"])
    parts = orc.partition_rows(X.shape[0], K)
    coefs, smcs, sigs = [], [], []
    for k in range(K):
        c, smc, sig = orc.logistic_model_block(X[parts[k]], y[parts[k]], icpt)
        coefs.append(c); smcs.append(smc); sigs.append(sig)
        assert rel_inf(c, z["coef"][k]) < TOL_MLE
        assert rel_inf(sig, z["Sig_inv"][k]) < TOL_MLE
        assert rel_inf(smc, z["Sig_invMcoef"][k]) < TOL_MLE
    ols, oneshot, S = orc.dlsa_mapred_blocks(coefs, smcs, sigs)
    assert rel_inf(ols, z["beta_byOLS"]) < TOL_MLE
    assert rel_inf(oneshot, z["beta_byONESHOT"]) < TOL_MLE
    assert rel_inf(S, z["Sig_inv_sum"]) < TOL_MLE


F1_SHIPPED = sorted(os.path.basename(f)[3:-12] for f in glob.glob(os.path.join(GOLDEN, "F1_*_shipped.npz")))


@pytest.mark.parametrize("name", F1_SHIPPED)
def test_map_step_close_to_reference_as_shipped(name):
    z = np.load(os.path.join(GOLDEN, "F1_%s_shipped.npz" % name))
    X, y = _inputs(z, name)
    K, icpt = int(z["K"]), bool(z["fit_intercept"])
    parts = orc.partition_rows(X.shape[0], K)
    for k in range(K):
        c, _, sig = orc.logistic_model_block(X[parts[k]], y[parts[k]], icpt)
        assert rel_inf(c, z["coef"][k]) < TOL_SHIPPED
        assert rel_inf(sig, z["Sig_inv"][k]) < TOL_SHIPPED


def test_output_columns_contract():
    z = np.load(os.path.join(GOLDEN, "F1_synth_n2000_p5_K4_icpt_mle.npz"))
    assert list(z["columns"][:4]) == ["par_id", "coef", "Sig_invMcoef", "intercept"]
    assert list(z["mapred_columns"][:3]) == ["beta_byOLS", "beta_byONESHOT", "intercept"]


def test_mapred_spd_blocks():
    z = np.load(os.path.join(GOLDEN, "F2_spd_K3_p4.npz"))
    ols, oneshot, S = orc.dlsa_mapred_blocks(z["coef"], z["Sig_invMcoef"], z["Sig_inv"])
    assert rel_inf(ols, z["beta_byOLS"]) < 1e-12
    assert rel_inf(oneshot, z["beta_byONESHOT"]) < 1e-14
    assert rel_inf(S, z["Sig_inv_sum"]) < 1e-14


F3 = sorted(os.path.basename(f)[:-4] for f in glob.glob(os.path.join(GOLDEN, "F3_*.npz")))


@pytest.mark.parametrize("name", F3)
def test_lars_lsa_path_matches_reference(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    typ = "lasso" if name.endswith("lasso") else "lar"
    S, b, n = lars_case(z)
    r = orc.lars_lsa(S, b, False, n, type=typ)
    if "beta" in z.files:
        assert r["beta"].shape == z["beta"].shape
        assert rel_inf(r["beta"], z["beta"]) < 1e-9
    else:                                   # p = 250: every 10th path row is stored
        assert r["beta"].shape[0] - 1 == int(z["steps"])
        assert rel_inf(r["beta"][z["beta_rows"]], z["beta_sub"]) < 1e-9
    assert rel_inf(r["AIC"], z["AIC"]) < 1e-9
    assert rel_inf(r["BIC"], z["BIC"]) < 1e-9
    assert np.all(r["beta0"] == 0)


F3I = sorted(os.path.basename(f)[:-4] for f in glob.glob(os.path.join(GOLDEN, "F3i_*.npz")))


@pytest.mark.parametrize("name", F3I)
def test_lars_lsa_intercept_branch_matches_reference_called_with_n_equal_p(name):
    """lsa.py:98-104,194-204: the reference's intercept branch indexes with n (defect D4) and therefore runs exactly when
    n = p; its beta / beta0 / AIC / BIC for that call pin the oracle's intercept algebra (BIC then carries log(p))."""
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    typ = "lasso" if name.endswith("lasso") else "lar"
    r = orc.lars_lsa(z["Sigma"], z["b"], True, int(z["n"]), type=typ)
    assert r["beta"].shape == z["beta"].shape and r["beta"].shape[1] == z["Sigma"].shape[0] - 1
    assert rel_inf(r["beta"], z["beta"]) < 1e-9
    assert np.max(np.abs(r["beta0"] - z["beta0"])) < 1e-9 * max(1.0, np.max(np.abs(z["beta0"])))
    assert rel_inf(r["AIC"], z["AIC"]) < 1e-9
    assert rel_inf(r["BIC"], z["BIC"]) < 1e-9


@pytest.mark.parametrize("case", ["zero", "dup", "both"])
def test_mapred_rank_deficient_blocks_give_the_min_norm_solution(case):
    """dlsa.py:48-49: lstsq(rcond=None) on a singular sum of blocks returns the minimum-norm solution."""
    z = np.load(os.path.join(GOLDEN, "F2r_rankdef_%s_K3_p6.npz" % case))
    ols, oneshot, S = orc.dlsa_mapred_blocks(z["coef"], z["Sig_invMcoef"], z["Sig_inv"])
    assert np.linalg.matrix_rank(S) == int(z["rank"]) < S.shape[0]
    assert rel_inf(ols, z["beta_byOLS"]) < 1e-10
    assert rel_inf(oneshot, z["beta_byONESHOT"]) < 1e-14
    assert rel_inf(S, z["Sig_inv_sum"]) < 1e-14


def test_lars_intercept_algebra():
    """No runnable reference for intercept=True (D4/D5).  Check the Schur-complement algebra:
    with the full path end point the intercept branch must reproduce the WLS estimate."""
    rng = np.random.default_rng(5)
    p, n = 8, 500
    X = np.column_stack([np.ones(n), rng.random((n, p - 1)) - 0.5])
    S = X.T @ (rng.random(n)[:, None] * 0.25 * X)
    b = rng.standard_normal(p)
    by_aic, by_bic, fit = orc.dlsa(S, b, n, fit_intercept=True)
    # last path point = unpenalised minimiser of the quadratic = b itself
    assert np.allclose(fit["beta"][-1], b[1:], rtol=0, atol=1e-10)
    assert abs(fit["beta0"][-1] + b[0] - b[0]) < 1e-10
    # every path point's intercept minimises the quadratic given the slopes
    for k in range(fit["beta"].shape[0]):
        th1 = fit["beta"][k]
        th0 = fit["beta0"][k] + b[0]
        grad0 = S[0, 0] * (th0 - b[0]) + S[0, 1:] @ (th1 - b[1:])
        assert abs(grad0) < 1e-8 * S[0, 0]


def test_philox_known_answer():
    """Random123 known-answer vectors for Philox-4x32-10."""
    z = np.zeros(1, np.uint32)
    r = orc.philox4x32_10(z, z, z, z, 0, 0)
    assert [int(v[0]) for v in r] == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]
    f = z + np.uint32(0xFFFFFFFF)
    r = orc.philox4x32_10(f, f, f, f, 0xFFFFFFFF, 0xFFFFFFFF)
    assert [int(v[0]) for v in r] == [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]
    a = orc.philox4x32_10(z + np.uint32(0x243F6A88), z + np.uint32(0x85A308D3), z + np.uint32(0x13198A2E),
                          z + np.uint32(0x03707344), 0xA4093822, 0x299F31D0)
    assert [int(v[0]) for v in a] == [0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]


def test_synth_rows_are_pure_functions_of_index():
    X, y = orc.synth_logistic(7, 0, 300, 9)
    X2, y2 = orc.synth_logistic(7, 100, 50, 9)
    assert np.array_equal(X[100:150], X2) and np.array_equal(y[100:150], y2)
    assert X.min() >= -0.5 and X.max() < 0.5
    G = orc.synth_features(7, 0, 20000, 4, orc.SYNTH_GAUSSIAN)
    assert abs(G.std() - (1 / 12) ** 0.5) < 0.01 and abs(G.mean()) < 0.01


def test_oracle_fp32_linear_stream_is_a_function_of_seed_and_row():
    """oracle.synth_linear32 (the restatement of csrc/synth.hip's fp32-native linear stream): rows depend on (seed, row index) only,
    the features have the stated moments, the response is X beta* + sigma N(0, 1)."""
    from oracle import dlsa_oracle as orc
    X, y = orc.synth_linear32(11, 1000, 6000, 10, sigma=0.5)
    Xa, ya = orc.synth_linear32(11, 1000, 2500, 10, sigma=0.5)
    Xb, yb = orc.synth_linear32(11, 3500, 3500, 10, sigma=0.5)
    assert X.dtype == np.float32 and y.dtype == np.float32
    assert np.array_equal(np.vstack([Xa, Xb]), X) and np.array_equal(np.concatenate([ya, yb]), y)
    assert not np.array_equal(orc.synth_linear32(12, 1000, 100, 10)[0], X[:100])
    Xl, yl = orc.synth_linear32(5, 0, 200000, 6, sigma=2.0)
    assert abs(float(Xl.mean())) < 3e-3 and abs(float(Xl.var()) - 1.0 / 12.0) < 2e-3
    C = np.corrcoef(Xl.astype(np.float64), rowvar=False)
    assert np.max(np.abs(C - np.eye(6))) < 0.01
    r = yl.astype(np.float64) - Xl.astype(np.float64) @ orc.true_beta(6)
    assert abs(r.mean()) < 0.02 and abs(r.std() - 2.0) < 0.02
