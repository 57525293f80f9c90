"""GPU: edge cases of the C ABI -- empty / degenerate partitions, extreme shapes, error reporting, and the
invariance of the IRLS result under its acceleration switches."""
import ctypes
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def rel_inf(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return float(np.max(np.abs(a - b)) / max(1e-300, np.max(np.abs(b))))


@pytest.fixture(scope="module")
def eng():
    assert torch.cuda.is_available()
    from dlsa_amd import engine
    return engine


@pytest.fixture(scope="module")
def orc():
    from oracle import dlsa_oracle
    return dlsa_oracle


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def test_empty_partition_gives_zero_block_and_status_empty(eng, orc):
    """models.py:84-91: a partition that cannot be fitted contributes an all-zero block."""
    X, y = orc.synth_logistic(3, 0, 3000, 6, orc.SYNTH_UNIFORM)
    r = eng.irls_fit(dev(X), dev(y), [0, 1000, 1000, 3000])
    assert r["status"] == [0, 4, 0]
    assert float(r["Sig_inv"][1].abs().max()) == 0.0 and float(r["coef"][1].abs().max()) == 0.0
    c0, _, s0 = orc.logistic_model_block(X[:1000], y[:1000])
    c2, _, s2 = orc.logistic_model_block(X[1000:], y[1000:])
    assert rel_inf(r["coef"][0].cpu().numpy(), c0) < 1e-10 and rel_inf(r["coef"][2].cpu().numpy(), c2) < 1e-10
    msg = eng.sum_blocks(r["coef"], r["Sig_invMcoef"], r["Sig_inv"]).cpu().numpy()
    assert rel_inf(msg[:36].reshape(6, 6), s0 + s2) < 1e-10


def test_collinear_design_reports_not_spd(eng, orc):
    X, y = orc.synth_logistic(4, 0, 2000, 5, orc.SYNTH_UNIFORM)
    X = np.column_stack([X, X[:, 0]])                     # duplicated column: singular Hessian
    r = eng.irls_fit(dev(X), dev(y), [0, 2000])
    assert r["status"][0] == 2 and r["rc"] == 4             # DLSA_PART_NOT_SPD / DLSA_ERR_NOT_SPD
    from dlsa_amd import _lib
    assert "positive definite" in _lib.last_error()


def test_nan_rows_report_nan_status(eng, orc):
    X, y = orc.synth_logistic(5, 0, 1000, 4, orc.SYNTH_UNIFORM)
    X[17, 2] = np.nan
    r = eng.irls_fit(dev(X), dev(y), [0, 1000])
    assert r["status"][0] == 3 and r["rc"] == 6


@pytest.mark.parametrize("n,p", [(0, 7), (1, 1), (1, 500), (3, 2048), (40, 2047)])
def test_gram_extreme_shapes(eng, n, p):
    rng = np.random.default_rng(n + p)
    X = rng.random((n, p)) - 0.5
    w = rng.random(n)
    H = eng.gram(dev(X) if n else torch.empty((0, p), dtype=torch.float64, device="cuda"),
                 dev(w) if n else torch.empty((0,), dtype=torch.float64, device="cuda")).cpu().numpy()
    Ho = X.T @ (w[:, None] * X)
    assert H.shape == (p, p)
    assert np.max(np.abs(H - Ho)) <= 1e-12 * max(1.0, np.max(np.abs(Ho)))
    assert np.array_equal(H, H.T)


def test_logit_pass_p_limit_and_invalid_arguments(eng):
    from dlsa_amd import _lib
    lib = _lib.load()
    X = torch.zeros((4, 2049), dtype=torch.float64, device="cuda")
    y = torch.zeros(4, dtype=torch.float64, device="cuda")
    with pytest.raises(_lib.DlsaError):
        eng.logit_pass(X, y, torch.zeros(2049, dtype=torch.float64, device="cuda"))
    # workspace too small -> DLSA_ERR_WORKSPACE (3) with the needed size in the message
    Xs = torch.zeros((1000, 64), dtype=torch.float64, device="cuda")
    H = torch.zeros((64, 64), dtype=torch.float64, device="cuda")
    ws = torch.zeros(256, dtype=torch.uint8, device="cuda")
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    rc = lib.dlsa_gram_f64(P(Xs), 64, None, 1000, 64, P(H), 64, 0, P(ws), 256, None)
    assert rc == 3 and "workspace" in _lib.last_error()
    rc = lib.dlsa_gram_f64(P(Xs), 32, None, 1000, 64, P(H), 64, 0, P(ws), 256, None)      # ldx < p
    assert rc == 1 and "bad shape" in _lib.last_error()


def test_fits_at_fused_widths_write_no_weights_and_agree_with_the_weighted_form(eng, orc, monkeypatch):
    """49 <= p <= 120: every Hessian of a fit, the closing one too, comes from the fused pass, so no pass writes the weight vector
    (DLSA_IRLS_LEAN, default on).  Same MLE / Hessian / loglik as the form that keeps the weights, as the oracle, and as a run whose
    switches force the cases that need weights again (inherited factors)."""
    n, p = 150_000, 100
    X, y = eng.synth(123, 0, n, p, kind=eng.SYNTH_GAUSSIAN)
    offs = [0, 70_001, n]
    lean = eng.irls_fit(X, y, offs)
    assert lean["status"] == [0, 0]
    monkeypatch.setenv("DLSA_IRLS_LEAN", "0")
    kept = eng.irls_fit(X, y, offs)
    monkeypatch.delenv("DLSA_IRLS_LEAN")
    with eng.irls_options(lean=False):                        # the same switch as an option field of the C ABI
        kept_opt = eng.irls_fit(X, y, offs)
    assert torch.equal(kept_opt["coef"], kept["coef"]) and torch.equal(kept_opt["Sig_inv"], kept["Sig_inv"])
    monkeypatch.setenv("DLSA_IRLS_INHERIT", "1")              # stand-in Hessians read the weights: lean switches itself off
    inh = eng.irls_fit(X, y, offs)
    monkeypatch.delenv("DLSA_IRLS_INHERIT")
    monkeypatch.setenv("DLSA_IRLS_FUSE_LAST", "0")            # the run ends on a logit pass: the closing Hessian is a fused pass of its own
    nolast = eng.irls_fit(X, y, offs)
    for other in (kept, inh, nolast):
        assert other["status"] == [0, 0]
        assert rel_inf(lean["coef"].cpu().numpy(), other["coef"].cpu().numpy()) < 1e-11
        assert rel_inf(lean["Sig_inv"].cpu().numpy(), other["Sig_inv"].cpu().numpy()) < 1e-11
        assert np.allclose(lean["loglik"], other["loglik"], rtol=1e-11)
    c, _, sg = orc.logistic_model_block(X[:70_001].cpu().numpy(), y[:70_001].cpu().numpy())
    assert rel_inf(lean["coef"][0].cpu().numpy(), c) < 1e-10 and rel_inf(lean["Sig_inv"][0].cpu().numpy(), sg) < 1e-10


def test_irls_result_does_not_depend_on_acceleration_switches(eng, orc):
    """Subsample warm start, frozen / inherited Cholesky factors, secant corrections and partition warm starts only change the
    path of the iteration: the MLE and the Hessian at the MLE must agree to the solver tolerance."""
    n, p = 240_000, 200                                     # p >= 192: factor inheritance is active
    X, y = eng.synth(99, 0, n, p, kind=eng.SYNTH_GAUSSIAN)
    offs = [0, 60_000, 120_000, n]                          # last partition large enough for the subsample start
    base = eng.irls_fit(X, y, offs)
    assert base["status"] == [0, 0, 0]
    keys = ("DLSA_IRLS_SUBSAMPLE", "DLSA_IRLS_FREEZE", "DLSA_IRLS_WARM", "DLSA_IRLS_INHERIT", "DLSA_IRLS_SECANT",
            "DLSA_IRLS_INVERSE", "DLSA_IRLS_POOL", "DLSA_IRLS_PREDICT")
    try:
        for k in keys:
            os.environ[k] = "0"
        plain = eng.irls_fit(X, y, offs)
    finally:
        for k in keys:
            os.environ.pop(k, None)
    assert plain["status"] == [0, 0, 0]
    assert base["n_iter"] != plain["n_iter"]                # the switches really changed the path
    assert rel_inf(base["coef"].cpu().numpy(), plain["coef"].cpu().numpy()) < 1e-11
    assert rel_inf(base["Sig_inv"].cpu().numpy(), plain["Sig_inv"].cpu().numpy()) < 1e-11
    # and both are the oracle's MLE on the first partition
    c, _, s = orc.logistic_model_block(X[:60_000].cpu().numpy(), y[:60_000].cpu().numpy())
    assert rel_inf(base["coef"][0].cpu().numpy(), c) < 1e-10 and rel_inf(base["Sig_inv"][0].cpu().numpy(), s) < 1e-10
    # the same switches through the C ABI's options struct (dlsa_irls_set_options, per thread): the path of the environment run,
    # bit for bit -- and an option field wins over the environment variable of the same switch
    off = eng.IrlsOptions(subsample_div=0, freeze_at=0.0, warm=False, inherit=False, secant=False, inverse=False, pool=False, predict=False)
    with eng.irls_options(off):
        opt = eng.irls_fit(X, y, offs)
    assert opt["n_iter"] == plain["n_iter"] and torch.equal(opt["coef"], plain["coef"]) and torch.equal(opt["Sig_inv"], plain["Sig_inv"])
    after = eng.irls_fit(X, y, offs)                       # the block's options are gone
    assert after["n_iter"] == base["n_iter"] and torch.equal(after["coef"], base["coef"])
    os.environ["DLSA_IRLS_SECANT"] = "0"
    try:
        with eng.irls_options(secant=True):
            both = eng.irls_fit(X, y, offs)
    finally:
        os.environ.pop("DLSA_IRLS_SECANT", None)
    assert both["n_iter"] == base["n_iter"] and torch.equal(both["coef"], base["coef"])


def test_fit_logistic_partitions_takes_the_driver_policy_as_keyword_arguments(eng, orc):
    """SURVEY 5 asked for keyword arguments + one small dataclass: chains / fused / predict ... on the operator-level entry."""
    import dlsa_amd
    n, p, K = 120_000, 64, 6
    X, y = eng.synth(5, 0, n, p, kind=eng.SYNTH_GAUSSIAN)
    a = dlsa_amd.fit_logistic_partitions(X, y, partition_num=K)
    b = dlsa_amd.fit_logistic_partitions(X, y, partition_num=K, chains=1, fused=False, predict=False, small=False)
    c = dlsa_amd.fit_logistic_partitions(X, y, partition_num=K, options=eng.IrlsOptions(chains=3, seeded=True))
    assert a.status == b.status == c.status == [0] * K
    for other in (b, c):
        assert rel_inf(other.coef.cpu().numpy(), a.coef.cpu().numpy()) < 1e-10
        assert rel_inf(other.Sig_inv.cpu().numpy(), a.Sig_inv.cpu().numpy()) < 1e-10
    with pytest.raises(TypeError):
        dlsa_amd.fit_logistic_partitions(X, y, partition_num=K, chain=2)          # not a field of IrlsOptions
    from dlsa_amd import _lib
    with pytest.raises(RuntimeError):
        with eng.irls_options(chains=9):
            pass
    assert "chains" in _lib.last_error()
