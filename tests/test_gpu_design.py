"""GPU: dlsa_design_f64/f32 (through the C ABI) against the oracle, and the dummy / standardise path of
logistic_model / logistic_model_eval against the reference's own outputs (fixture F4)."""
import warnings

import numpy as np
import pytest

from f4_fixture import load_f4

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


def _random_spec(rng, q, f, p, nlev=7):
    kind = rng.integers(0, 3, p).astype(np.int32)
    if q == 0:
        kind[kind == 1] = 2
    if f == 0:
        kind[kind == 2] = 0
    src = np.where(kind == 1, rng.integers(0, max(q, 1), p), rng.integers(0, max(f, 1), p)).astype(np.int32)
    level = rng.integers(0, nlev, p).astype(np.int32)
    shift = np.where(kind == 1, rng.normal(size=p), 0.0)
    scale = np.where(kind == 1, rng.uniform(0.5, 3.0, p), 1.0)
    return kind, src, level, shift, scale


@pytest.mark.parametrize("n,q,f,p", [(1, 1, 1, 1), (1000, 3, 2, 11), (777, 7, 5, 260), (300, 4, 3, 256),
                                     (129, 2, 6, 700), (65, 5, 0, 33), (200, 0, 4, 64), (50, 9, 9, 2048)])
def test_design_kernel_bit_exact(api, orc, n, q, f, p):
    from dlsa_amd import engine
    rng = np.random.default_rng(n * 31 + p)
    num = rng.normal(size=(n, q)) * 10
    codes = rng.integers(-1, 7, (n, f)).astype(np.int32)
    kind, src, level, shift, scale = _random_spec(rng, q, f, p)
    Xo, seen_o = orc.design_matrix(num, codes, kind, src, level, shift, scale)
    d = lambda a: torch.from_numpy(a).cuda()
    X, seen = engine.design(d(num) if q else None, d(codes) if f else None, d(kind), d(src), d(level), d(shift), d(scale))
    assert np.array_equal(X.cpu().numpy(), Xo)                  # IEEE subtract + divide: bit-identical
    assert np.array_equal(seen.cpu().numpy(), seen_o)
    X32, _ = engine.design(d(num.astype(np.float32)) if q else None, d(codes) if f else None, d(kind), d(src),
                           d(level), d(shift), d(scale), dtype=torch.float32)
    Xo32, _ = orc.design_matrix(num.astype(np.float32).astype(np.float64), codes, kind, src, level, shift, scale)
    assert np.array_equal(X32.cpu().numpy(), Xo32.astype(np.float32))


def test_design_into_strided_output(api, orc):
    from dlsa_amd import engine
    rng = np.random.default_rng(5)
    n, q, f, p = 100, 2, 2, 9
    num, codes = rng.normal(size=(n, q)), rng.integers(0, 4, (n, f)).astype(np.int32)
    kind, src, level, shift, scale = _random_spec(rng, q, f, p, nlev=4)
    d = lambda a: torch.from_numpy(a).cuda()
    big = torch.full((n, 16), -7.0, dtype=torch.float64, device="cuda")
    engine.design(d(num), d(codes), d(kind), d(src), d(level), d(shift), d(scale), out=big[:, :p])
    Xo, _ = orc.design_matrix(num, codes, kind, src, level, shift, scale)
    assert np.array_equal(big[:, :p].cpu().numpy(), Xo) and float(big[:, p:].min()) == -7.0 == float(big[:, p:].max())


def test_logistic_model_dummy_path_matches_reference(api):
    z, df, dummy_info, baseline, data_info = load_f4()
    out = api.logistic_model(df, "label", fit_intercept=True, dummy_info=dummy_info,
                             dummy_factors_baseline=baseline, data_info=data_info)
    assert list(out.columns) == list(z["columns"])
    assert out["par_id"].tolist() == list(range(9))
    assert rel_inf(out["coef"], z["coef_mle"]) < TOL_MLE                 # north-star tolerance
    assert rel_inf(out["Sig_invMcoef"], z["Sig_invMcoef_mle"]) < TOL_MLE
    assert rel_inf(out.iloc[:, 3:], z["Sig_inv_mle"]) < TOL_MLE
    assert rel_inf(out["coef"], z["coef_shipped"]) < 2e-2                # the reference as shipped (tol=1e-4)


def test_logistic_model_zero_block_when_level_missing(api):
    z, df, dummy_info, baseline, data_info = load_f4()
    sub = df[df["carrier"] != "CC"].reset_index(drop=True)
    with warnings.catch_warnings(record=True) as wl:
        warnings.simplefilter("always")
        zero = api.logistic_model(sub, "label", fit_intercept=True, dummy_info=dummy_info,
                                  dummy_factors_baseline=baseline, data_info=data_info)
    assert any("missing in this data chunk" in str(w.message) and "carrier_CC" in str(w.message) for w in wl)
    assert list(zero.columns) == list(z["zero_columns"]) and zero.shape == tuple(z["zero_shape"])
    assert float(np.abs(zero.to_numpy(dtype=float)).max()) == 0.0


def test_logistic_model_eval_dummy_path_matches_reference(api):
    import pandas as pd
    z, df, dummy_info, baseline, data_info = load_f4()
    coef = z["coef_mle"]
    par = pd.DataFrame({"beta_byOLS": coef, "beta_half": 0.5 * coef, "beta_zero": 0.0 * coef})
    ll = api.logistic_model_eval(df, "label", par, fit_intercept=True, dummy_info=dummy_info,
                                 dummy_factors_baseline=baseline, data_info=data_info)
    assert list(ll.columns) == list(z["eval_columns"])
    assert rel_inf(ll.to_numpy().ravel(), z["eval_loglik"]) < TOL_MLE


def test_design_matrix_fast_path_from_device_codes(api, orc):
    """Tensor fast path (no pandas): device-resident codes -> X -> fit; equals the frame path."""
    z, df, dummy_info, baseline, data_info = load_f4()
    spec = api.DesignSpec.from_reference(list(df.columns), "label", True, dummy_info, baseline, data_info)
    num, codes, _ = spec.encode(df, dummy_info)
    X, missing = api.design_matrix(torch.from_numpy(num).cuda(), torch.from_numpy(codes).cuda(), spec)
    assert missing == [] and X.shape == (len(df), 9)
    mb = api.fit_logistic_partitions(X, torch.from_numpy(z["label"]).cuda(), partition_num=1, names=spec.names)
    assert rel_inf(mb.coef[0].cpu().numpy(), z["coef_mle"]) < TOL_MLE
