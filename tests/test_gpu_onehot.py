"""GPU: the structured one-hot passes (gather / histogram on raw numerics + level codes) must reproduce the dense
kernels' results on the matrix the design kernel builds -- against the oracle, and against the reference's own
dummy-path outputs (fixture F4)."""
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


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _random_design(rng, n, q, nlevels, intercept=True, baseline=True):
    """column plan: [1] + q numerics + every level of every factor except level 0 (the baseline)"""
    kind, src, level, shift, scale = [], [], [], [], []
    if intercept:
        kind.append(0); src.append(0); level.append(0); shift.append(0.0); scale.append(1.0)
    for a in range(q):
        kind.append(1); src.append(a); level.append(0); shift.append(float(rng.normal())); scale.append(float(rng.uniform(0.5, 2)))
    level_col, nl = [], []
    for t, L in enumerate(nlevels):
        nl.append(L)
        for l in range(L):
            if baseline and l == 0:
                level_col.append(-1)
            else:
                level_col.append(len(kind))
                kind.append(2); src.append(t); level.append(l); shift.append(0.0); scale.append(1.0)
    p = len(kind)
    num = rng.normal(size=(n, q)) * 2 + 0.5
    codes = np.stack([rng.integers(0, L, n) for L in nlevels], 1).astype(np.int32) if nlevels else np.zeros((n, 0), np.int32)
    arr = lambda v, t: np.asarray(v, dtype=t)
    desc = (arr(kind, np.int32), arr(src, np.int32), arr(level, np.int32), arr(shift, np.float64), arr(scale, np.float64))
    return p, num, codes, desc, nl, level_col


def _plan(api, p, desc, nl, level_col):
    from dlsa_amd import engine
    kind, src, level, shift, scale = desc
    dense = [j for j in range(p) if kind[j] in (0, 1)]
    return engine.OnehotPlan(p, kind[dense], src[dense], shift[dense], scale[dense], dense, nl, level_col)


@pytest.mark.parametrize("n,q,nlevels", [(5000, 7, (11, 6, 20)), (3001, 2, (3,)), (20000, 0, (5, 4)), (777, 7, (40, 40, 9, 2)),
                                         (4000, 3, (110, 110, 20, 6)), (1, 1, (2,)), (6000, 2, (1400, 5)), (3000, 0, (700, 8, 3))])
def test_onehot_passes_match_dense_oracle(api, orc, n, q, nlevels):
    from dlsa_amd import engine
    rng = np.random.default_rng(n + q)
    p, num, codes, desc, nl, level_col = _random_design(rng, n, q, nlevels)
    codes[rng.integers(0, n, max(1, n // 50)), 0] = -1          # unknown level: no column
    plan = _plan(api, p, desc, nl, level_col)
    X, _ = orc.design_matrix(num, codes, *desc)
    beta = rng.normal(size=p) * 0.4
    y = (rng.random(n) < 0.5).astype(np.float64)
    wo, go, llo = orc.logit_pass(X, y, beta)
    w, g, ll = engine.onehot_logit_pass(plan, dev(num) if q else None, dev(codes), dev(y), dev(beta))
    assert rel_inf(w.cpu().numpy(), wo) < 1e-12
    assert rel_inf(g.cpu().numpy(), go) < 1e-11 and abs(ll.item() - llo) < 1e-11 * abs(llo)
    H = engine.onehot_gram(plan, dev(num) if q else None, dev(codes), dev(wo)).cpu().numpy()
    Ho = orc.gram(X, wo)
    assert np.max(np.abs(H - Ho)) < 1e-12 * np.max(np.abs(Ho))
    assert np.array_equal(H, H.T)
    if nlevels[:2] == (110, 110):
        assert plan.roles >= 2                                  # the 110 x 110 table gets a role of its own


@pytest.mark.parametrize("seed", range(20))
def test_onehot_passes_randomised_designs(api, orc, seed):
    """Seeded random designs: 0-7 numerics, 1-5 factors of 2-120 levels (a few rows with an unknown level), with and without
    intercept / baselines, 1-40000 rows -- structured logit pass and Gram against the oracle on the dense matrix."""
    from dlsa_amd import engine
    rng = np.random.default_rng(7000 + seed)
    n = int(rng.choice([rng.integers(1, 300), rng.integers(300, 6000), rng.integers(6000, 40000)]))
    q = int(rng.integers(0, 8))                                  # the structured plan holds at most 8 dense columns (1 + 7)
    nlevels = tuple(int(rng.choice([rng.integers(2, 8), rng.integers(8, 40), rng.integers(40, 121)])) for _ in range(int(rng.integers(1, 6))))    # (a pair table must fit LDS: <= ~140 x 140)
    p, num, codes, desc, nl, level_col = _random_design(rng, n, q, nlevels, intercept=bool(rng.random() < 0.7), baseline=bool(rng.random() < 0.7))
    if n > 10:
        codes[rng.integers(0, n, max(1, n // 40)), int(rng.integers(0, len(nlevels)))] = -1
    plan = _plan(api, p, desc, nl, level_col)
    X, _ = orc.design_matrix(num, codes, *desc)
    beta = rng.normal(size=p) * float(rng.choice([0.05, 0.4, 1.5]))
    y = (rng.random(n) < 0.5).astype(np.float64)
    wo, go, llo = orc.logit_pass(X, y, beta)
    w, g, ll = engine.onehot_logit_pass(plan, dev(num) if q else None, dev(codes), dev(y), dev(beta))
    assert rel_inf(w.cpu().numpy(), wo) < 1e-12, (n, q, nlevels)
    assert np.max(np.abs(g.cpu().numpy() - go)) < 1e-11 * max(1.0, np.max(np.abs(X).sum(0))) and abs(ll.item() - llo) < 1e-11 * abs(llo)
    H = engine.onehot_gram(plan, dev(num) if q else None, dev(codes), dev(wo)).cpu().numpy()
    Ho = orc.gram(X, wo)
    assert np.max(np.abs(H - Ho)) < 1e-12 * max(np.max(np.abs(Ho)), 1e-300), (n, q, nlevels)
    assert np.array_equal(H, H.T)


def test_onehot_passes_are_bit_reproducible(api):
    """Deterministic accumulation (the default): the Gram adds 64-bit fixed-point integers to its LDS tables (order-independent),
    the logit pass takes turns by wave, so the gradient, the Hessian and a whole structured fit come out bit-identical run
    after run -- with skewed levels (many lanes on one address), several table roles and enough rows for every workgroup to
    run many rounds."""
    from dlsa_amd import engine
    rng = np.random.default_rng(2026)
    n, q, nlevels = 600000, 7, (11, 6, 20, 110, 110)
    p, num, codes, desc, nl, level_col = _random_design(rng, n, q, nlevels)
    for t, L in enumerate(nlevels):                             # Zipf-like levels: hot cells
        pr = 1.0 / np.arange(1, L + 1); pr /= pr.sum()
        codes[:, t] = rng.choice(L, size=n, p=pr)
    plan = _plan(api, p, desc, nl, level_col)
    beta = rng.normal(size=p) * 0.2
    y = (rng.random(n) < 0.4).astype(np.float64)
    dn, dc, dy, db = dev(num), dev(codes), dev(y), dev(beta)
    w0, g0, ll0 = engine.onehot_logit_pass(plan, dn, dc, dy, db)
    H0 = engine.onehot_gram(plan, dn, dc, w0)
    for _ in range(6):
        w, g, ll = engine.onehot_logit_pass(plan, dn, dc, dy, db)
        assert torch.equal(g, g0) and torch.equal(w, w0) and ll.item() == ll0.item()
        assert torch.equal(engine.onehot_gram(plan, dn, dc, w0), H0)
    offs = [0, n // 3, n]
    r0 = engine.onehot_irls_fit(plan, dn, dc, dy, offs)
    r1 = engine.onehot_irls_fit(plan, dn, dc, dy, offs)
    assert torch.equal(r0["coef"], r1["coef"]) and torch.equal(r0["Sig_inv"], r1["Sig_inv"])


def test_onehot_gram_exact_mode_agrees_with_float_modes_and_falls_back_on_overflow(api, monkeypatch):
    """The default Gram accumulates 64-bit fixed-point integers in LDS (order-independent, hence bit-reproducible at the speed
    of the unordered adds).  It must agree with the ordered and the unordered floating-point modes to rounding, and an addend
    outside its range (|w d| > 16) must send the launch to the ordered mode instead of corrupting H."""
    from dlsa_amd import engine
    rng = np.random.default_rng(7)
    n, q, nlevels = 200000, 5, (9, 4, 30, 60)
    p, num, codes, desc, nl, level_col = _random_design(rng, n, q, nlevels)
    plan = _plan(api, p, desc, nl, level_col)
    w = rng.random(n) * 0.25
    dn, dc, dw = dev(num), dev(codes), dev(w)
    H2 = engine.onehot_gram(plan, dn, dc, dw)                     # exact (default)
    assert torch.equal(H2, engine.onehot_gram(plan, dn, dc, dw))
    with engine.kernel_options(onehot_ordered=1):
        H1 = engine.onehot_gram(plan, dn, dc, dw)
    with engine.kernel_options(onehot_ordered=0):
        H0 = engine.onehot_gram(plan, dn, dc, dw)
    scale = float(H1.abs().max())
    assert float((H2 - H1).abs().max()) < 1e-13 * scale and float((H0 - H1).abs().max()) < 1e-13 * scale
    d = H1.diagonal().clamp_min(1e-300).sqrt()
    assert float(((H2 - H1).abs() / (d[:, None] * d[None, :])).max()) < 1e-11       # small cells on their own scale
    # weights far above the fixed-point range: every addend overflows -> ordered fall-back, same matrix as the ordered mode
    big = dev(w * 1e6)
    Hb = engine.onehot_gram(plan, dn, dc, big)
    with engine.kernel_options(onehot_ordered=1):
        Hb1 = engine.onehot_gram(plan, dn, dc, big)
    assert torch.equal(Hb, Hb1) and bool(torch.isfinite(Hb).all())
    assert float((Hb - H1 * 1e6).abs().max()) < 1e-12 * float(Hb.abs().max())
    # caller weights of a tiny scale keep their RELATIVE accuracy (the fixed-point mode is absolute, 2^-40: w ~ 1e-8 would
    # keep five digits there, w < 4.5e-13 none -- the public entry sums a caller's weights in ordered floating point)
    for sc in (1e-8, 1e-14):
        Hs = engine.onehot_gram(plan, dn, dc, dev(w * sc))
        assert float((Hs - H1 * sc).abs().max()) < 1e-12 * float(H1.abs().max()) * sc
    # a NaN weight must surface as NaN (through the fall-back), not vanish
    wn = w.copy(); wn[12345] = np.nan
    assert not bool(torch.isfinite(engine.onehot_gram(plan, dn, dc, dev(wn))).all())


def test_onehot_plan_refuses_what_the_structured_path_cannot_hold(api):
    from dlsa_amd import engine, _lib
    rng = np.random.default_rng(0)
    p, num, codes, desc, nl, level_col = _random_design(rng, 10, 1, (2500,))
    with pytest.raises(_lib.DlsaError, match="bad p"):              # more columns than any kernel here holds (p <= 2048)
        _plan(api, p, desc, nl, level_col)
    p, num, codes, desc, nl, level_col = _random_design(rng, 10, 9, (3,))
    with pytest.raises(_lib.DlsaError, match="dense columns"):
        _plan(api, p, desc, nl, level_col)


@pytest.mark.parametrize("n,q,nlevels", [(30000, 2, (300, 300)), (20000, 0, (170, 160, 5)), (9000, 3, (700, 40))])
def test_onehot_pair_tables_beyond_lds_are_cut_into_row_bands(api, orc, n, q, nlevels):
    """A factor-pair table larger than the LDS budget (300 x 300 levels = 703 KB) is cut into bands of whole rows that go to
    different workgroup roles; passes and fit must equal the dense path on the design kernel's matrix."""
    from dlsa_amd import engine
    rng = np.random.default_rng(n)
    p, num, codes, desc, nl, level_col = _random_design(rng, n, q, nlevels)
    plan = _plan(api, p, desc, nl, level_col)
    assert plan.roles >= 2
    X, _ = orc.design_matrix(num, codes, *desc)
    w = rng.random(n) * 0.25
    H = engine.onehot_gram(plan, dev(num) if q else None, dev(codes), dev(w)).cpu().numpy()
    Ho = orc.gram(X, w)
    assert np.max(np.abs(H - Ho)) < 1e-12 * np.max(np.abs(Ho)) and np.array_equal(H, H.T)
    H2 = engine.onehot_gram(plan, dev(num) if q else None, dev(codes), dev(w)).cpu().numpy()
    assert np.array_equal(H, H2)                                       # exact accumulation: bit-reproducible


def test_structured_fit_matches_reference_dummy_path(api):
    """fit_logistic_design on raw numerics + codes == the reference's logistic_model on the dummy path (F4)."""
    z, df, dummy_info, baseline, data_info = load_f4()
    spec = api.DesignSpec.from_reference(list(df.columns), "label", True, dummy_info, baseline, data_info)
    assert spec.onehot_plan() is not None
    num, codes, unknown = spec.encode(df, dummy_info)
    assert not unknown and spec.missing_levels(codes) == []
    y = dev(z["label"])
    mb = api.fit_logistic_design(dev(num), dev(codes), y, spec)
    assert ["par_id", "coef", "Sig_invMcoef"] + mb.names == list(z["columns"]) and mb.status == [0]
    assert rel_inf(mb.coef[0].cpu().numpy(), z["coef_mle"]) < TOL_MLE
    assert rel_inf(mb.Sig_inv[0].cpu().numpy(), z["Sig_inv_mle"]) < TOL_MLE
    assert rel_inf(mb.Sig_invMcoef[0].cpu().numpy(), z["Sig_invMcoef_mle"]) < TOL_MLE
    dense = api.fit_logistic_design(dev(num), dev(codes), y, spec, structured=False)
    assert rel_inf(mb.coef[0].cpu().numpy(), dense.coef[0].cpu().numpy()) < 1e-11
    sub = df[df["carrier"] != "CC"].reset_index(drop=True)
    assert spec.missing_levels(spec.encode(sub, dummy_info)[1]) == ["carrier_CC"]


def test_structured_fit_many_partitions_matches_dense(api, orc):
    from dlsa_amd import engine
    rng = np.random.default_rng(11)
    n = 120_000
    p, num, codes, desc, nl, level_col = _random_design(rng, n, 4, (9, 5, 30))
    plan = _plan(api, p, desc, nl, level_col)
    X, _ = orc.design_matrix(num, codes, *desc)
    beta = rng.normal(size=p) * 0.3
    y = (rng.random(n) < 1 / (1 + np.exp(-X @ beta))).astype(np.float64)
    offs = [0, 30_000, 30_000, 75_000, n]                       # one empty partition
    r = engine.onehot_irls_fit(plan, dev(num), dev(codes), dev(y), offs)
    d = engine.irls_fit(dev(X), dev(y), offs)
    assert r["status"] == d["status"] == [0, 4, 0, 0]
    assert rel_inf(r["coef"].cpu().numpy(), d["coef"].cpu().numpy()) < 1e-10
    assert rel_inf(r["Sig_inv"].cpu().numpy(), d["Sig_inv"].cpu().numpy()) < 1e-10
    c, _, s = orc.logistic_model_block(X[:30_000], y[:30_000])
    assert rel_inf(r["coef"][0].cpu().numpy(), c) < TOL_MLE and rel_inf(r["Sig_inv"][0].cpu().numpy(), s) < TOL_MLE


@pytest.mark.parametrize("K", [1, 3, 7])
def test_structured_fit_strided_partitions_no_gather(api, orc, K):
    """partition_id = i % K (models.py:33) on the RAW representation: strided views of the numerics and the level codes go to
    dlsa_onehot_irls_fit_ex_f64 -- no gathered copy -- and give the oracle's per-partition MLE and Hessian on the rows i % K == k."""
    from dlsa_amd import engine
    rng = np.random.default_rng(100 + K)
    n = 90_001
    p, num, codes, desc, nl, level_col = _random_design(rng, n, 3, (7, 4, 12))
    plan = _plan(api, p, desc, nl, level_col)
    X, _ = orc.design_matrix(num, codes, *desc)
    beta = rng.normal(size=p) * 0.3
    y = (rng.random(n) < 1 / (1 + np.exp(-X @ beta))).astype(np.float64)
    dn, dc, dy = dev(num), dev(codes), dev(y)
    first, rows = list(range(K)), [len(range(k, n, K)) for k in range(K)]
    torch.cuda.synchronize(); torch.cuda.reset_peak_memory_stats()
    base = torch.cuda.memory_allocated()
    r = engine.onehot_irls_fit_ex(plan, dn, dc, dy, first, rows, row_step=K)
    torch.cuda.synchronize()
    assert torch.cuda.max_memory_allocated() - base < 0.5 * (num.nbytes + codes.nbytes) + 48e6      # scratch, not a copy of the rows
    assert r["status"] == [0] * K
    for k in range(K):
        c, smc, sig = orc.logistic_model_block(X[k::K], y[k::K])
        assert rel_inf(r["coef"][k].cpu().numpy(), c) < TOL_MLE
        assert rel_inf(r["Sig_inv"][k].cpu().numpy(), sig) < TOL_MLE
        assert rel_inf(r["Sig_invMcoef"][k].cpu().numpy(), smc) < TOL_MLE
    # contiguous partitions through the same entry (row_step = 1) equal dlsa_onehot_irls_fit_f64
    offs = [0, 20000, 20000, n]
    a = engine.onehot_irls_fit_ex(plan, dn, dc, dy, offs[:-1], [offs[i + 1] - offs[i] for i in range(3)], 1)
    b = engine.onehot_irls_fit(plan, dn, dc, dy, offs)
    assert a["status"] == b["status"] == [0, 4, 0] and torch.equal(a["coef"], b["coef"]) and torch.equal(a["Sig_inv"], b["Sig_inv"])
