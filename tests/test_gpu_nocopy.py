"""GPU: the map step without copies of the shard -- implicit intercept (the ones column of models.py:121-122 is never
materialised) and strided partitions (partition_id = i % K, models.py:33, as a view) -- against the oracle at small size,
and at BASELINE config 3's per-GPU size (2.5e7 x 500 fp64 = 100 GB) under a peak-memory bound."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

TOL_MLE = 1e-10


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


@pytest.mark.parametrize("n,p", [(3000, 7), (5001, 100), (20000, 127), (9000, 500), (70000, 499)])
def test_implicit_intercept_passes_match_oracle_on_the_materialised_column(eng, orc, n, p):
    X, y = orc.synth_logistic(41 + p, 0, n, p, orc.SYNTH_GAUSSIAN)
    X1 = np.column_stack([np.ones(n), X])
    rng = np.random.default_rng(p)
    beta = rng.normal(size=p + 1) * 0.2
    wo, go, llo = orc.logit_pass(X1, y, beta)
    w, g, ll = eng.logit_pass(dev(X), dev(y), dev(beta), fit_intercept=True)
    assert rel_inf(w.cpu().numpy(), wo) < 1e-12 and rel_inf(g.cpu().numpy(), go) < 1e-11
    assert abs(float(ll) - llo) < 1e-12 * abs(llo)
    H = eng.gram_icpt(dev(X), w)
    assert torch.equal(H, H.T)
    assert rel_inf(H.cpu().numpy(), orc.gram(X1, wo)) < 1e-12
    H1 = eng.gram_icpt(dev(X), None).cpu().numpy()                 # w = 1: the linear model's [1 | X]'[1 | X]
    assert H1[0, 0] == n and rel_inf(H1, X1.T @ X1) < 1e-12
    par = rng.normal(size=(p + 1, 5)) * 0.1
    assert rel_inf(eng.loglik(dev(X), dev(y), dev(par), fit_intercept=True).cpu().numpy(), orc.logistic_loglik(X1, y, par)) < 1e-12


@pytest.mark.parametrize("K,icpt,p", [(1, True, 12), (4, False, 12), (5, True, 12), (3, True, 100), (7, True, 33)])
def test_strided_partitions_and_implicit_intercept_match_oracle(eng, orc, K, icpt, p):
    """partition_id = i % K as a strided view + implicit intercept = the reference's logistic_model on each partition."""
    import dlsa_amd
    n = 4000 * K + 3
    X, y = orc.synth_logistic(51 + K, 0, n, p)
    mb = dlsa_amd.fit_logistic_partitions(dev(X), dev(y), partition_num=K, fit_intercept=icpt)
    assert mb.status == [0] * K and mb.coef.shape == (K, p + int(icpt))
    assert mb.names[0] == ("intercept" if icpt else "x0")
    parts = orc.partition_rows(n, K)
    for k in range(K):
        c, smc, sig = orc.logistic_model_block(X[parts[k]], y[parts[k]], icpt)
        assert rel_inf(mb.coef[k].cpu().numpy(), c) < TOL_MLE
        assert rel_inf(mb.Sig_inv[k].cpu().numpy(), sig) < TOL_MLE
        assert rel_inf(mb.Sig_invMcoef[k].cpu().numpy(), smc) < TOL_MLE
    # contiguous partitions through the same entry (row_step = 1) and ragged / empty ones
    offs = [0, 1500, 1500, n]
    r = eng.irls_fit_ex(dev(X), dev(y), offs[:-1], [offs[i + 1] - offs[i] for i in range(3)], 1, fit_intercept=icpt)
    assert r["status"] == [0, 4, 0]
    c, _, sig = orc.logistic_model_block(X[1500:], y[1500:], icpt)
    assert rel_inf(r["coef"][2].cpu().numpy(), c) < TOL_MLE and rel_inf(r["Sig_inv"][2].cpu().numpy(), sig) < TOL_MLE


def test_config3_scale_fit_without_copies_stays_under_130GB(eng):
    """BASELINE config 3's per-GPU shard, the reference-faithful call: 2.5e7 x 500 rows, partition_id = i % 25, fit_intercept.
    No gathered copy, no [1 | X] copy: peak device memory must stay under 130 GB (the shard itself is 100 GB), and the result
    must equal the fit of the same partitions laid out contiguously WITH a materialised ones column to 1e-11."""
    import dlsa_amd
    from conftest import need_hbm
    need_hbm(140e9)            # shard 100 GB + fit scratch < 30 GB + one gathered 4 GB partition: asserted, never skipped on a whole MI355X
    n, p, K = 25_000_000, 500, 25
    X, y = eng.synth(20260101, 0, n, p, kind=eng.SYNTH_GAUSSIAN)
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    base = torch.cuda.memory_allocated()
    mb = dlsa_amd.fit_logistic_partitions(X, y, partition_num=K, fit_intercept=True)
    out = dlsa_amd.dlsa_mapred(mb)
    torch.cuda.synchronize()
    peak = torch.cuda.max_memory_allocated()
    assert mb.status == [0] * K
    assert peak < 130e9, "peak %.1f GB" % (peak / 1e9)
    assert peak - base < 30e9
    # reference layout for three of the partitions: gathered rows + materialised ones column
    eng.release_workspace()
    for k in (0, 11, 24):
        Xk = eng.with_ones_column(X[k::K])
        r = eng.irls_fit(Xk, y[k::K].contiguous(), [0, Xk.shape[0]])
        assert r["status"] == [0]
        for key, got in (("coef", mb.coef[k]), ("Sig_inv", mb.Sig_inv[k]), ("Sig_invMcoef", mb.Sig_invMcoef[k])):
            err = float((got - r[key][0]).abs().max()) / float(r[key][0].abs().max())
            assert err < 1e-11, (k, key, err)
        del Xk, r
    assert np.isfinite(out.to_numpy()).all()
    # the combined estimate recovers the generating coefficients (intercept 0, first 200 slopes 1)
    truth = np.concatenate([[0.0], np.ones(200), np.zeros(300)])
    assert float(np.max(np.abs(out["beta_byOLS"].to_numpy() - truth))) < 0.02


@pytest.mark.parametrize("K,nk,p,icpt", [(20, 5000, 50, False), (20, 5000, 50, True), (6, 3001, 63, True), (3, 700, 5, False),
                                         (40, 1000, 16, True), (2, 65536, 64, False)])
def test_many_small_partitions_one_launch_matches_oracle(eng, orc, monkeypatch, K, nk, p, icpt):
    """Config 1's shape (20 partitions of 5 000 x 50, projects/logistic_dlsa.py:89-92) and relatives: every partition is
    fitted by its own workgroup in ONE launch (irls_small.hip).  Same exact MLE / Hessian as the oracle, and as the
    host-driven path (DLSA_IRLS_SMALL=0)."""
    import dlsa_amd
    n = K * nk
    X, y = orc.synth_logistic(61 + p, 0, n, p)
    mb = dlsa_amd.fit_logistic_partitions(dev(X), dev(y), partition_num=K, fit_intercept=icpt)
    assert mb.status == [0] * K
    parts = orc.partition_rows(n, K)
    for k in range(0, K, max(1, K // 5)):
        c, smc, sig = orc.logistic_model_block(X[parts[k]], y[parts[k]], icpt)
        assert rel_inf(mb.coef[k].cpu().numpy(), c) < TOL_MLE
        assert rel_inf(mb.Sig_inv[k].cpu().numpy(), sig) < TOL_MLE
        assert rel_inf(mb.Sig_invMcoef[k].cpu().numpy(), smc) < TOL_MLE
    assert max(mb.n_iter) <= 12 and min(mb.n_iter) >= 3
    monkeypatch.setenv("DLSA_IRLS_SMALL", "0")
    ref = dlsa_amd.fit_logistic_partitions(dev(X), dev(y), partition_num=K, fit_intercept=icpt)
    assert float((ref.coef - mb.coef).abs().max()) < 1e-11 * float(ref.coef.abs().max())
    assert float((ref.Sig_inv - mb.Sig_inv).abs().max()) < 1e-11 * float(ref.Sig_inv.abs().max())
    assert np.allclose(ref.loglik, mb.loglik, rtol=1e-10)


@pytest.mark.parametrize("C", [1, 3, 8, 16])
def test_small_partition_clusters_give_the_single_workgroup_result(eng, orc, monkeypatch, C):
    """With fewer partitions than CUs, C workgroups share a partition (irls_small.hip: partial sums through global scratch, a
    bounded per-partition barrier, workgroup 0 decides).  Any C gives the MLE / Hessian of the oracle; the strided, implicit-
    intercept call and ragged partitions (one with fewer row batches than workgroups, an empty one) included."""
    import dlsa_amd
    monkeypatch.setenv("DLSA_IRLS_SMALL_CLUSTER", str(C))
    monkeypatch.setenv("DLSA_IRLS_SMALL", "2")
    K, nk, p = 7, 4003, 37
    X, y = orc.synth_logistic(83, 0, K * nk, p)
    mb = dlsa_amd.fit_logistic_partitions(dev(X), dev(y), partition_num=K, fit_intercept=True)
    assert eng.irls_last_fit_path() == eng.IRLS_PATH_SMALL and mb.status == [0] * K
    parts = orc.partition_rows(K * nk, K)
    for k in (0, 3, 6):
        c, smc, sig = orc.logistic_model_block(X[parts[k]], y[parts[k]], True)
        assert rel_inf(mb.coef[k].cpu().numpy(), c) < TOL_MLE
        assert rel_inf(mb.Sig_inv[k].cpu().numpy(), sig) < TOL_MLE
        assert rel_inf(mb.Sig_invMcoef[k].cpu().numpy(), smc) < TOL_MLE
    if C == 3:                                                     # the cluster size as an option field instead of the environment
        monkeypatch.delenv("DLSA_IRLS_SMALL_CLUSTER")
        with eng.irls_options(small_cluster=3):
            mo = dlsa_amd.fit_logistic_partitions(dev(X), dev(y), partition_num=K, fit_intercept=True)
        assert torch.equal(mo.coef, mb.coef) and torch.equal(mo.Sig_inv, mb.Sig_inv)
        monkeypatch.setenv("DLSA_IRLS_SMALL_CLUSTER", "3")
    offs = [0, 9000, 9040, 9040, 15000, 21000]                     # 9000 rows, 40 rows (separable or not: status only), empty, 5960, 6000
    r = eng.irls_fit(dev(X[:21000]), dev(y[:21000]), offs)
    assert r["status"][0] == 0 and r["status"][2] == 4 and r["status"][3] == 0 and r["status"][4] == 0
    c0, _, s0 = orc.logistic_model_block(X[:9000], y[:9000])
    assert rel_inf(r["coef"][0].cpu().numpy(), c0) < TOL_MLE and rel_inf(r["Sig_inv"][0].cpu().numpy(), s0) < TOL_MLE
    c4, _, s4 = orc.logistic_model_block(X[15000:21000], y[15000:21000])
    assert rel_inf(r["coef"][4].cpu().numpy(), c4) < TOL_MLE and rel_inf(r["Sig_inv"][4].cpu().numpy(), s4) < TOL_MLE


def test_small_partition_cluster_barrier_timeout_falls_back(eng, orc, monkeypatch):
    """The cluster barrier is bounded: with a timeout of one tick every wait is too long, the launch is given up (nothing written) and
    rerun with one workgroup per partition -- same results, nothing hangs."""
    import dlsa_amd
    from dlsa_amd import _lib
    lib = _lib.load()
    monkeypatch.setenv("DLSA_IRLS_SMALL", "2")
    K, nk, p = 5, 6000, 20
    X, y = orc.synth_logistic(85, 0, K * nk, p)
    before = lib.dlsa_irls_small_cluster_timeout(1e-9)
    try:
        mb = dlsa_amd.fit_logistic_partitions(dev(X), dev(y), partition_num=K)
        after = lib.dlsa_irls_small_cluster_timeout(0.0)
    finally:
        lib.dlsa_irls_small_cluster_timeout(0.0)
    assert after > before and mb.status == [0] * K
    parts = orc.partition_rows(K * nk, K)
    c, smc, sig = orc.logistic_model_block(X[parts[2]], y[parts[2]])
    assert rel_inf(mb.coef[2].cpu().numpy(), c) < TOL_MLE and rel_inf(mb.Sig_inv[2].cpu().numpy(), sig) < TOL_MLE
    mb2 = dlsa_amd.fit_logistic_partitions(dev(X), dev(y), partition_num=K)          # default timeout: the clusters complete
    assert lib.dlsa_irls_small_cluster_timeout(0.0) == after
    assert float((mb2.coef - mb.coef).abs().max()) < 1e-12 * float(mb.coef.abs().max())


def test_small_partition_kernel_soft_failures(eng, orc):
    """Empty, collinear, NaN and separable partitions in one launch: per-partition statuses as the host-driven path reports them."""
    X, y = orc.synth_logistic(71, 0, 4000, 6)
    X = X.copy(); y = y.copy()
    X[1000:2000, 5] = X[1000:2000, 0]                 # partition 1: duplicated column -> singular Hessian
    X[2500, 2] = np.nan                               # partition 2: a NaN row
    r = eng.irls_fit(dev(X), dev(y), [0, 1000, 2000, 3000, 3000, 4000])
    assert r["status"][0] == 0 and r["status"][3] == 4 and r["status"][4] == 0
    assert r["status"][1] in (2, 3) and r["status"][2] == 3
    c0, _, s0 = orc.logistic_model_block(X[:1000], y[:1000])
    assert rel_inf(r["coef"][0].cpu().numpy(), c0) < TOL_MLE and rel_inf(r["Sig_inv"][0].cpu().numpy(), s0) < TOL_MLE
    assert float(r["Sig_inv"][3].abs().max()) == 0.0
    Xs = np.tile(np.array([[-2.0], [-1.0], [1.0], [2.0]]), (16, 1))
    ys = (Xs[:, 0] > 0).astype(np.float64)
    r = eng.irls_fit(dev(Xs), dev(ys), [0, 32, 64], max_iter=25)
    assert all(s in (1, 2, 3) for s in r["status"])  # perfectly separable: no finite MLE, must not claim OK
