"""GPU: the lock-step map step for many partitions of a narrow design (csrc/irls_batch.hip): all partitions of a call fitted together,
one launch per stage of a Newton iteration -- against the host-driven chained path, the oracle (dlsa/models.py:110-131 restated)
and itself."""
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


@pytest.mark.parametrize("p,sizes", [(64, [3000, 70000, 5000, 24577, 4100, 33000, 2500, 9000, 3100, 12000]),
                                     (100, [20000] * 12 + [50001, 2000]),
                                     (50, [2048, 4096, 100000, 3000, 7000, 2100, 2200, 2300]),
                                     (99, [20000] * 9 + [33001]), (101, [9000, 30000, 12000, 8192, 15000, 21000, 9100, 9200]),
                                     (111, [10000] * 8), (51, [4000, 9000, 70000, 3000, 5000, 2500, 2600, 2700])])
def test_lock_step_fit_equals_chained_fit_and_oracle(eng, orc, p, sizes):
    n, K = sum(sizes), len(sizes)
    X, y = eng.synth(4242 + p, 0, n, p, kind=eng.SYNTH_GAUSSIAN)
    offs = np.concatenate([[0], np.cumsum(sizes)]).tolist()
    with eng.irls_options(batched=True, small=False):
        b = eng.irls_fit(X, y, offs)
        assert eng.irls_last_fit_path() == eng.IRLS_PATH_BATCHED
        b2 = eng.irls_fit(X, y, offs)
    with eng.irls_options(batched=False, small=False):
        c = eng.irls_fit(X, y, offs)
        assert eng.irls_last_fit_path() == eng.IRLS_PATH_CHAINS
    assert b["status"] == c["status"] == [0] * K
    for key in ("coef", "Sig_inv", "Sig_invMcoef"):
        assert torch.equal(b[key], b2[key]), key                                   # bit-reproducible
        assert rel_inf(b[key].cpu().numpy(), c[key].cpu().numpy()) < 1e-10, key
    assert rel_inf(b["loglik"], c["loglik"]) < 1e-12
    S = b["Sig_inv"].cpu().numpy()
    assert np.array_equal(S, np.swapaxes(S, 1, 2))                                 # exactly symmetric
    for k in (0, K - 1):                                                           # and the oracle's MLE / Hessian
        Xk, yk = X[offs[k]:offs[k + 1]].cpu().numpy(), y[offs[k]:offs[k + 1]].cpu().numpy()
        co, smc, sig = orc.logistic_model_block(Xk, yk)
        assert rel_inf(b["coef"][k].cpu().numpy(), co) < 1e-10 and rel_inf(S[k], sig) < 1e-10
        assert rel_inf(b["Sig_invMcoef"][k].cpu().numpy(), smc) < 1e-10
    # the operator-level entry takes the same route (contiguous partitions, no intercept)
    import dlsa_amd
    with eng.irls_options(batched=True, small=False):
        mb = dlsa_amd.fit_logistic_partitions(X, y, part_offsets=offs)
    assert eng.irls_last_fit_path() == eng.IRLS_PATH_BATCHED and rel_inf(mb.coef.cpu().numpy(), b["coef"].cpu().numpy()) == 0.0


def test_lock_step_is_chosen_for_many_small_partitions_only(eng):
    p = 64
    X, y = eng.synth(7, 0, 64 * 3000, p, kind=eng.SYNTH_GAUSSIAN)
    offs = [3000 * k for k in range(65)]
    with eng.irls_options(small=False):
        r = eng.irls_fit(X, y, offs)
        assert eng.irls_last_fit_path() == eng.IRLS_PATH_BATCHED and r["status"] == [0] * 64
    r = eng.irls_fit(X, y, [0, 96000, 192000])                    # two large partitions: the chained path
    assert eng.irls_last_fit_path() == eng.IRLS_PATH_CHAINS and r["status"] == [0, 0]
    Xo, yo = eng.synth(8, 0, 40000, 40, kind=eng.SYNTH_GAUSSIAN)          # too narrow: not the fused class
    with eng.irls_options(batched=True, small=False):
        eng.irls_fit(Xo, yo, [4000 * k for k in range(11)])
        assert eng.irls_last_fit_path() == eng.IRLS_PATH_CHAINS
    Xp = eng.synth(9, 0, 40000, 64, kind=eng.SYNTH_GAUSSIAN)[0][:, :63]   # odd width in PADDED rows (pitch 64): the bytes behind a row are not data
    with eng.irls_options(batched=True, small=False):
        eng.irls_fit(Xp, yo, [4000 * k for k in range(11)])
        assert eng.irls_last_fit_path() == eng.IRLS_PATH_CHAINS


def test_lock_step_partitions_end_independently(eng, orc):
    """a perfectly separated partition ends NOT_CONVERGED / not finite without holding up or disturbing the others"""
    p, sizes = 50, [3000, 3000, 3000, 3000]
    X, y = eng.synth(99, 0, sum(sizes), p, kind=eng.SYNTH_GAUSSIAN)
    y = y.clone()
    y[3000:6000] = (X[3000:6000, 0] > 0).double()                           # partition 1: separable by the first column
    offs = [0, 3000, 6000, 9000, 12000]
    with eng.irls_options(batched=True, small=False):
        b = eng.irls_fit(X, y, offs, max_iter=25)
    assert b["status"][1] != 0 and [b["status"][k] for k in (0, 2, 3)] == [0, 0, 0]
    for k in (0, 2, 3):
        co, _, sig = orc.logistic_model_block(X[offs[k]:offs[k + 1]].cpu().numpy(), y[offs[k]:offs[k + 1]].cpu().numpy())
        assert rel_inf(b["coef"][k].cpu().numpy(), co) < 1e-10 and rel_inf(b["Sig_inv"][k].cpu().numpy(), sig) < 1e-10


@pytest.mark.parametrize("p,K,n,icpt,strided", [(100, 12, 150001, True, False), (100, 10, 100000, True, True), (64, 16, 96000, False, True),
                                                  (50, 7, 70007, True, True), (110, 6, 60000, True, False)])
def test_lock_step_with_intercept_and_strided_partitions(eng, orc, p, K, n, icpt, strided):
    """the reference-faithful call: partition_id = i % K (models.py:33: the strided views X[k::K]) and fit_intercept (logistic_dlsa.py:79)
    -- the fused kernel carries the intercept as a ones column in its LDS stages, the slabs walk the rows with pitch ldx K"""
    import dlsa_amd
    X, y = eng.synth(3100 + p + K, 0, n, p, kind=eng.SYNTH_GAUSSIAN)
    kw = dict(partition_num=K) if strided else dict(part_offsets=[int(n * k / K) for k in range(K + 1)])
    b = dlsa_amd.fit_logistic_partitions(X, y, fit_intercept=icpt, batched=True, small=False, **kw)
    assert eng.irls_last_fit_path() == eng.IRLS_PATH_BATCHED
    c = dlsa_amd.fit_logistic_partitions(X, y, fit_intercept=icpt, batched=False, small=False, **kw)
    assert eng.irls_last_fit_path() == eng.IRLS_PATH_CHAINS
    assert b.status == c.status == [0] * K and b.names == c.names
    for key in ("coef", "Sig_inv", "Sig_invMcoef"):
        assert rel_inf(getattr(b, key).cpu().numpy(), getattr(c, key).cpu().numpy()) < 1e-10, key
    k = K - 1
    rows = np.arange(k, n, K) if strided else np.arange(int(n * k / K), n)
    co, smc, sig = orc.logistic_model_block(X.cpu().numpy()[rows], y.cpu().numpy()[rows], icpt)
    assert rel_inf(b.coef[k].cpu().numpy(), co) < 1e-10 and rel_inf(b.Sig_inv[k].cpu().numpy(), sig) < 1e-10
    assert rel_inf(b.Sig_invMcoef[k].cpu().numpy(), smc) < 1e-10
    out = dlsa_amd.dlsa_mapred(b)                                   # and the reduce takes the blocks as they are
    assert list(out.columns[:2]) == ["beta_byOLS", "beta_byONESHOT"] and out.shape == (p + int(icpt), 2 + p + int(icpt))


def test_max_iter_means_full_row_iterations_on_every_driver(eng):
    """the lock-step driver counted its subsample phase against the caller's max_iter and reported it in n_iter; the
    chained driver counts full-row iterations only.  Partitions long enough for the subsample phase, a small max_iter that the
    full-row phase just meets: the same status on both drivers, and n_iter of the same meaning (full-row iterations)."""
    p, sizes = 80, [40000] * 10
    n, K = sum(sizes), len(sizes)
    X, y = eng.synth(77, 0, n, p, kind=eng.SYNTH_GAUSSIAN)
    offs = np.concatenate([[0], np.cumsum(sizes)]).tolist()
    with eng.irls_options(batched=True, small=False):
        ref = eng.irls_fit(X, y, offs)                  # generous max_iter: how many full-row iterations the lock step needs
        assert eng.irls_last_fit_path() == eng.IRLS_PATH_BATCHED and ref["status"] == [0] * K
    need = max(ref["n_iter"])
    assert need <= 8, ref["n_iter"]                     # (full-row iterations only: the subsample phase took the first ones)
    with eng.irls_options(batched=True, small=False):
        b = eng.irls_fit(X, y, offs, max_iter=need)
    assert b["status"] == [0] * K and max(b["n_iter"]) <= need      # (before: the subsample phase's iterations used the budget up)
    with eng.irls_options(batched=True, small=False):
        b12 = eng.irls_fit(X, y, offs, max_iter=12)
    with eng.irls_options(batched=False, small=False):
        c12 = eng.irls_fit(X, y, offs, max_iter=12)       # (the chains take more, cheaper iterations: reused factors contract linearly)
    assert b12["status"] == c12["status"] == [0] * K
    with eng.irls_options(batched=True, small=False):
        short = eng.irls_fit(X, y, offs, max_iter=1)     # one full-row iteration cannot converge: NOT_CONVERGED on every partition, n_iter = 1
    assert all(s != 0 for s in short["status"]) and short["n_iter"] == [1] * K


@pytest.mark.parametrize("p,K,nk,icpt,tol,strided", [(100, 40, 20000, False, 1e-13, False), (64, 24, 30000, True, 1e-10, False),
                                                        (99, 40, 25000, False, 1e-13, False), (80, 32, 20000, True, 1e-13, True),
                                                        (50, 48, 9000, False, 1e-10, True)])
def test_pooled_start_changes_the_path_not_the_result(eng, orc, p, K, nk, icpt, tol, strided):
    """round 5: the full-row iterations of a lock-step call start from ONE fit on a few leading rows of all partitions together
    (dlsa_irls_options.pooled_start), followed by gradient-only passes whose steps use the pooled Hessian (grad_passes), instead
    of every partition's own subsample MLE: fewer Newton passes, the same MLEs and Hessians (models.py:110-131 per partition),
    also against the oracle -- contiguous partitions and the reference's i % K (models.py:33), with and without the intercept."""
    import dlsa_amd
    n = K * nk
    X, y = eng.synth(9100 + p, 0, n, p, kind=eng.SYNTH_GAUSSIAN)
    part = dict(partition_num=K) if strided else dict(part_offsets=[k * nk for k in range(K + 1)])
    res = {}
    for name, opt in (("pooled+grad", dict(pooled_start=True)), ("pooled", dict(pooled_start=True, grad_passes=0)), ("own", dict(pooled_start=False))):
        res[name] = dlsa_amd.fit_logistic_partitions(X, y, fit_intercept=icpt, tol=tol, batched=True, small=False, **part, **opt)
        assert eng.irls_last_fit_path() == eng.IRLS_PATH_BATCHED and res[name].status == [0] * K
    a, b, c = res["pooled+grad"], res["pooled"], res["own"]
    for key in ("coef", "Sig_inv", "Sig_invMcoef"):
        assert rel_inf(getattr(a, key).cpu().numpy(), getattr(c, key).cpu().numpy()) < 1e-10, key
        assert rel_inf(getattr(b, key).cpu().numpy(), getattr(c, key).cpu().numpy()) < 1e-10, key
    assert max(a.n_iter) < max(b.n_iter) <= max(c.n_iter), (a.n_iter, b.n_iter, c.n_iter)     # Newton passes: the gradient-only ones are not counted
    assert sorted(a.n_iter)[K // 2] <= 3 and max(a.n_iter) <= 4             # (three for nearly every partition: the phase's end is a prediction)
    k = K // 2
    rows = np.arange(k, n, K) if strided else np.arange(k * nk, (k + 1) * nk)
    co, smc, sig = orc.logistic_model_block(X.cpu().numpy()[rows], y.cpu().numpy()[rows], icpt)
    assert rel_inf(a.coef[k].cpu().numpy(), co) < 1e-10 and rel_inf(a.Sig_inv[k].cpu().numpy(), sig) < 1e-10
    assert rel_inf(a.Sig_invMcoef[k].cpu().numpy(), smc) < 1e-10


def test_pooled_fit_that_fails_falls_back_to_the_subsample_start(eng, orc):
    """the pooled sample -- the leading rows of every partition -- is perfectly separable (the partitions themselves are not): the
    pooled fit ends without a finite MLE and the call goes on as before round 5, to the same MLEs (oracle)"""
    p, K, nk = 64, 24, 24000
    X, y = eng.synth(616, 0, K * nk, p, kind=eng.SYNTH_GAUSSIAN)
    y = y.clone()
    lead = max(256, ((max(8 * nk, 1000 * p) + K - 1) // K + 31) // 32 * 32)        # irls_batch.hip's pooled rows per partition
    for k in range(K):
        y[k * nk:k * nk + lead] = (X[k * nk:k * nk + lead, 0] > 0).double()
    offs = [k * nk for k in range(K + 1)]
    with eng.irls_options(batched=True, small=False):
        b = eng.irls_fit(X, y, offs)
        assert eng.irls_last_fit_path() == eng.IRLS_PATH_BATCHED
    assert b["status"] == [0] * K
    for k in (0, K - 1):
        co, _, sig = orc.logistic_model_block(X[offs[k]:offs[k + 1]].cpu().numpy(), y[offs[k]:offs[k + 1]].cpu().numpy())
        assert rel_inf(b["coef"][k].cpu().numpy(), co) < 1e-10 and rel_inf(b["Sig_inv"][k].cpu().numpy(), sig) < 1e-10


def test_pooled_start_with_partitions_that_differ(eng, orc):
    """partitions drawn from DIFFERENT coefficient vectors (rows not exchangeable: the pooled estimate is nobody's MLE), one of them
    perfectly separable: the pooled start costs iterations, never the result -- every partition ends at its own MLE (oracle)"""
    p, K, nk = 64, 24, 24000
    X, _ = eng.synth(515, 0, K * nk, p, kind=eng.SYNTH_GAUSSIAN)
    g = torch.Generator(device="cpu").manual_seed(5)
    y = torch.empty(K * nk, dtype=torch.float64, device=X.device)
    for k in range(K):
        bk = (torch.randn(p, generator=g, dtype=torch.float64) * (0.2 + 0.15 * k) * (-1.0) ** k).to(X.device)
        eta = X[k * nk:(k + 1) * nk] @ bk
        u = torch.rand(nk, generator=g, dtype=torch.float64).to(X.device)
        y[k * nk:(k + 1) * nk] = (u < torch.sigmoid(eta)).double()
    y[5 * nk:6 * nk] = (X[5 * nk:6 * nk, 3] > 0).double()
    X = X.clone()
    X[2 * nk:3 * nk] *= 3.0                          # ... and different scales: the pooled Hessian is 9x off for partition 2, 16x for 7
    X[7 * nk:8 * nk] *= 0.25
    offs = [k * nk for k in range(K + 1)]
    with eng.irls_options(batched=True, small=False, pooled_start=True, trace=True):
        b = eng.irls_fit(X, y, offs, max_iter=30)
        assert eng.irls_last_fit_path() == eng.IRLS_PATH_BATCHED
    assert b["status"][5] != 0 and [s for k, s in enumerate(b["status"]) if k != 5] == [0] * (K - 1)
    for k in (0, 2, 4, 7, 11):
        co, _, sig = orc.logistic_model_block(X[offs[k]:offs[k + 1]].cpu().numpy(), y[offs[k]:offs[k + 1]].cpu().numpy())
        assert rel_inf(b["coef"][k].cpu().numpy(), co) < 1e-10 and rel_inf(b["Sig_inv"][k].cpu().numpy(), sig) < 1e-10


def test_pooled_start_with_row_counts_that_differ_by_100x(eng, orc):
    """the gradient-only passes step with (rows_k / rows_pool) x the pooled Hessian (irls_batch.hip `pool_scale`): partitions of
    3 000 and 300 000 rows in ONE call scale it by factors 100x apart; every partition still ends at its own MLE (oracle), and
    the lock step is the driver that ran"""
    p = 64
    rows = [3000, 300000, 3000, 150000, 6000, 300000, 3100, 30000, 3000, 299999, 4000, 3000]
    K = len(rows)
    offs = [0]
    for r in rows:
        offs.append(offs[-1] + r)
    X, y = eng.synth(717, 0, offs[-1], p, kind=eng.SYNTH_GAUSSIAN)
    with eng.irls_options(batched=True, small=False, pooled_start=True):
        b = eng.irls_fit(X, y, offs)
        assert eng.irls_last_fit_path() == eng.IRLS_PATH_BATCHED
    with eng.irls_options(batched=False, small=False):
        c = eng.irls_fit(X, y, offs)
    assert b["status"] == [0] * K and c["status"] == [0] * K
    assert rel_inf(b["coef"].cpu().numpy(), c["coef"].cpu().numpy()) < 1e-10
    assert rel_inf(b["Sig_inv"].cpu().numpy(), c["Sig_inv"].cpu().numpy()) < 1e-10
    for k in (0, 1, 4, 9):
        co, smc, sig = orc.logistic_model_block(X[offs[k]:offs[k + 1]].cpu().numpy(), y[offs[k]:offs[k + 1]].cpu().numpy())
        assert rel_inf(b["coef"][k].cpu().numpy(), co) < 1e-10 and rel_inf(b["Sig_inv"][k].cpu().numpy(), sig) < 1e-10
        assert rel_inf(b["Sig_invMcoef"][k].cpu().numpy(), smc) < 1e-10
