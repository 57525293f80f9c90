"""GPU: the wide Newton pass (csrc/irls_wide.hip, dlsa_newton_wide_pass_f64) -- the logit pass of a wide design (dlsa/models.py:110-114)
that also yields the partition's own Hessian (models.py:130) from bf16 products -- against the oracle: w / g / loglik to the logit
pass's tolerance, the reduced-precision Hessian as a PRECONDITIONER (its spectrum against the fp64 Hessian); and the fits that use
it for their Newton steps against fits that do not and against the oracle's MLE (the fixed point is unchanged)."""
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


def spectrum_of(Happrox, H):
    """eigenvalues of H^-1/2 Happrox H^-1/2: 1 +- what the preconditioned iteration contracts by"""
    L = np.linalg.cholesky(H)
    M = np.linalg.solve(L, np.linalg.solve(L, Happrox).T)
    ev = np.linalg.eigvalsh((M + M.T) / 2)
    return float(ev[0]), float(ev[-1])


# every column-block count of the bf16 Gram kernel (4 .. 16 blocks of 32, one or two workgroups per slab), odd widths, ragged rows
# (not a multiple of the 16-row chunk), with and without the implicit intercept
@pytest.mark.parametrize("p,n,icpt", [(121, 40000, False), (128, 33000, True), (160, 50001, False), (192, 40000, True), (200, 36007, False),
                                      (256, 40000, True), (257, 33333, False), (289, 40000, False), (320, 32768, True), (351, 45000, False),
                                      (384, 40010, False), (400, 40000, True), (449, 38000, False), (480, 40000, False), (500, 60000, False),
                                      (500, 40003, True), (511, 40000, True), (512, 36000, False)])
def test_wide_pass_matches_oracle(eng, orc, p, n, icpt):
    X, y = orc.synth_logistic(700 + p, 0, n, p, orc.SYNTH_GAUSSIAN)
    rng = np.random.default_rng(p)
    beta = orc.true_beta(p) * 0.7 + 0.03 * rng.standard_normal(p)
    Xd = np.hstack([np.ones((n, 1)), X]) if icpt else X
    bd = np.concatenate([[0.25], beta]) if icpt else beta
    wo, go, llo = orc.logit_pass(Xd, y, bd)
    Ho = orc.gram(Xd, wo)
    H, g, ll, w = eng.newton_wide_pass(dev(X), dev(y), dev(bd), fit_intercept=icpt, want_w=True)
    Hn = H.cpu().numpy()
    assert torch.equal(H, H.T)
    assert rel_inf(w.cpu().numpy(), wo) < 1e-12
    assert np.max(np.abs(g.cpu().numpy() - go)) < 1e-12 * np.max(np.abs(Xd).sum(0))       # a gradient near the MLE cancels: absolute scale
    assert abs(float(ll) - llo) < 1e-12 * abs(llo)
    # bf16 products: entries to ~1e-3 of the diagonal scale, the spectrum far better (rounding errors average over the rows)
    dg = np.sqrt(np.diag(Ho))
    assert np.max(np.abs(Hn - Ho) / np.outer(dg, dg)) < 2e-3
    lo, hi = spectrum_of(Hn, Ho)
    assert 1 - 5e-3 < lo and hi < 1 + 5e-3, (lo, hi)
    # same call again: same bits (fixed summation orders); and g / loglik / w are the logit pass's own bits
    H2, g2, ll2, _ = eng.newton_wide_pass(dev(X), dev(y), dev(bd), fit_intercept=icpt)
    assert torch.equal(H, H2) and torch.equal(g, g2) and torch.equal(ll, ll2)
    w1, g1, ll1 = eng.logit_pass(dev(X), dev(y), dev(bd), fit_intercept=icpt)
    if not (icpt and p % 128 == 0):             # (there the ones column costs the image form one more column chunk per lane: another summation order)
        assert torch.equal(g, g1) and torch.equal(ll, ll1) and torch.equal(w, w1)


def test_wide_pass_strided_rows_and_extreme_eta(eng, orc):
    """partition_id = i % K views (row pitch K * ldx) and |eta| up to ~30 (w underflows towards 0)"""
    n, p, K = 150000, 300, 3
    gen = torch.Generator(device="cuda"); gen.manual_seed(5)
    Xall = torch.randn((n, p), dtype=torch.float64, device="cuda", generator=gen)
    y = (torch.rand(n, dtype=torch.float64, device="cuda", generator=gen) < 0.5).double()
    beta = torch.randn(p, dtype=torch.float64, device="cuda", generator=gen) * 0.5
    Xk, yk = Xall[1::K], y[1::K].contiguous()
    H, g, ll, w = eng.newton_wide_pass(Xk, yk, beta, want_w=True)
    w0, g0, ll0 = eng.logit_pass(Xk.contiguous(), yk, beta)
    H0 = eng.gram(Xk.contiguous(), w0)
    assert float((w - w0).abs().max()) < 1e-14 and float((g - g0).abs().max()) < 1e-11 * float(g0.abs().max())
    lo, hi = spectrum_of(H.cpu().numpy(), H0.cpu().numpy())
    assert 1 - 5e-3 < lo and hi < 1 + 5e-3, (lo, hi)


def test_wide_pass_refuses_other_shapes(eng):
    X = torch.zeros((40000, 100), dtype=torch.float64, device="cuda")
    y = torch.zeros(40000, dtype=torch.float64, device="cuda")
    with pytest.raises(Exception):
        eng.newton_wide_pass(X, y, torch.zeros(100, dtype=torch.float64, device="cuda"))           # too narrow
    X = torch.zeros((1000, 300), dtype=torch.float64, device="cuda")
    with pytest.raises(Exception):
        eng.newton_wide_pass(X, y[:1000], torch.zeros(300, dtype=torch.float64, device="cuda"))    # too few rows


@pytest.mark.parametrize("p,K,nk,icpt,step", [(300, 4, 60000, False, 1), (500, 3, 50000, True, 1), (200, 3, 40000, True, 3), (130, 5, 40000, False, 5)])
def test_fit_with_own_hessian_equals_fit_without_and_oracle(eng, orc, p, K, nk, icpt, step):
    """The partition's own reduced-precision Hessian only shortens the path: same MLE, same Sig_inv (a fresh fp64 Gram at the returned
    coef), fewer iterations; against the oracle's Newton MLE to the north-star tolerance."""
    n = K * nk
    X, y = orc.synth_logistic(11 + p, 0, n, p, orc.SYNTH_UNIFORM)
    Xd, yd = dev(X), dev(y)
    res = {}
    for own in (False, True):
        with eng.irls_options(own_hessian=own, batched=False, small=False):
            if step == 1:
                offs = [k * nk for k in range(K + 1)]
                res[own] = eng.irls_fit_ex(Xd, yd, [offs[k] for k in range(K)], [nk] * K, 1, fit_intercept=icpt)
            else:                               # partition_id = i % K
                res[own] = eng.irls_fit_ex(Xd, yd, list(range(K)), [len(range(k, n, K)) for k in range(K)], K, fit_intercept=icpt)
    a, b = res[False], res[True]
    assert set(int(v) for v in b["status"]) == {0}
    assert rel_inf(b["coef"].cpu().numpy(), a["coef"].cpu().numpy()) < 1e-11
    assert rel_inf(b["Sig_inv"].cpu().numpy(), a["Sig_inv"].cpu().numpy()) < 1e-11
    if nk >= 150 * p:           # (a partition of 100 p rows is so noisy against its neighbours that the first, pooled step is a poor start either way)
        assert sum(int(v) for v in b["n_iter"]) < sum(int(v) for v in a["n_iter"])
    for k in range(K):
        rows = slice(k * nk, (k + 1) * nk) if step == 1 else slice(k, n, K)
        co, smo, so = orc.logistic_model_block(X[rows], y[rows], fit_intercept=icpt)
        assert rel_inf(b["coef"][k].cpu().numpy(), co) < 1e-10
        assert rel_inf(b["Sig_inv"][k].cpu().numpy(), so) < 1e-10
        assert rel_inf(b["Sig_invMcoef"][k].cpu().numpy(), smo) < 1e-10


def test_unequal_partitions_straddling_the_own_hessian_row_limit(eng):
    """a call whose LARGEST partition is beyond the own-Hessian row limit (4e6 rows: no image scratch in the
    workspace dlsa_irls_workspace_bytes() sizes for it) that also holds a partition inside the limit used to pick the wide pass for
    the smaller one and fail it with DLSA_ERR_WORKSPACE.  The gate now also asks whether the pass's scratch fits the workspace of
    THIS call; the smaller partition falls back to plain logit passes and ends at the same MLE as when it is fitted alone (where it
    does take the own-Hessian steps)."""
    p = 130
    offs = [0, 4_200_000, 8_400_000, 10_500_000]
    X, y = eng.synth(909, 0, offs[-1], p, kind=eng.SYNTH_GAUSSIAN)
    r = eng.irls_fit(X, y, offs)
    assert r["rc"] == 0 and r["status"] == [0, 0, 0], (r["rc"], r["status"])
    alone = eng.irls_fit(X[offs[2]:], y[offs[2]:], [0, offs[3] - offs[2]])
    assert alone["status"] == [0]
    assert rel_inf(r["coef"][2].cpu().numpy(), alone["coef"][0].cpu().numpy()) < 1e-10
    assert rel_inf(r["Sig_inv"][2].cpu().numpy(), alone["Sig_inv"][0].cpu().numpy()) < 1e-10
    # i % K partitions of 4 000 001 / 4 000 000 rows with the intercept: the other entry point, the same gate
    n = 8_000_001
    ex = eng.irls_fit_ex(X[:n], y[:n], [0, 1], [4_000_001, 4_000_000], row_step=2, fit_intercept=True)
    assert ex["rc"] == 0 and ex["status"] == [0, 0]
    H = ex["Sig_inv"][1].cpu().numpy()
    assert np.all(np.isfinite(H)) and rel_inf(H, H.T) < 1e-13
