"""GPU: every BASELINE.json configuration at its full per-GPU size, through size-independent properties (the oracle
cannot finish these sizes), plus the map step at config 2's and config 5's widths against the oracle at reduced n."""
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


def _independent_rows_of_H(X, w, rows, step=1_000_000):
    """diag(H) and the given rows of H = X'diag(w)X through plain torch elementwise / reduction code."""
    p = X.shape[1]
    diag = torch.zeros(p, dtype=torch.float64, device="cuda")
    out = {r: torch.zeros(p, dtype=torch.float64, device="cuda") for r in rows}
    for r0 in range(0, X.shape[0], step):
        xs, ws = X[r0:r0 + step], w[r0:r0 + step]
        diag += (xs * xs * ws[:, None]).sum(0)
        for r in rows:
            out[r] += (xs * (xs[:, r] * ws)[:, None]).sum(0)
    return diag, out


def test_full_size_properties_config2_p100(eng):
    """BASELINE config 2: n = 1e7, p = 100 fp64 on one GPU (8 GB).  Gram: exact symmetry, linearity over ragged row
    blocks, diagonal and two rows against an independent torch path; logit pass: g and loglik additive over blocks,
    w identical block by block."""
    n, p = 10_000_000, 100
    X, y = eng.synth(20260101, 0, n, p, kind=eng.SYNTH_GAUSSIAN)
    beta = torch.zeros(p, dtype=torch.float64, device="cuda")
    beta[: int(0.4 * p)] = 1.0
    w, g, ll = eng.logit_pass(X, y, beta)
    H = eng.gram(X, w)
    assert torch.equal(H, H.T)
    scale = float(H.abs().max())
    cut = (n // 3 // 7) * 7 + 5
    H1, H2 = eng.gram(X[:cut], w[:cut]), eng.gram(X[cut:], w[cut:])
    assert float((H1 + H2 - H).abs().max()) < 1e-11 * scale
    diag, rows = _independent_rows_of_H(X, w, (0, 57, 99))
    assert float((diag - torch.diagonal(H)).abs().max()) < 1e-11 * scale
    for r, v in rows.items():
        assert float((v - H[r]).abs().max()) < 1e-11 * scale
    w1, g1, ll1 = eng.logit_pass(X[:cut], y[:cut], beta)
    w2, g2, ll2 = eng.logit_pass(X[cut:], y[cut:], beta)
    assert torch.equal(torch.cat([w1, w2]), w)
    assert float((g1 + g2 - g).abs().max()) < 1e-11 * float(g.abs().max())
    assert abs(float(ll1 + ll2 - ll)) < 1e-11 * abs(float(ll))
    # independent torch evaluation of the gradient and the log-likelihood
    eta = X @ beta
    gt = X.T @ (y - torch.sigmoid(eta))
    llt = (y * eta - torch.nn.functional.softplus(eta)).sum()
    assert float((gt - g).abs().max()) < 1e-10 * float(g.abs().max())
    assert abs(float(llt - ll)) < 1e-11 * abs(float(ll))


def test_full_size_config4_structured_equals_dense(eng):
    """BASELINE config 4's per-GPU shard (113.9M / 8 = 1.4e7 rows, airline-shaped synthetic, p = 260): the structured
    one-hot passes on the raw row (76 B) must give the dense kernels' results on the matrix the design kernel builds
    (2080 B per row): Hessian, gradient, log-likelihood, and the 14-partition map step."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench"))
    from surrogates import airline_shaped        # bench/surrogates.py: test / bench data, not product code
    n, K = 14_000_000, 14
    c = airline_shaped(n)
    X, num, codes, y, plan, p = c["X"], c["num"], c["codes"], c["y"], c["plan"], c["p"]
    assert p == 260 and int(c["seen"].sum()) == p
    beta = c["beta"]
    wd, gd, lld = eng.logit_pass(X, y, beta)
    ws, gs, lls = eng.onehot_logit_pass(plan, num, codes, y, beta)
    assert float((ws - wd).abs().max()) < 1e-13
    assert float((gs - gd).abs().max()) < 1e-11 * float(gd.abs().max())
    assert abs(float(lls - lld)) < 1e-12 * abs(float(lld))
    Hd = eng.gram(X, wd)
    Hs = eng.onehot_gram(plan, num, codes, wd)
    assert torch.equal(Hd, Hd.T)
    assert float((Hs - Hd).abs().max()) < 1e-12 * float(Hd.abs().max())
    # the intercept row of H is the per-column weighted sum: an independent check of both
    col = torch.zeros(p, dtype=torch.float64, device="cuda")
    for r0 in range(0, n, 1_000_000):
        col += (X[r0:r0 + 1_000_000] * wd[r0:r0 + 1_000_000, None]).sum(0)
    assert float((col - Hd[0]).abs().max()) < 1e-11 * float(Hd.abs().max())
    offs = [int(n * k / K) for k in range(K + 1)]
    rs = eng.onehot_irls_fit(plan, num, codes, y, offs)
    rd = eng.irls_fit(X, y, offs)
    assert rs["status"] == [0] * K and rd["status"] == [0] * K
    assert float((rs["coef"] - rd["coef"]).abs().max()) < TOL_MLE * float(rd["coef"].abs().max())
    assert float((rs["Sig_inv"] - rd["Sig_inv"]).abs().max()) < TOL_MLE * float(rd["Sig_inv"].abs().max())
    # and the fitted coefficients solve the score equations of their partition (size-independent MLE check)
    k = 5
    _, gk, _ = eng.logit_pass(X[offs[k]:offs[k + 1]], y[offs[k]:offs[k + 1]], rd["coef"][k].contiguous())
    assert float(gk.abs().max()) < 1e-7


def test_predicted_convergence_agrees_with_confirmed_run_at_config3_scale(eng, monkeypatch):
    """the predicted-convergence exit skips the confirming logit pass.  At BASELINE config 3's per-GPU
    size (2.5e7 x 500) the result must agree with the confirmed run (DLSA_IRLS_PREDICT=0) to 1e-11 -- coef, Sig_inv and
    Sig_invMcoef -- for one shard-sized partition and for 25 partitions of 1e6 rows (logistic_dlsa.py:170)."""
    from conftest import need_hbm
    n, p = 25_000_000, 500
    need_hbm(115e9)            # asserted, never shrunk
    X, y = eng.synth(20260101, 0, n, p, kind=eng.SYNTH_GAUSSIAN)
    for K in (1, 25):
        offs = [int(n * k / K) for k in range(K + 1)]
        monkeypatch.setenv("DLSA_IRLS_PREDICT", "1")
        a = eng.irls_fit(X, y, offs)
        monkeypatch.setenv("DLSA_IRLS_PREDICT", "0")
        b = eng.irls_fit(X, y, offs)
        assert a["status"] == [0] * K and b["status"] == [0] * K
        assert sum(a["n_iter"]) <= sum(b["n_iter"])
        for key in ("coef", "Sig_inv", "Sig_invMcoef"):
            err = float((a[key] - b[key]).abs().max()) / float(b[key].abs().max())
            assert err < 1e-11, (K, key, err)
        # the score at the returned coefficients vanishes to the stopping rule's accuracy
        _, g, _ = eng.logit_pass(X[offs[0]:offs[1]], y[offs[0]:offs[1]], a["coef"][0].contiguous())
        H00 = float(a["Sig_inv"][0].abs().max())
        assert float(g.abs().max()) < 1e-9 * H00


@pytest.mark.parametrize("K", [1, 10])
def test_irls_fit_p100_matches_oracle(eng, orc, K):
    """The map step at BASELINE config 2's width (p = 100: the row-split Gram kernel, the p <= 128 logit pass), K = 1 and
    K = 10 partitions, against the oracle's exact MLE."""
    n, p = 60000, 100
    X, y = orc.synth_logistic(20260102, 0, n, p, orc.SYNTH_GAUSSIAN)
    offs = [int(n * k / K) for k in range(K + 1)]
    r = eng.irls_fit(dev(X), dev(y), offs)
    assert r["status"] == [0] * K
    for k in range(K):
        c, smc, sig = orc.logistic_model_block(X[offs[k]:offs[k + 1]], y[offs[k]:offs[k + 1]])
        assert rel_inf(r["coef"][k].cpu().numpy(), c) < TOL_MLE
        assert rel_inf(r["Sig_inv"][k].cpu().numpy(), sig) < TOL_MLE
        assert rel_inf(r["Sig_invMcoef"][k].cpu().numpy(), smc) < TOL_MLE


def test_linear_partitions_p2000_fp32_wide_kernel_vs_fp64_lstsq(orc):
    """BASELINE config 5's shape (p = 2000, fp32 rows): fit_linear_partitions runs the 256-column-panel Gram kernel and
    xtv_f32 per partition; the WLS combine of the OLS blocks is the global OLS estimate, checked against numpy's fp64
    lstsq on the same rows.  Tolerance: the Gram is accumulated in fp32 (k-ordered fmaf chains over <= 8000 rows per
    partition, ~1e-7 relative per entry); with cond(X'X) ~ 10 for this design theta agrees to 2e-4 relative l_inf."""
    import dlsa_amd
    from dlsa_amd import engine
    n, p, K = 16000, 2000, 2
    X, _ = engine.synth(77, 0, n, p, kind=engine.SYNTH_GAUSSIAN, labels=False, dtype=torch.float32)
    beta = torch.zeros(p, dtype=torch.float32, device="cuda")
    beta[: int(0.4 * p)] = 1.0
    gen = torch.Generator(device="cuda"); gen.manual_seed(5)
    y = X @ beta + torch.randn(n, dtype=torch.float32, device="cuda", generator=gen)
    offs = [0, 8000, n]
    mb = dlsa_amd.fit_linear_partitions(X, y, part_offsets=offs)
    assert mb.status == [0, 0]
    out = dlsa_amd.dlsa_mapred(mb)
    Xh, yh = X.double().cpu().numpy(), y.double().cpu().numpy()
    theta = np.linalg.lstsq(Xh, yh, rcond=None)[0]
    assert rel_inf(out["beta_byOLS"].to_numpy(), theta) < 2e-4
    for k in range(K):
        tk = np.linalg.lstsq(Xh[offs[k]:offs[k + 1]], yh[offs[k]:offs[k + 1]], rcond=None)[0]
        assert rel_inf(mb.coef[k].cpu().numpy(), tk) < 5e-4
        Sk = Xh[offs[k]:offs[k + 1]].T @ Xh[offs[k]:offs[k + 1]]
        assert rel_inf(mb.Sig_inv[k].cpu().numpy(), Sk) < 2e-6
        assert rel_inf(mb.Sig_invMcoef[k].cpu().numpy(), Xh[offs[k]:offs[k + 1]].T @ yh[offs[k]:offs[k + 1]]) < 2e-6
