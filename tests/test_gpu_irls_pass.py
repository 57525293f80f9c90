"""GPU: the fused Newton pass (csrc/irls_pass.hip, dlsa_irls_pass_f64): w, g = X'(y - mu), loglik and H = X'WX in ONE read of
the rows for narrow designs, against the oracle's logit_pass + gram (dlsa/models.py:110-114,130 restated) and against the
two-launch form, over every shape class of the kernel; and the fit that uses it against the reference's goldens."""
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


# NT = 3..7 full tiles x G = 0..3 tail groups; ragged row counts (not a multiple of the 32-row chunk, of the slab)
# (round 5: odd widths too -- packed rows at 8-byte offsets, the piece of a row's last column carries the next row's first element)
@pytest.mark.parametrize("p,n", [(50, 8192), (52, 9001), (56, 10000), (60, 12345), (64, 8200), (66, 20000), (72, 15000), (76, 9999),
                                 (80, 30011), (84, 8192), (88, 17000), (92, 8193), (96, 40000), (100, 60000), (104, 25001),
                                 (108, 33333), (112, 50001), (116, 20000), (120, 30001),
                                 (49, 9000), (51, 8192), (63, 10001), (65, 20000), (81, 12000), (99, 60000), (101, 20001), (111, 50000), (119, 9999)])
def test_fused_pass_matches_oracle(eng, orc, p, n):
    X, y = orc.synth_logistic(300 + p, 0, n, p, orc.SYNTH_GAUSSIAN)
    rng = np.random.default_rng(p)
    beta = orc.true_beta(p) * 0.8 + 0.05 * rng.standard_normal(p)
    wo, go, llo = orc.logit_pass(X, y, beta)
    Ho = orc.gram(X, wo)
    H, g, ll, w = eng.irls_pass(dev(X), dev(y), dev(beta), want_w=True)
    assert eng.gram_last_kernel()[0].startswith("irls_pass_narrow_kernel<true,true"), eng.gram_last_kernel()
    assert torch.equal(H, H.T)
    assert rel_inf(w.cpu().numpy(), wo) < 1e-12
    assert rel_inf(H.cpu().numpy(), Ho) < 1e-12
    assert np.max(np.abs(g.cpu().numpy() - go)) < 1e-12 * np.max(np.abs(X).sum(0))       # a gradient near the MLE cancels: absolute scale
    assert abs(float(ll) - llo) < 1e-12 * abs(llo)
    # without w_out: the other instantiation, same bits for H / g / loglik; twice the same call: same bits (fixed orders)
    H2, g2, ll2, _ = eng.irls_pass(dev(X), dev(y), dev(beta))
    assert eng.gram_last_kernel()[0].startswith("irls_pass_narrow_kernel<false,true")
    assert torch.equal(H, H2) and torch.equal(g, g2) and torch.equal(ll, ll2)


def test_fused_pass_equals_two_launch_form_and_extreme_eta(eng, monkeypatch):
    """|eta| up to ~40 (w underflows towards 0, softplus is linear) and the two-launch form of the same entry point."""
    n, p = 40000, 100
    gen = torch.Generator(device="cuda"); gen.manual_seed(3)
    X = torch.randn((n, p), dtype=torch.float64, device="cuda", generator=gen)
    beta = torch.randn(p, dtype=torch.float64, device="cuda", generator=gen) * 1.2
    y = (torch.rand(n, dtype=torch.float64, device="cuda", generator=gen) < 0.5).double()
    H, g, ll, w = eng.irls_pass(X, y, beta, want_w=True)
    monkeypatch.setenv("DLSA_IRLS_FUSED", "0")
    H0, g0, ll0, w0 = eng.irls_pass(X, y, beta, want_w=True)
    assert not eng.gram_last_kernel()[0].startswith("irls_pass")
    assert float((w - w0).abs().max()) < 1e-14
    assert float((H - H0).abs().max()) < 1e-12 * float(H0.abs().max())
    assert float((g - g0).abs().max()) < 1e-11 * float(g0.abs().max())
    assert abs(float(ll) - float(ll0)) < 1e-12 * abs(float(ll0))


def test_other_shapes_take_the_two_launches(eng, orc):
    for p, n in ((30, 9000), (48, 20000), (200, 9000), (100, 5000), (122, 20000)):   # too narrow (twice), too wide, too few rows, 7 tiles + 3 groups
        X, y = orc.synth_logistic(p, 0, n, p)
        beta = orc.true_beta(p) * 0.5
        H, g, ll, w = eng.irls_pass(dev(X), dev(y), dev(beta), want_w=True)
        assert not eng.gram_last_kernel()[0].startswith("irls_pass"), (p, n)
        wo, go, llo = orc.logit_pass(X, y, beta)
        assert rel_inf(H.cpu().numpy(), orc.gram(X, wo)) < 1e-12 and rel_inf(g.cpu().numpy(), go) < 1e-11


def test_fused_pass_full_size_config2_properties(eng):
    """BASELINE config 2 (n = 1e7, p = 100): the fused pass over the whole shard equals the sum of its parts over ragged
    row blocks, and w / g / loglik / H those of the separate passes."""
    n, p = 10_000_000, 100
    X, y = eng.synth(20260101, 0, n, p, kind=eng.SYNTH_GAUSSIAN)
    beta = torch.zeros(p, dtype=torch.float64, device="cuda"); beta[:40] = 0.9
    H, g, ll, _ = eng.irls_pass(X, y, beta)
    assert eng.gram_last_kernel()[0] == "irls_pass_narrow_kernel<false,true,6,1>"
    w0, g0, ll0 = eng.logit_pass(X, y, beta)
    H0 = eng.gram(X, w0)
    scale = float(H0.abs().max())
    assert float((H - H0).abs().max()) < 1e-12 * scale
    assert float((g - g0).abs().max()) < 1e-12 * scale and abs(float(ll) - float(ll0)) < 1e-12 * abs(float(ll0))
    cut = (n // 3 // 7) * 7 + 5
    Ha, ga, lla, _ = eng.irls_pass(X[:cut], y[:cut], beta)
    Hb, gb, llb, _ = eng.irls_pass(X[cut:], y[cut:], beta)
    assert float((Ha + Hb - H).abs().max()) < 1e-12 * scale
    assert float((ga + gb - g).abs().max()) < 1e-12 * scale and abs(float(lla) + float(llb) - float(ll)) < 1e-12 * abs(float(ll))


def test_ring_logit_pass_is_block_invariant_and_handles_odd_label_offsets(eng, orc):
    """The narrow logit pass now streams through the same LDS-DMA ring (irls_pass_narrow_kernel<*, false, ...>): w of a row
    must not depend on how the rows are cut into calls (odd cuts leave y 8 bytes off a 16-byte boundary), g and loglik add up."""
    n, p = 50001, 100
    X, y = orc.synth_logistic(77, 0, n, p, orc.SYNTH_GAUSSIAN)
    beta = orc.true_beta(p) * 0.7
    Xd, yd, bd = dev(X), dev(y), dev(beta)
    w, g, ll = eng.logit_pass(Xd, yd, bd)
    assert eng.gram_last_kernel()[0].startswith("irls_pass_narrow_kernel<true,false"), eng.gram_last_kernel()
    wo, go, llo = orc.logit_pass(X, y, beta)
    assert rel_inf(w.cpu().numpy(), wo) < 1e-12 and np.max(np.abs(g.cpu().numpy() - go)) < 1e-12 * np.max(np.abs(X).sum(0))
    assert abs(float(ll) - llo) < 1e-12 * abs(llo)
    for cut in (20001, 8193, 30000):
        w1, g1, l1 = eng.logit_pass(Xd[:cut], yd[:cut], bd)
        w2, g2, l2 = eng.logit_pass(Xd[cut:], yd[cut:], bd)
        assert torch.equal(torch.cat([w1, w2]), w), cut
        assert float((g1 + g2 - g).abs().max()) < 1e-11 * float(g.abs().max())
        assert abs(float(l1 + l2 - ll)) < 1e-12 * abs(float(ll))
    # g only / loglik only / w only
    _, g3, _ = eng.logit_pass(Xd, yd, bd, want_w=False, want_loglik=False)
    assert torch.equal(g3, g)


@pytest.mark.parametrize("p,n", [(50, 8192), (50, 70001), (64, 33333), (80, 41234), (96, 20000), (120, 9000)])
def test_ring_logit_pass_short_rows_two_workgroups_per_cu(eng, orc, p, n):
    """Short rows (<= 5 tile columns) run the logit-only ring with three stages and two workgroups per CU; longer ones one.
    Both against the oracle; repeated calls give identical bits (fixed slab order)."""
    X, y = orc.synth_logistic(91 + p, 0, n, p, orc.SYNTH_GAUSSIAN)
    beta = orc.true_beta(p) * 0.6
    Xd, yd, bd = dev(X), dev(y), dev(beta)
    w, g, ll = eng.logit_pass(Xd, yd, bd)
    assert eng.gram_last_kernel()[0].startswith("irls_pass_narrow_kernel<true,false"), eng.gram_last_kernel()
    wo, go, llo = orc.logit_pass(X, y, beta)
    assert rel_inf(w.cpu().numpy(), wo) < 1e-12
    assert np.max(np.abs(g.cpu().numpy() - go)) < 1e-12 * np.max(np.abs(X).sum(0))
    assert abs(float(ll) - llo) < 1e-12 * abs(llo)
    w2, g2, ll2 = eng.logit_pass(Xd, yd, bd)
    assert torch.equal(w, w2) and torch.equal(g, g2) and torch.equal(ll, ll2)


@pytest.mark.parametrize("p,n", [(100, 40011), (50, 9001), (64, 20000), (118, 33333), (98, 16385)])
def test_fits_with_the_implicit_intercept_take_the_fused_pass(eng, orc, p, n):
    """Round 4: the fused kernel carries the intercept of models.py:121-122 as a ones column in its LDS stages (the reference's driver
    always fits one, logistic_dlsa.py:79), rows past a slab's end masked by their validity.  A fit with fit_intercept must give the
    two-launch path's and the oracle's MLE / Hessian, intercept first (models.py:136-142), for ragged row counts too."""
    X, y = orc.synth_logistic(500 + p, 0, n, p, orc.SYNTH_GAUSSIAN)
    Xd, yd = dev(X), dev(y)
    first, rows = [0, n // 3], [n // 3, n - n // 3]
    with eng.irls_options(batched=False, small=False):
        f = eng.irls_fit_ex(Xd, yd, first, rows, fit_intercept=True)
    kern = eng.gram_last_kernel()[0]
    assert "icpt" in kern or rows[1] < 8192, kern
    with eng.irls_options(batched=False, small=False, fused=False):
        u = eng.irls_fit_ex(Xd, yd, first, rows, fit_intercept=True)
    assert f["status"] == u["status"] == [0, 0]
    for key in ("coef", "Sig_inv", "Sig_invMcoef"):
        assert rel_inf(f[key].cpu().numpy(), u[key].cpu().numpy()) < 1e-10, key
    k = 1
    co, smc, sig = orc.logistic_model_block(X[first[k]:first[k] + rows[k]], y[first[k]:first[k] + rows[k]], True)
    assert rel_inf(f["coef"][k].cpu().numpy(), co) < 1e-10
    assert rel_inf(f["Sig_inv"][k].cpu().numpy(), sig) < 1e-10
    assert rel_inf(f["Sig_invMcoef"][k].cpu().numpy(), smc) < 1e-10
    S = f["Sig_inv"][k].cpu().numpy()
    assert np.array_equal(S, S.T) and abs(S[0, 0] - float(np.sum(orc.logit_pass(np.hstack([np.ones((rows[k], 1)), X[first[k]:]]), y[first[k]:], co)[0]))) < 1e-9 * S[0, 0]
