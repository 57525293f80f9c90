"""GPU: partition chains of the per-partition fit (csrc/irls.hip, irls_fit_core): small partitions are dealt round-robin to up
to four chains (own stream, host thread, workspace slice, warm-start state) that overlap on the device.  The MLE is unique,
so the blocks must agree with the one-chain run and with the oracle; a chained run must be bit-reproducible."""
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


@pytest.mark.parametrize("layout,seed", [("ranges", "0"), ("strided+icpt", "0"), ("ranges", "1"), ("strided+icpt", "1")])
def test_chained_fit_equals_single_chain_and_oracle(eng, orc, monkeypatch, layout, seed):
    """seed = 1: partition 0 is fitted alone and its MLE / factor / pooled Hessian seed every chain (one cold start per call)."""
    n, p, K = 9 * 70000 + 5, 40, 9                        # partitions above the pooled small-partition kernel (<= 65536 rows)
    X, y = orc.synth_logistic(31, 0, n, p, orc.SYNTH_UNIFORM)
    Xd, yd = dev(X), dev(y)

    def fit():
        if layout == "ranges":
            offs = [int(n * k / K) for k in range(K + 1)]
            return eng.irls_fit(Xd, yd, offs), [(X[offs[k]:offs[k + 1]], y[offs[k]:offs[k + 1]]) for k in range(K)]
        rows = [(n - k + K - 1) // K for k in range(K)]
        parts = [(np.hstack([np.ones((rows[k], 1)), X[k::K]]), y[k::K]) for k in range(K)]
        return eng.irls_fit_ex(Xd, yd, list(range(K)), rows, row_step=K, fit_intercept=True), parts

    monkeypatch.setenv("DLSA_IRLS_SEED", seed)
    monkeypatch.setenv("DLSA_IRLS_CHAINS", "1")
    r1, parts = fit()
    monkeypatch.setenv("DLSA_IRLS_CHAINS", "4")
    r4, _ = fit()
    r4b, _ = fit()
    monkeypatch.delenv("DLSA_IRLS_CHAINS")
    rd, _ = fit()                                          # the default count: four chains for partitions of this size
    assert r1["status"] == [0] * K and r4["status"] == [0] * K and rd["status"] == [0] * K
    if seed == "0":
        assert r4["n_iter"] != r1["n_iter"]                # four cold starts instead of one: the chains really were separate
    else:
        assert r4["n_iter"][0] == r1["n_iter"][0]          # the seed partition is the one-chain run's first partition
    for key in ("coef", "Sig_inv", "Sig_invMcoef"):
        assert torch.equal(r4[key], r4b[key]) and torch.equal(r4[key], rd[key]), key        # bit-reproducible
        assert rel_inf(r4[key].cpu().numpy(), r1[key].cpu().numpy()) < 1e-10, key
    for k in (0, 3, K - 1):
        Ak, yk = parts[k]
        c, smc, sig = orc.logistic_model_block(Ak, yk)
        assert rel_inf(r4["coef"][k].cpu().numpy(), c) < 1e-10
        assert rel_inf(r4["Sig_inv"][k].cpu().numpy(), sig) < 1e-10 and rel_inf(r4["Sig_invMcoef"][k].cpu().numpy(), smc) < 1e-10


def test_chained_structured_fit_and_an_empty_partition(eng, monkeypatch):
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench"))
    import surrogates                            # bench/surrogates.py: test / bench data, not product code
    n, K = 8 * 70000, 8
    d = surrogates.airline_shaped(n, dense=False)
    offs = [int(n * k / K) for k in range(K + 1)]
    offs[4] = offs[3]                                      # partition 3 is empty: the reference's zero block, on whatever chain
    monkeypatch.setenv("DLSA_IRLS_CHAINS", "1")
    r1 = eng.onehot_irls_fit(d["plan"], d["num"], d["codes"], d["y"], offs)
    monkeypatch.setenv("DLSA_IRLS_CHAINS", "3")
    r3 = eng.onehot_irls_fit(d["plan"], d["num"], d["codes"], d["y"], offs)
    assert r1["status"] == r3["status"] and r3["status"][3] == 4 and all(s == 0 for i, s in enumerate(r3["status"]) if i != 3)
    assert float(r3["Sig_inv"][3].abs().max()) == 0.0 and float(r3["coef"][3].abs().max()) == 0.0
    for key in ("coef", "Sig_inv", "Sig_invMcoef"):
        assert rel_inf(r3[key].cpu().numpy(), r1[key].cpu().numpy()) < 1e-9, key


def test_partitions_one_row_apart_share_the_workspace_of_the_larger(eng, orc):
    """Found by bench/fit_fuzz.py: with i % 2 partitions of 319489 and 319488 rows the SMALLER one needed 512 Gram slabs where
    the workspace, sized for the larger, held 504 (slab counts are not monotone in the row count; the workspace bounds now are)."""
    n, p = 638977, 5
    X, y = eng.synth(20261002, 0, n, p, kind=eng.SYNTH_UNIFORM)
    r = eng.irls_fit_ex(X, y, [0, 1], [(n + 1) // 2, n // 2], row_step=2, fit_intercept=True)
    assert r["status"] == [0, 0]
    for k in (0, 1):
        Ak = np.hstack([np.ones(((n - k + 1) // 2, 1)), X[k::2].cpu().numpy()])
        c, smc, sig = orc.logistic_model_block(Ak, y[k::2].cpu().numpy())
        assert rel_inf(r["coef"][k].cpu().numpy(), c) < 1e-10 and rel_inf(r["Sig_inv"][k].cpu().numpy(), sig) < 1e-10
