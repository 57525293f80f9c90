"""GPU: the WLS combine with lstsq semantics (dlsa/dlsa.py:48-49) -- Cholesky for an SPD sum, the minimum-norm
least-squares solution (device Jacobi eigendecomposition) for a singular one -- against the reference's own
dlsa_mapred outputs on rank-deficient blocks (fixture F2r) and against numpy's lstsq / eigvalsh."""
import os
import warnings

import numpy as np
import pandas as pd
import pytest

from conftest import GOLDEN

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


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("case", ["zero", "dup", "both"])
def test_rank_deficient_mapred_matches_reference(eng, case):
    """The reference's dlsa_mapred on blocks whose sum is singular: beta_byOLS is lstsq's minimum-norm solution."""
    import dlsa_amd
    z = np.load(os.path.join(GOLDEN, "F2r_rankdef_%s_K3_p6.npz" % case))
    K, p = z["coef"].shape
    names = ["x%d" % i for i in range(p)]
    mb = dlsa_amd.MappedBlocks(dev(z["coef"]), dev(z["Sig_invMcoef"]), dev(z["Sig_inv"]), names)
    with pytest.warns(UserWarning, match="rank %d < %d" % (int(z["rank"]), p)):
        out = dlsa_amd.dlsa_mapred(mb)
    assert list(out.columns) == ["beta_byOLS", "beta_byONESHOT"] + names
    assert rel_inf(out["beta_byOLS"], z["beta_byOLS"]) < 1e-10
    assert rel_inf(out["beta_byONESHOT"], z["beta_byONESHOT"]) < 1e-14
    assert rel_inf(out.iloc[:, 2:], z["Sig_inv_sum"]) < 1e-14
    # the same through the stacked-frame entry (the reference's layout)
    frames = [pd.DataFrame(np.column_stack([np.arange(p), z["coef"][k], z["Sig_invMcoef"][k], z["Sig_inv"][k]]),
                           columns=["par_id", "coef", "Sig_invMcoef"] + names) for k in range(K)]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        out2 = dlsa_amd.dlsa_mapred(pd.concat(frames, ignore_index=True))
    assert rel_inf(out2["beta_byOLS"], z["beta_byOLS"]) < 1e-10
    theta, rank = eng.wls_solve(dev(z["Sig_inv_sum"]), dev(z["Sig_invMcoef"].sum(0)))
    assert rank == int(z["rank"])


def test_spd_sum_takes_the_cholesky_path_and_reports_full_rank(eng):
    z = np.load(os.path.join(GOLDEN, "F2_spd_K3_p4.npz"))
    theta, rank = eng.wls_solve(dev(z["Sig_inv_sum"]), dev(z["Sig_invMcoef"].sum(0)))
    assert rank == 4 and rel_inf(theta.cpu().numpy(), z["beta_byOLS"]) < 1e-12
    assert torch.equal(theta, eng.spd_solve(dev(z["Sig_inv_sum"]), dev(z["Sig_invMcoef"].sum(0))))


@pytest.mark.parametrize("p,deficit", [(1, 0), (2, 1), (7, 0), (64, 5), (257, 30), (500, 1)])
def test_sym_pinv_solve_matches_numpy_lstsq_and_eigvalsh(eng, p, deficit):
    rng = np.random.default_rng(p * 31 + deficit)
    A = rng.standard_normal((p + 10, p))
    if deficit:
        idx = rng.choice(p, deficit, replace=False)
        A[:, idx[: deficit // 2]] = 0.0                                    # absent levels
        for j in idx[deficit // 2:]:
            A[:, j] = A[:, (j + 1) % p] if (j + 1) % p not in idx else 0.0   # duplicated columns
    S = A.T @ A
    v = S @ rng.standard_normal(p)
    theta, rank, eig = eng.sym_pinv_solve(dev(S), dev(v))
    ref = np.linalg.lstsq(S, v, rcond=None)[0]
    assert rank == np.linalg.matrix_rank(S)
    assert rel_inf(theta.cpu().numpy(), ref) < 1e-9
    lam = np.linalg.eigvalsh(S)
    assert np.max(np.abs(np.sort(np.asarray(eig)) - lam)) < 1e-12 * max(1.0, lam.max())
    th2, rank2 = eng.wls_solve(dev(S), dev(v))
    assert rank2 == rank and rel_inf(th2.cpu().numpy(), ref) < 1e-9


def test_wls_solve_handles_an_indefinite_matrix_like_lstsq(eng):
    """lstsq does not need S >= 0: singular values are |eigenvalues|."""
    rng = np.random.default_rng(9)
    Q, _ = np.linalg.qr(rng.standard_normal((12, 12)))
    lam = np.array([5, 3, 2, 1, 0.5, -0.7, -2, 0, 0, 4, 6, -1.5])
    S = (Q * lam) @ Q.T
    v = rng.standard_normal(12)
    theta, rank = eng.wls_solve(dev(S), dev(v))
    assert rank == 10
    assert rel_inf(theta.cpu().numpy(), np.linalg.lstsq(S, v, rcond=None)[0]) < 1e-10


def test_all_zero_blocks_give_the_zero_estimate(eng):
    """Every chunk skipped (models.py:84-91 returns all-zero blocks): lstsq of the zero system is 0."""
    import dlsa_amd
    K, p = 3, 5
    z = torch.zeros((K, p), dtype=torch.float64, device="cuda")
    mb = dlsa_amd.MappedBlocks(z, z.clone(), torch.zeros((K, p, p), dtype=torch.float64, device="cuda"), ["x%d" % i for i in range(p)])
    with pytest.warns(UserWarning, match="rank 0"):
        out = dlsa_amd.dlsa_mapred(mb)
    assert np.all(out["beta_byOLS"].to_numpy() == 0.0) and np.all(out["beta_byONESHOT"].to_numpy() == 0.0)


def test_badly_scaled_spd_sum_follows_lstsq_not_cholesky():
    """lstsq(rcond=None) cuts singular values relative to sigma_MAX.  diag(1, 1e-20) factors without a failing
    pivot (each pivot is fine relative to its own diagonal entry) but numpy returns the truncated minimum-norm solution."""
    from dlsa_amd import engine as eng
    for S, v in ((np.diag([1.0, 1e-20]), np.array([2.0, 3e-20])),
                 (np.diag([4.0, 1.0, 1e-18, 2.0]), np.array([1.0, -1.0, 1e-18, 0.5]))):
        ref = np.linalg.lstsq(S, v, rcond=None)[0]
        th, rank = eng.wls_solve(torch.from_numpy(S).cuda(), torch.from_numpy(v).cuda())
        assert rank == np.linalg.matrix_rank(S)
        assert np.allclose(th.cpu().numpy(), ref, rtol=1e-12, atol=1e-14), (th, ref)
    # a well-scaled but wide-range SPD matrix (cond 1e8) still takes the Cholesky path and solves exactly
    rng = np.random.default_rng(4)
    Q, _ = np.linalg.qr(rng.standard_normal((40, 40)))
    S = Q @ np.diag(np.logspace(0, 8, 40)) @ Q.T
    S = (S + S.T) / 2
    v = rng.standard_normal(40)
    th, rank = eng.wls_solve(torch.from_numpy(S).cuda(), torch.from_numpy(v).cuda())
    assert rank == 40 and np.allclose(S @ th.cpu().numpy(), v, rtol=1e-6, atol=1e-6)
