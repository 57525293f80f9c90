"""Inputs of the golden fixtures that are stored as a seed rather than as arrays (tests/golden/F3_lars_p250_*.npz):
regenerated here exactly as oracle/make_golden_r02.py generated them, guarded by the stored checksum."""
import numpy as np


def lars_inputs(p, seed):
    """Sigma = a logistic-Hessian-like SPD matrix, b = noisy sparse vector (the F3 recipe)."""
    rng = np.random.default_rng(seed)
    n = 40 * p
    X = rng.random((n, p)) - 0.5
    w = rng.random(n) * 0.25
    S = X.T @ (w[:, None] * X)
    b = np.zeros(p)
    b[: int(p * 0.4)] = 1.0
    b = b + 0.3 * rng.standard_normal(p) / np.sqrt(n / 50)
    return S, b, n


def lars_case(z):
    """(Sigma, b, n) of an F3 / F3i fixture: stored arrays, or regenerated from the stored seed."""
    if "Sigma" in z.files:
        return z["Sigma"], z["b"], int(z["n"])
    S, b, n = lars_inputs(int(z["p"]), int(z["seed"]))
    chk = np.array([S.sum(), np.abs(S).max(), S[3, 7], b.sum(), b[11]])
    assert np.allclose(chk, z["checksum"], rtol=1e-13, atol=0), "seeded F3 inputs do not regenerate on this numpy"
    return S, b, n
