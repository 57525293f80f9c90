"""LARS path of the least-squares approximation -- host mirror of the reference's dlsa/lsa.py.

`lars_lsa` keeps the reference's signature and result keys (lsa.py:90-212) and runs the whole
path in the HIP engine's persistent LARS kernel.  Inputs may be numpy arrays / np.matrix /
pandas objects / torch tensors; outputs are numpy arrays (`beta` is (steps+1) x m).
Differences from the reference as shipped: plain ndarrays are accepted
(D2), the intercept branch indexes by the matrix dimension and uses `n` only inside log(n) (D4).
"""
import numpy as np
import torch

from . import engine


def _dev(a):
    if isinstance(a, torch.Tensor):
        return a.to(device="cuda", dtype=torch.float64).contiguous()
    return torch.from_numpy(np.ascontiguousarray(np.asarray(a, dtype=np.float64))).cuda()


def lars_path_device(Sigma0, b0, intercept, n, type="lar", eps=np.finfo(float).eps, max_steps=None):
    """Device-tensor variant: returns dict of cuda tensors (no host copy of the path)."""
    S = _dev(Sigma0)
    b = _dev(b0).reshape(-1)
    if S.dim() != 2 or S.shape[0] != S.shape[1] or S.shape[0] != b.shape[0]:
        raise ValueError("Sigma0 must be p x p and b0 of length p")
    return engine.lars_path(S, b, bool(intercept), float(n), type=type, eps=float(eps), max_steps=max_steps)


def lars_lsa(Sigma0, b0, intercept, n, type="lar", eps=np.finfo(float).eps, max_steps=None):
    """Compute the Least Angle Regression or Lasso path of the LSA objective (lsa.py:90-212).
    Returns {'AIC', 'BIC', 'beta', 'beta0'}."""
    r = lars_path_device(Sigma0, b0, intercept, n, type=type, eps=eps, max_steps=max_steps)
    return {"AIC": r["AIC"].cpu().numpy(), "BIC": r["BIC"].cpu().numpy(),
            "beta": r["beta"].cpu().numpy(), "beta0": r["beta0"].cpu().numpy()}
