"""Map step of DLSA on the GPU -- the host-side mirror of the reference's dlsa/models.py.

Same function names, argument meaning, output columns and soft-fail behaviour as the
reference (`simulate_logistic` models.py:6-40, `logistic_model` :42-147,
`logistic_model_eval` :151-225); the numeric core (fit, weights, Gram, Sig_inv.coef,
log-likelihood) runs in the HIP engine.  pandas is used exactly where the reference uses it:
for the column bookkeeping of the frames that cross the operator boundary.

Fitting difference (deliberate, SURVEY.md finding 1): the reference stops sklearn's newton-cg at
tol=1e-4, ~1e-3 away from the MLE; this engine returns the exact MLE (Newton/IRLS to a 1e-13
step), which is what the reference computes when its tolerance is tightened.
"""
import warnings

import numpy as np
import pandas as pd
import torch

from . import engine


class MappedBlocks:
    """Device-resident result of the map step for K partitions:
    coef [K,p], Sig_invMcoef [K,p], Sig_inv [K,p,p] (+ column names, per-partition status).
    `to_frame()` gives the reference's stacked layout: K*p rows x (3+p) columns
    `par_id, coef, Sig_invMcoef, <names...>` (models.py:136-142)."""

    def __init__(self, coef, Sig_invMcoef, Sig_inv, names, status=None, n_iter=None, loglik=None,
                 num_partitions=None, sample_size=None):
        self.coef, self.Sig_invMcoef, self.Sig_inv = coef, Sig_invMcoef, Sig_inv
        self.names = list(names)
        K = coef.shape[0]
        self.status = list(status) if status is not None else [0] * K
        self.n_iter = list(n_iter) if n_iter is not None else [0] * K
        self.loglik = list(loglik) if loglik is not None else [0.0] * K
        self.num_partitions = K if num_partitions is None else int(num_partitions)
        self.sample_size = sample_size

    @property
    def columns(self):
        return ["par_id", "coef", "Sig_invMcoef"] + self.names

    def block_frame(self, k):
        p = self.coef.shape[1]
        out_np = torch.cat([self.coef[k][:, None], self.Sig_invMcoef[k][:, None], self.Sig_inv[k]], 1).cpu().numpy()
        out = pd.DataFrame(out_np, columns=pd.Index(["coef", "Sig_invMcoef"] + self.names))
        out.insert(0, "par_id", np.arange(p))
        return out

    def to_frame(self):
        return pd.concat([self.block_frame(k) for k in range(self.coef.shape[0])], ignore_index=True)


def simulate_logistic(sample_size, p, partition_method, partition_num, seed=20260101, kind="uniform"):
    """models.py:6-40 with a seeded counter RNG on the GPU: features ~ U(-0.5,0.5) (:22), beta =
    ones on the first int(0.4p) coordinates (:12,18-19), label ~ Bernoulli(sigmoid(x.beta)) (:23,30),
    partition_id = i % partition_num (:33).  Returns the reference's frame: partition_id, label, x0.."""
    if partition_method != "systematic":
        raise Exception("No such partition method implemented!")      # models.py:35
    X, y = engine.synth(seed, 0, int(sample_size), int(p),
                        kind=engine.SYNTH_UNIFORM if kind == "uniform" else engine.SYNTH_GAUSSIAN)
    n = int(sample_size)
    pid = (np.arange(n) % partition_num).astype(np.float64)
    data_np = np.concatenate((pid[:, None], y.cpu().numpy()[:, None], X.cpu().numpy()), 1)
    return pd.DataFrame(data_np, columns=["partition_id"] + ["label"] + ["x" + str(x) for x in range(p)])


def _design_frame(sample_df, Y_name, fit_intercept, dummy_info, dummy_factors_baseline, data_info, for_eval=False):
    """models.py:50-108 (and :159-206 for eval): column bookkeeping of the design matrix.
    Returns (x_train DataFrame or None when the chunk must be skipped, usecols_full)."""
    col_intercept_name = ["intercept"] if fit_intercept else []
    if len(dummy_info) > 0:
        convert_dummies = list(dummy_info["factor_selected"].keys())
        # fold the dropped levels into one key (models.py:60)
        sample_df = sample_df.replace({k: v for k, v in dummy_info["factor_dropped"].items() if len(v) > 0},
                                      "000_OTHERS")
        X_with_dummies = pd.get_dummies(data=sample_df, drop_first=False, columns=convert_dummies, dtype=float)
        drop = ["partition_id", Y_name] + (list(dummy_factors_baseline) if not for_eval else list(dummy_factors_baseline))
        x_train = X_with_dummies.drop([c for c in drop if c in X_with_dummies.columns], axis=1)
        usecols_x0 = sorted(list(set(sample_df.columns.drop(["partition_id", Y_name])) - set(convert_dummies)))
        usecols_x = usecols_x0.copy()
        for i in convert_dummies:
            for j in sorted(dummy_info["factor_selected_names"][i]):
                usecols_x.append(j)
        usecols_x = [i for i in usecols_x if i not in dummy_factors_baseline]
        usecols_full = ["par_id", "coef", "Sig_invMcoef"] + col_intercept_name + usecols_x
        if set(x_train.columns) != set(usecols_x):
            missing = set(usecols_x) - set(x_train.columns)
            if not for_eval:
                warnings.warn("Dummies:" + str(missing) + "missing in this data chunk " + str(x_train.shape)
                              + "Skip modeling this part of data.")
                return None, usecols_full, usecols_x0
            warnings.warn("Dummies:" + str(missing) + "missing in this data chunk " + str(x_train.shape))
            for c in missing:                    # models.py:196-198: absent levels count as zeros
                x_train[c] = 0.0
    else:
        drop = ["partition_id", Y_name] + ([] if for_eval else list(dummy_factors_baseline))
        x_train = sample_df.drop(drop, axis=1)
        usecols_x0 = list(x_train.columns)
        usecols_x = usecols_x0
        usecols_full = ["par_id", "coef", "Sig_invMcoef"] + col_intercept_name + list(usecols_x)
    x_train = x_train.copy()
    if len(data_info) > 0:                        # models.py:99-101: rows 1,2 of describe() = mean, stddev
        for i in usecols_x0:
            x_train[i] = (x_train[i] - float(data_info[i][1])) / float(data_info[i][2])
    x_train = x_train.reindex(columns=usecols_x)  # models.py:104
    return x_train, usecols_full, usecols_x0


def _to_device_design(x_train, fit_intercept):
    """Numeric frame -> row-major fp64 device matrix; the intercept is a leading ones column
    (models.py:121-122)."""
    xn = np.ascontiguousarray(x_train.to_numpy(dtype=np.float64))
    if fit_intercept:
        xn = np.concatenate([np.ones((xn.shape[0], 1)), xn], axis=1)
    return torch.from_numpy(xn).cuda()


def logistic_model(sample_df, Y_name, fit_intercept=False, dummy_info=[], dummy_factors_baseline=[],
                   data_info=[]):
    """Run the logistic model on one partition (a pandas frame), as the reference's GROUPED_MAP
    UDF does (models.py:42-147).  Returns the p x (3+p) frame `par_id, coef, Sig_invMcoef,
    [intercept,] <features>`; a chunk that lacks an expected dummy level returns the all-zero
    block with a warning (models.py:84-91)."""
    x_train, usecols_full, _ = _design_frame(sample_df, Y_name, fit_intercept, dummy_info,
                                             dummy_factors_baseline, data_info)
    if x_train is None:
        return pd.DataFrame(0, index=np.arange(len(usecols_full) - 3), columns=usecols_full)
    names = (["intercept"] if fit_intercept else []) + list(x_train.columns)
    Xd = _to_device_design(x_train, fit_intercept)
    yd = torch.from_numpy(np.ascontiguousarray(sample_df[Y_name].to_numpy(dtype=np.float64))).cuda()
    r = engine.irls_fit(Xd, yd, [0, Xd.shape[0]])
    st = r["status"][0]
    if st == 1:
        warnings.warn("logistic_model: Newton iterations did not converge (max_iter reached)")
    elif st == 2:
        warnings.warn("logistic_model: Hessian not positive definite (collinear or separable data)")
    blocks = MappedBlocks(r["coef"], r["Sig_invMcoef"], r["Sig_inv"], names, r["status"], r["n_iter"], r["loglik"])
    out = blocks.block_frame(0)
    if out.isna().values.any():
        warnings.warn("NAs appear in the final output")     # models.py:144-145
    return out


def fit_logistic_partitions(X, y, partition_num=None, part_offsets=None, fit_intercept=False, names=None,
                            tol=1e-13, max_iter=100):
    """Tensor fast path of the map step for MANY partitions of one device-resident shard.

    X [n, p] fp64 row-major on the GPU, y [n].  Either `part_offsets` (K+1 ints: partition k is the
    contiguous row range [off[k], off[k+1]) -- the layout `repartition(K, "partition_id")` gives,
    logistic_dlsa.py:295) or `partition_num` (systematic partition_id = i % K, models.py:33; the
    rows are gathered into contiguous partitions on the device first).  With fit_intercept a
    leading ones column is materialised (models.py:121-122).  Returns MappedBlocks."""
    if not X.is_cuda:
        raise RuntimeError("fit_logistic_partitions runs on the GPU only (no CPU fallback)")
    n, p = X.shape
    if part_offsets is None:
        K = int(partition_num) if partition_num else 1
        if K > 1:
            idx = torch.arange(n, device=X.device)
            order = torch.argsort(idx % K, stable=True)
            X, y = X[order], y[order]
            counts = torch.bincount(idx % K, minlength=K).cpu().tolist()
        else:
            counts = [n]
        part_offsets = np.concatenate([[0], np.cumsum(counts)])
    if fit_intercept:
        X = torch.cat([torch.ones((n, 1), dtype=X.dtype, device=X.device), X], dim=1)
    if names is None:
        names = ["x" + str(i) for i in range(p)]
    names = (["intercept"] if fit_intercept else []) + list(names)
    r = engine.irls_fit(X.contiguous(), y.contiguous(), part_offsets, tol=tol, max_iter=max_iter)
    return MappedBlocks(r["coef"], r["Sig_invMcoef"], r["Sig_inv"], names, r["status"], r["n_iter"], r["loglik"],
                        sample_size=n)


def logistic_model_eval(sample_df, Y_name, par, fit_intercept=False, dummy_info=[], dummy_factors_baseline=[],
                        data_info=[]):
    """Log-likelihood of every estimator column of `par` on one partition (models.py:151-225)."""
    x_train, _, _ = _design_frame(sample_df, Y_name, fit_intercept, dummy_info, dummy_factors_baseline,
                                  data_info, for_eval=True)
    Xd = _to_device_design(x_train, fit_intercept)
    yd = torch.from_numpy(np.ascontiguousarray(sample_df[Y_name].to_numpy(dtype=np.float64))).cuda()
    pard = torch.from_numpy(np.ascontiguousarray(np.asarray(par, dtype=np.float64))).cuda()
    ll = engine.loglik(Xd, yd, pard).cpu().numpy()
    return pd.DataFrame({par.columns[i]: [ll[i]] for i in range(par.shape[1])})


# ---------------------------------------------------------------------------------------------
# Linear-regression map step.  The reference claims linear DLSA (README.md:6) but ships
# only the logistic map; this sibling keeps the same block layout so dlsa_mapred / dlsa apply unchanged:
#   coef = OLS estimate, Sig_inv = X'X, Sig_invMcoef = X'y (= X'X coef).
# With these blocks the WLS combine of dlsa_mapred equals the global OLS estimate exactly.
# ---------------------------------------------------------------------------------------------
def fit_linear_partitions(X, y, partition_num=None, part_offsets=None, fit_intercept=False, names=None):
    """Tensor fast path: one Gram pass X'X + one X'y pass per partition.  Returns MappedBlocks whose
    `loglik` slot carries the residual sum of squares of each partition."""
    if not X.is_cuda:
        raise RuntimeError("fit_linear_partitions runs on the GPU only (no CPU fallback)")
    n, p = X.shape
    if part_offsets is None:
        K = int(partition_num) if partition_num else 1
        if K > 1:
            idx = torch.arange(n, device=X.device)
            order = torch.argsort(idx % K, stable=True)
            X, y = X[order], y[order]
            counts = torch.bincount(idx % K, minlength=K).cpu().tolist()
        else:
            counts = [n]
        part_offsets = np.concatenate([[0], np.cumsum(counts)])
    if fit_intercept:
        X = torch.cat([torch.ones((n, 1), dtype=X.dtype, device=X.device), X], dim=1)
    X, y = X.contiguous(), y.contiguous()
    pp = X.shape[1]
    if names is None:
        names = ["x" + str(i) for i in range(p)]
    names = (["intercept"] if fit_intercept else []) + list(names)
    offs = [int(v) for v in part_offsets]
    K = len(offs) - 1
    coef = torch.zeros((K, pp), dtype=torch.float64, device=X.device)
    smc = torch.zeros((K, pp), dtype=torch.float64, device=X.device)
    sig = torch.zeros((K, pp, pp), dtype=torch.float64, device=X.device)
    f32 = X.dtype == torch.float32       # config 5: fp32 rows, fp32 Gram; the p x p blocks are kept in fp64
    Hk = torch.empty((pp, pp), dtype=X.dtype, device=X.device) if f32 else None
    status, rss = [], []
    for k in range(K):
        lo, hi = offs[k], offs[k + 1]
        if hi <= lo:
            status.append(4); rss.append(0.0)
            continue
        if f32:
            engine.gram(X[lo:hi], None, out=Hk)
            sig[k] = Hk.double()
        else:
            engine.gram(X[lo:hi], None, out=sig[k])
        g, yy = engine.xtv(X[lo:hi], y[lo:hi])
        g, yy = g.double(), yy.double()
        smc[k] = g
        try:
            coef[k] = engine.spd_solve(sig[k], g)
            status.append(0)
            rss.append(float((yy - torch.dot(coef[k], g)).item()))       # y'y - theta'X'y
        except Exception:
            status.append(2); rss.append(float("nan"))
            warnings.warn("linear_model: X'X not positive definite (collinear design)")
    return MappedBlocks(coef, smc, sig, names, status, [1] * K, rss, sample_size=n)


def linear_model(sample_df, Y_name, fit_intercept=False, dummy_info=[], dummy_factors_baseline=[], data_info=[]):
    """Frame-level sibling of logistic_model for a linear response: same arguments, same
    p x (3+p) output frame `par_id, coef, Sig_invMcoef, [intercept,] <features>`."""
    x_train, usecols_full, _ = _design_frame(sample_df, Y_name, fit_intercept, dummy_info,
                                             dummy_factors_baseline, data_info)
    if x_train is None:
        return pd.DataFrame(0, index=np.arange(len(usecols_full) - 3), columns=usecols_full)
    Xd = torch.from_numpy(np.ascontiguousarray(x_train.to_numpy(dtype=np.float64))).cuda()
    yd = torch.from_numpy(np.ascontiguousarray(sample_df[Y_name].to_numpy(dtype=np.float64))).cuda()
    mb = fit_linear_partitions(Xd, yd, fit_intercept=fit_intercept, names=list(x_train.columns))
    out = mb.block_frame(0)
    if out.isna().values.any():
        warnings.warn("NAs appear in the final output")
    return out
