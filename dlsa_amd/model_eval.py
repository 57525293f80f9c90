"""Total log-likelihood of every estimator column over all partitions -- host mirror of the reference's
dlsa/model_eval.py (`logistic_model_eval_sdf`, :10-42): the per-partition `logistic_model_eval` (models.py:151-225)
summed over the partitions (the reference's `groupby().sum()` + `toPandas()`), and over the ranks of a
torch.distributed job with one all-reduce of c doubles."""
import numpy as np
import pandas as pd
import torch

from . import distributed, engine
from .models import logistic_model_eval


def logistic_model_eval_sdf(data_sdf, par, fit_intercept, Y_name, dummy_info=[], dummy_factors_baseline=[], data_info=[]):
    """Evaluate model performance (model_eval.py:10-42).  `par`: p-row frame, one column per method (e.g. beta_byAIC,
    beta_byBIC, beta_byOLS, beta_byONESHOT, logistic_dlsa.py:353-363).  `data_sdf`: this rank's rows as a pandas frame with
    a `partition_id` column (grouped like the reference's `groupby("partition_id").apply`), or an iterable of
    per-partition frames.  Returns a one-row frame with par's columns holding the total log-likelihoods."""
    if isinstance(data_sdf, pd.DataFrame):
        chunks = (g for _, g in data_sdf.groupby("partition_id", sort=True)) if "partition_id" in data_sdf.columns else [data_sdf]
    else:
        chunks = data_sdf
    total = np.zeros(par.shape[1], dtype=np.float64)
    for chunk in chunks:
        out = logistic_model_eval(sample_df=chunk, Y_name=Y_name, fit_intercept=fit_intercept, par=par,
                                  dummy_info=dummy_info, dummy_factors_baseline=dummy_factors_baseline, data_info=data_info)
        total += out.to_numpy(dtype=np.float64).ravel()
    if distributed.is_distributed():
        dev = "cuda" if torch.cuda.is_available() else "cpu"
        total = distributed.allreduce_message(torch.from_numpy(total).to(dev)).cpu().numpy()
    return pd.DataFrame([total], columns=list(par.columns))


def loglik_partitions(X, y, par, fit_intercept=False):
    """Tensor fast path: X [n, p(-1)] device-resident shard, y [n], par [p, c] (frame, array or tensor).  One read of X for
    all c columns (dlsa_loglik_f64), summed over the ranks.  Returns a device tensor [c]."""
    if isinstance(par, pd.DataFrame):
        par = par.to_numpy(dtype=np.float64)
    if not torch.is_tensor(par):
        par = torch.from_numpy(np.ascontiguousarray(np.asarray(par, dtype=np.float64)))
    par = par.to(device=X.device, dtype=torch.float64)
    out = torch.zeros(par.shape[1], dtype=torch.float64, device=X.device)
    for c0 in range(0, par.shape[1], 8):                 # dlsa_loglik_f64 takes up to 8 columns per pass
        out[c0:c0 + 8] = engine.loglik(engine.row_major(X), y.to(torch.float64).contiguous(), par[:, c0:c0 + 8].contiguous(),
                                       fit_intercept=fit_intercept)      # implicit ones column: no copy of the shard
    return distributed.allreduce_message(out)
