"""Result objects of the driver -- the reference's on-disk formats.

* `coef_table` / `write_coef_csv`: the coefficient comparison table of projects/results/plot_coef.py:43-51 as stored in
  projects/results/coef.csv:1 -- index `Var`, columns MLE, DLSA_AIC, DLSA_BIC, WLSE, ONE_SHOT.
* `time_table`: the `out_time` frame of projects/logistic_dlsa.py:393-407.
* `save_results` / `load_results`: the pickle list [Sig_inv_beta, out_dlsa, out_par, out_model_eval, out_time]
  (logistic_dlsa.py:411-412)."""
import os
import pickle

import numpy as np
import pandas as pd

COEF_COLUMNS = ["MLE", "DLSA_AIC", "DLSA_BIC", "WLSE", "ONE_SHOT"]      # coef.csv:1, plot_coef.py:50


def coef_table(out_par, names, beta_byMLE=None):
    """out_par: frame with columns beta_byAIC, beta_byBIC, beta_byOLS, beta_byONESHOT (logistic_dlsa.py:353-355);
    names: the p variable names (`intercept` first when fitted); beta_byMLE: the global MLE column (the reference fills it
    from a separate global fit, plot_coef.py:20-41), NaN when not given."""
    p = len(names)
    if out_par.shape[0] != p:
        raise ValueError("coef_table: %d names for %d coefficients" % (p, out_par.shape[0]))
    mle = np.full(p, np.nan) if beta_byMLE is None else np.asarray(beta_byMLE, dtype=np.float64).reshape(p)
    mat = np.column_stack([mle, np.asarray(out_par["beta_byAIC"], dtype=np.float64), np.asarray(out_par["beta_byBIC"], dtype=np.float64),
                           np.asarray(out_par["beta_byOLS"], dtype=np.float64), np.asarray(out_par["beta_byONESHOT"], dtype=np.float64)])
    return pd.DataFrame(mat, index=pd.Index(list(names), name="Var"), columns=COEF_COLUMNS)


def write_coef_csv(path, out_par, names, beta_byMLE=None):
    tab = coef_table(out_par, names, beta_byMLE)
    tab.to_csv(os.path.expanduser(path), index_label="Var")       # plot_coef.py:51
    return tab


def read_coef_csv(path):
    return pd.read_csv(os.path.expanduser(path), index_col="Var")


def time_table(sample_size, partition_num, n_par, memsize_total, time_repartition, time_mapred, time_dlsa,
               time_model_fit, time_model_eval):
    """out_time (logistic_dlsa.py:393-407): one row, the reference's field names."""
    return pd.DataFrame({"sample_size": sample_size, "sample_size_per_partition": sample_size / partition_num, "n_par": n_par,
                         "partition_num": partition_num, "memsize_total": memsize_total, "time_repartition": time_repartition,
                         "time_mapred": time_mapred, "time_dlsa": time_dlsa, "time_model_fit": time_model_fit,
                         "time_model_eval": time_model_eval}, index=[0])


def save_results(path, Sig_inv_beta, out_dlsa, out_par, out_model_eval, out_time):
    with open(os.path.expanduser(path), "wb") as f:
        pickle.dump([Sig_inv_beta, out_dlsa, out_par, out_model_eval, out_time], f)      # logistic_dlsa.py:411-412


def load_results(path):
    with open(os.path.expanduser(path), "rb") as f:
        return pickle.load(f)
