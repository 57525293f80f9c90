"""Reduce / combine / shrink -- host mirror of the reference's dlsa/dlsa.py.

`dlsa_mapred` (dlsa.py:21-61) sums the per-partition blocks (on this rank, then across ranks with
ONE all-reduce when torch.distributed is initialised -- the algorithm's single round of
communication), solves the WLS system on the GPU and returns the reference's frame
`beta_byOLS, beta_byONESHOT, <names...>`.  `dlsa` (dlsa.py:70-107) runs the LARS path on the GPU
and picks the AIC / BIC minimisers.  Aliases `dlsa_mapreduce` / `dlsa_fit` follow README.md:25-27.
"""
import warnings

import numpy as np
import pandas as pd
import torch

from . import distributed, engine
from .lsa import lars_path_device
from .models import MappedBlocks


def _blocks_from_frame(pdf):
    """Stacked reference layout (K*p rows, columns par_id | coef | Sig_invMcoef | names...) ->
    per-par_id sums on the device (dlsa.py:30-34 groupby('par_id').sum)."""
    cols = list(pdf.columns)
    names = cols[3:]
    p = len(names)
    par_id = pdf.iloc[:, 0].to_numpy(dtype=np.int64)
    vals = pdf.iloc[:, 1:].to_numpy(dtype=np.float64)                 # as pandas holds it (often column-major): no host reorder
    if vals.shape[0] == 0:
        raise Exception("Zero-length grouped pandas DataFrame obtained, check the input.")   # dlsa.py:36-39
    # A stacked FRAME is host data: it goes to the device as it lies (engine.rows_to_device puts it in row order there) and is
    # viewed as [K, p, 2 + p] blocks when every par_id run is a whole block (the layout logistic_model emits);
    # dlsa_sum_blocks_f64 sums them.  A frame in any other row order is grouped on the host first (numpy, the frame's own
    # memory) and handed over as one block.  No torch arithmetic here: tensors are storage (north star).
    K = vals.shape[0] // p if p else 0
    whole = K >= 1 and vals.shape[0] == K * p and np.array_equal(par_id, np.tile(np.arange(p), K))
    if not whole:
        summed = np.zeros((p, 2 + p))
        np.add.at(summed, par_id, vals)
        vals, K = summed, 1
    blk = engine.rows_to_device(vals).view(K, p, 2 + p)
    coef, smc, sig = blk[:, :, 0].contiguous(), blk[:, :, 1].contiguous(), blk[:, :, 2:].contiguous()      # packed in HBM, not on the host
    # message layout [Sig_inv (p*p) | Sig_invMcoef (p) | coef (p)]
    return engine.sum_blocks(coef, smc, sig), names, p


def dlsa_mapred(model_mapped_sdf, num_partitions=None, comm=None):
    """MapReduce for partitioned data with a given model (dlsa.py:21-61).

    Accepts the device-resident `MappedBlocks` of `fit_logistic_partitions`, a pandas frame in the
    reference's stacked layout, or any object with `.toPandas()` (and optionally
    `.rdd.getNumPartitions()`) such as a Spark DataFrame.  `num_partitions` is the divisor of the
    one-shot mean and has the reference's meaning (dlsa.py:51-52: the number of partitions of the WHOLE
    job): given explicitly, or read from `.rdd.getNumPartitions()`, it is used as is on every rank.
    When it is not given the divisor is the number of blocks: this rank's count, summed over the ranks in
    the same all-reduce as the blocks.  In a torch.distributed job every rank passes ITS blocks; `comm` (an
    `engine.RcclComm`) carries the all-reduce through the C ABI's dlsa_allreduce_f64 instead (hosts without
    torch.distributed) -- the same RCCL collective on the same message either way."""
    if isinstance(model_mapped_sdf, MappedBlocks):
        mb = model_mapped_sdf
        names, p = mb.names, mb.coef.shape[1]
        msg = engine.sum_blocks(mb.coef, mb.Sig_invMcoef, mb.Sig_inv)
        nblocks = mb.num_partitions
    else:
        pdf = model_mapped_sdf
        spark_side_sum = False
        if not isinstance(pdf, pd.DataFrame):
            if num_partitions is None and hasattr(pdf, "rdd"):
                num_partitions = pdf.rdd.getNumPartitions()
            if hasattr(pdf, "groupby") and hasattr(pdf, "columns"):
                # A Spark DataFrame: the one-round sum runs where the blocks live, exactly the reference's calls (dlsa.py:30-34) --
                # p rows come back instead of K * p (config 3 at K = 200: 2 MB instead of 400 MB through Arrow)
                cols = list(pdf.columns)
                pdf = pdf.groupby("par_id").sum(*cols[1:]).toPandas().sort_values("par_id")
                pdf.columns = cols                      # 'sum(coef)' ... -> the block's own column names
                spark_side_sum = True
            else:
                pdf = pdf.toPandas()
        msg, names, p = _blocks_from_frame(pdf)
        nblocks = max(1, pdf.shape[0] // max(1, p))
        if spark_side_sum and num_partitions is None:
            raise ValueError("dlsa_mapred: a Spark-side sum needs num_partitions (or .rdd.getNumPartitions()) for the one-shot mean")
    counts = torch.tensor([float(nblocks)], dtype=torch.float64, device=msg.device)
    msg = distributed.allreduce_message(torch.cat([msg, counts]), comm=comm)
    K = float(msg[-1].item()) if num_partitions is None else float(num_partitions)
    Sig_inv_sum = msg[: p * p].view(p, p)
    Sig_invMcoef_sum = msg[p * p: p * p + p]
    # dlsa.py:48-49 lstsq(rcond=None): a Cholesky solve for an SPD sum; the minimum-norm least-squares solution when the
    # sum is singular (a dummy level present in no partition, models.py:84-91, or a collinear design)
    beta_byOLS, rank = engine.wls_solve(Sig_inv_sum, Sig_invMcoef_sum)
    if rank < p:
        warnings.warn("dlsa_mapred: the summed Sig_inv has rank %d < %d; beta_byOLS is the minimum-norm least-squares "
                      "solution (numpy.linalg.lstsq semantics)" % (rank, p))
    beta_byONESHOT = msg[p * p + p: p * p + 2 * p] / K                                  # dlsa.py:51-52
    out = torch.cat([beta_byOLS[:, None], beta_byONESHOT[:, None], Sig_inv_sum], 1).cpu().numpy()
    return pd.DataFrame(out, columns=["beta_byOLS", "beta_byONESHOT"] + list(names))


def dlsa(Sig_inv_, beta_, sample_size, fit_intercept=False, type="lar"):
    """Distributed Least Squares Approximation (dlsa.py:70-107): LARS path of the quadratic
    (theta - beta_)' Sig_inv_ (theta - beta_) and the path points minimising AIC and BIC.
    With an intercept the intercept estimate is beta0[idx] + beta_[0] (the rpy2 original,
    dlsa.py:97 comment).  Returns DataFrame {beta_byAIC, beta_byBIC}."""
    S = np.asarray(Sig_inv_, dtype=np.float64) if not isinstance(Sig_inv_, torch.Tensor) else Sig_inv_
    b = np.asarray(beta_, dtype=np.float64) if not isinstance(beta_, torch.Tensor) else beta_
    fit = lars_path_device(S, b, fit_intercept, sample_size, type=type)
    ia = int(np.argmin(fit["AIC"].cpu().numpy()))          # dlsa.py:87-91: argmin on the host (steps + 1 values)
    ib = int(np.argmin(fit["BIC"].cpu().numpy()))
    by_aic, by_bic = fit["beta"][ia], fit["beta"][ib]
    if fit_intercept:
        b0 = float(b[0])
        by_aic = torch.cat([(fit["beta0"][ia] + b0).reshape(1), by_aic])
        by_bic = torch.cat([(fit["beta0"][ib] + b0).reshape(1), by_bic])
    return pd.DataFrame({"beta_byAIC": by_aic.cpu().numpy(), "beta_byBIC": by_bic.cpu().numpy()})


# README.md:25-27 names
dlsa_mapreduce = dlsa_mapred
dlsa_fit = dlsa
