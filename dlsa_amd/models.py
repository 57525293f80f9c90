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
import os
import warnings

import numpy as np
import pandas as pd
import torch

from . import engine
from .design import DesignSpec


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


def _device_design(sample_df, Y_name, fit_intercept, dummy_info, dummy_factors_baseline, data_info, for_eval=False):
    """models.py:50-108,121-122 (and :159-206 for eval): the chunk's design matrix, built ON THE DEVICE.
    The host only does the column bookkeeping and turns categorical strings into level codes
    (design.DesignSpec); get_dummies / baseline drop / standardise / reindex / ones column are one
    dlsa_design_f64 launch.  Returns (X device tensor or None when the chunk must be skipped,
    names = [intercept,] usecols_x)."""
    spec = DesignSpec.from_reference(list(sample_df.columns), Y_name, fit_intercept, dummy_info,
                                     ([] if (for_eval and len(dummy_info) == 0) else dummy_factors_baseline), data_info)
    _, codes, unknown = spec.encode(sample_df, dummy_info, numeric=False)
    dev = torch.device("cuda")
    X, missing = spec.build(spec.numeric_to_device(sample_df, dev), torch.from_numpy(codes).to(dev))
    if missing or unknown:          # models.py:80-91 / :187-194
        shape = (len(sample_df), len(spec.names) - (1 if fit_intercept else 0) - len(missing))
        if not for_eval:
            warnings.warn("Dummies:" + str(set(missing)) + "missing in this data chunk " + str(shape)
                          + "Skip modeling this part of data.")
            return None, spec.names
        warnings.warn("Dummies:" + str(set(missing)) + "missing in this data chunk " + str(shape))
    return X, spec.names


def logistic_model(sample_df, Y_name, fit_intercept=False, dummy_info=[], dummy_factors_baseline=[],
                   data_info=[]):
    """Run the logistic model on one partition (a pandas frame), as the reference's GROUPED_MAP
    UDF does (models.py:42-147).  Returns the p x (3+p) frame `par_id, coef, Sig_invMcoef,
    [intercept,] <features>`; a chunk that lacks an expected dummy level returns the all-zero
    block with a warning (models.py:84-91)."""
    yd = torch.from_numpy(np.ascontiguousarray(sample_df[Y_name].to_numpy(dtype=np.float64))).cuda()
    r = None
    if len(dummy_info) > 0:
        # dummy path: fit on the raw numerics + level codes when the design qualifies (structured passes, the dense
        # one-hot matrix is never built); the missing-level check of models.py:80-91 runs on the codes
        spec = DesignSpec.from_reference(list(sample_df.columns), Y_name, fit_intercept, dummy_info,
                                         dummy_factors_baseline, data_info)
        plan = spec.onehot_plan()
        if plan is not None:
            names = spec.names
            _, codes, unknown = spec.encode(sample_df, dummy_info, numeric=False)
            missing = spec.missing_levels(codes)
            if missing or unknown:
                shape = (len(sample_df), len(names) - (1 if fit_intercept else 0) - len(missing))
                warnings.warn("Dummies:" + str(set(missing)) + "missing in this data chunk " + str(shape)
                              + "Skip modeling this part of data.")
                return pd.DataFrame(0, index=np.arange(len(names)), columns=["par_id", "coef", "Sig_invMcoef"] + names)
            r = engine.onehot_irls_fit(plan, spec.numeric_to_device(sample_df), torch.from_numpy(codes).cuda(), yd, [0, len(sample_df)])
    if r is None:
        Xd, names = _device_design(sample_df, Y_name, fit_intercept, dummy_info, dummy_factors_baseline, data_info)
        if Xd is None:
            return pd.DataFrame(0, index=np.arange(len(names)), columns=["par_id", "coef", "Sig_invMcoef"] + names)
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
                            tol=1e-13, max_iter=100, options=None, **option_fields):
    """Tensor fast path of the map step for MANY partitions of one device-resident shard.

    X [n, p] fp64 row-major on the GPU, y [n].  Either `part_offsets` (K+1 ints: partition k is the
    contiguous row range [off[k], off[k+1]) -- the layout `repartition(K, "partition_id")` gives,
    logistic_dlsa.py:295) or `partition_num` (systematic partition_id = i % K, models.py:33: partition k is
    the strided view X[k::K], nothing is gathered).  With fit_intercept the leading ones column of
    models.py:121-122 is implicit in the kernels (results have p + 1 columns, `intercept` first).
    `options` (an engine.IrlsOptions) or its fields as keyword arguments -- chains=, seeded=, fused=, predict=, subsample_div=, ... --
    set the IRLS driver's policy for this call (default: chosen from the shapes; same results to the parity tolerance either way).
    Returns MappedBlocks."""
    if not X.is_cuda:
        raise RuntimeError("fit_logistic_partitions runs on the GPU only (no CPU fallback)")
    if X.dtype != torch.float64:
        raise TypeError("fit_logistic_partitions: X must be float64 (the logistic path is fp64 end to end; "
                        "fit_linear_partitions takes fp32 rows), got %s" % X.dtype)
    y = y.to(torch.float64)                 # labels may arrive as integers / bools / fp32
    n, p = X.shape
    if names is None:
        names = ["x" + str(i) for i in range(p)]
    names = (["intercept"] if fit_intercept else []) + list(names)
    # No copy of the shard either way: partition_id = i % K is the strided view rows k, k + K, ... (row pitch ldx K), and the
    # intercept's ones column (models.py:121-122) is implicit in the kernels.  At config-3 scale (2.5e7 x 500 fp64 = 100 GB
    # of a 288 GB part) a gathered copy plus a [1 | X] copy would not fit.
    if part_offsets is None:
        K = int(partition_num) if partition_num else 1
        first = list(range(K))
        rows = [len(range(k, n, K)) for k in range(K)]
        step = K
    else:
        offs = [int(v) for v in part_offsets]
        first, rows, step = offs[:-1], [offs[k + 1] - offs[k] for k in range(len(offs) - 1)], 1
    with engine.irls_options(options, **option_fields):
        r = engine.irls_fit_ex(engine.row_major(X), y.contiguous(), first, rows, row_step=step, fit_intercept=fit_intercept,
                               tol=tol, max_iter=max_iter)
    return MappedBlocks(r["coef"], r["Sig_invMcoef"], r["Sig_inv"], names, r["status"], r["n_iter"], r["loglik"],
                        sample_size=n)


def fit_logistic_design(num, codes, y, spec, partition_num=None, part_offsets=None, structured=True, tol=1e-13,
                        max_iter=100, options=None, **option_fields):
    """Tensor fast path of the map step for a design given by its RAW columns: num [n, q] fp64 (columns in
    spec.numeric_cols order) and codes [n, f] int32 level codes (DesignSpec.encode), both on the GPU.
    With structured=True and a qualifying design (<= 8 dense columns, <= 8 factors; pair tables larger than LDS are cut into bands) the fit runs on
    the raw representation -- gather / histogram passes, the dense [n, p] matrix is never built (config 4: 76 B
    instead of 2080 B per row and pass); otherwise the matrix is built once by the design kernel and the dense
    kernels run.  Same MappedBlocks either way."""
    if not y.is_cuda:
        raise RuntimeError("fit_logistic_design runs on the GPU only (no CPU fallback)")
    y = y.to(torch.float64)
    if num is not None and num.dtype != torch.float64:
        raise TypeError("fit_logistic_design: num must be float64, got %s" % num.dtype)
    n = y.numel()
    # partition_id = i % K (models.py:33) as strided views: nothing is gathered (the structured fit takes (first, rows, step); the
    # dense one builds the matrix once and hands the same views to dlsa_irls_fit_ex_f64)
    if part_offsets is None:
        K = int(partition_num) if partition_num else 1
        first, rows, step = list(range(K)), [len(range(k, n, K)) for k in range(K)], K
    else:
        offs = [int(v) for v in part_offsets]
        first, rows, step = offs[:-1], [offs[k + 1] - offs[k] for k in range(len(offs) - 1)], 1
    plan = spec.onehot_plan() if structured else None
    with engine.irls_options(options, **option_fields):       # (the IRLS driver's policy for this call: see fit_logistic_partitions)
        if plan is not None:
            r = engine.onehot_irls_fit_ex(plan, engine.row_major(num) if num is not None else None,
                                          engine.row_major(codes) if codes is not None else None, y.contiguous(), first, rows,
                                          row_step=step, tol=tol, max_iter=max_iter)
        else:
            X, _ = spec.build(num, codes)
            r = engine.irls_fit_ex(X, y.contiguous(), first, rows, row_step=step, tol=tol, max_iter=max_iter)
    return MappedBlocks(r["coef"], r["Sig_invMcoef"], r["Sig_inv"], spec.names, r["status"], r["n_iter"], r["loglik"],
                        sample_size=n)


def logistic_model_eval(sample_df, Y_name, par, fit_intercept=False, dummy_info=[], dummy_factors_baseline=[],
                        data_info=[]):
    """Log-likelihood of every estimator column of `par` on one partition (models.py:151-225)."""
    Xd, _ = _device_design(sample_df, Y_name, fit_intercept, dummy_info, dummy_factors_baseline, data_info,
                           for_eval=True)
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
class _LinearBlock:
    """Sufficient statistics of one partition of a linear model, accumulated chunk by chunk ON THE DEVICE by library
    kernels only: H = [1 | X]'[1 | X] (fp64; the ones column of models.py:121-122 implicit), g = [1 | X]'y, y'y, n.
    fp64 rows: dlsa_gram_f64 (accumulate) straight into H's X-block; fp32 rows: dlsa_gram_f32_acc64 (fp32 MFMA passes,
    fp64 sum).  X'y, the column sums (the intercept's border), y'y and sum y: dlsa_xtv_stats_* -- one read of the chunk."""

    def __init__(self, p, fit_intercept, sig_out, smc_out, device):
        self.p, self.icpt = p, 1 if fit_intercept else 0
        self.sig, self.smc = sig_out, smc_out                      # [pp, pp] / [pp] views of the result block (fp64)
        self.HX = sig_out[self.icpt:, self.icpt:]                  # the X'X block, written in place (row pitch pp)
        self.g = torch.zeros((p,), dtype=torch.float64, device=device)
        self.colsum = torch.zeros((p,), dtype=torch.float64, device=device) if fit_intercept else None
        self.stats = torch.zeros((2,), dtype=torch.float64, device=device)
        self.rows = 0

    def add(self, Xc, yc):
        # (the X'y pass on a second stream next to the Gram of the same chunk was measured in round 4: 2.139 vs 2.141 s for config 5's
        # stream -- kernels on two streams do run side by side here, but take the sum of their times; bench/overlap_probe.py)
        first = self.rows == 0
        if Xc.dtype == torch.float32:
            engine.gram_acc64(Xc, None, out=self.HX, accumulate=not first)
        else:
            engine.gram(Xc, None, out=self.HX, accumulate=not first)
        engine.xtv_stats(Xc, yc, g=self.g, colsum=self.colsum, stats=self.stats, want_colsum=bool(self.icpt), accumulate=True)
        self.rows += Xc.shape[0]

    def finish(self):
        """Borders of the implicit ones column, X'y into the block; returns y'y (host float)."""
        st = self.stats.cpu().numpy()
        self.smc[self.icpt:] = self.g
        if self.icpt:
            self.sig[0, 0] = float(self.rows)
            self.sig[0, 1:] = self.colsum
            self.sig[1:, 0] = self.colsum
            self.smc[0] = float(st[1])
        return float(st[0])


def _linear_finish(blocks, coef, smc, sig, names, n):
    status, rss = [], []
    for k, blk in enumerate(blocks):
        if blk is None or blk.rows == 0:
            status.append(4); rss.append(0.0)
            continue
        yy = blk.finish()
        try:
            coef[k] = engine.spd_solve(sig[k], smc[k])
            status.append(0)
            rss.append(yy - float(np.dot(coef[k].cpu().numpy(), smc[k].cpu().numpy())))      # y'y - theta'X'y
        except Exception:
            status.append(2); rss.append(float("nan"))
            warnings.warn("linear_model: X'X not positive definite (collinear design)")
    return MappedBlocks(coef, smc, sig, names, status, [1] * len(blocks), rss, sample_size=n)


def fit_linear_partitions(X, y, partition_num=None, part_offsets=None, fit_intercept=False, names=None):
    """Tensor fast path: one Gram pass X'X + one X'y pass per partition, no copy of the shard -- partition_id = i % K
    (models.py:33) is the strided view X[k::K] (every kernel takes any row pitch), contiguous partitions are row ranges, and
    the intercept's ones column (models.py:121-122) stays implicit (its border comes from the X'y pass).  fp32 rows
    (config 5) run the fp32 MFMA Gram with fp64 slab sums; blocks are fp64 either way.  Returns MappedBlocks whose
    `loglik` slot carries the residual sum of squares of each partition."""
    if not X.is_cuda:
        raise RuntimeError("fit_linear_partitions runs on the GPU only (no CPU fallback)")
    if X.dtype not in (torch.float64, torch.float32):
        raise TypeError("fit_linear_partitions: X must be float64 or float32, got %s" % X.dtype)
    X = engine.row_major(X)
    y = y.to(X.dtype)
    n, p = X.shape
    if part_offsets is None:
        K = int(partition_num) if partition_num else 1
        views = [(X[k::K], y[k::K].contiguous() if K > 1 else y.contiguous()) for k in range(K)]
    else:
        offs = [int(v) for v in part_offsets]
        K = len(offs) - 1
        yc = y.contiguous()
        views = [(X[offs[k]:offs[k + 1]], yc[offs[k]:offs[k + 1]]) for k in range(K)]
    pp = p + (1 if fit_intercept else 0)
    if names is None:
        names = ["x" + str(i) for i in range(p)]
    names = (["intercept"] if fit_intercept else []) + list(names)
    coef = torch.zeros((K, pp), dtype=torch.float64, device=X.device)
    smc = torch.zeros((K, pp), dtype=torch.float64, device=X.device)
    sig = torch.zeros((K, pp, pp), dtype=torch.float64, device=X.device)
    blocks = []
    for k, (Xk, yk) in enumerate(views):
        if Xk.shape[0] == 0:
            blocks.append(None)
            continue
        blk = _LinearBlock(p, fit_intercept, sig[k], smc[k], X.device)
        blk.add(Xk, yk)
        blocks.append(blk)
    return _linear_finish(blocks, coef, smc, sig, names, n)


def fit_linear_streaming(n, p, partition_num=1, chunk_rows=1 << 22, seed=20260101, row0=0, kind="gaussian", sigma=1.0,
                         fit_intercept=False, dtype=torch.float32, names=None, device="cuda", on_chunk=None, overlap=False):
    """The linear map step for a shard that does NOT fit HBM (BASELINE config 5: 6.25e7 x 2000 fp32 = 500 GB per GPU, SURVEY
    8(d)): rows row0 .. row0 + n of the seeded stream are generated ON THE DEVICE chunk by chunk (dlsa_synth_*: a row is a
    pure function of (seed, i); y = X beta* + sigma N(0,1) by dlsa_synth_response_*), each chunk goes once through the Gram
    kernel (accumulating) and once through the X'y pass, and is overwritten by the next.  The K partitions are contiguous
    row ranges of the stream (the layout repartition(K, "partition_id") gives, logistic_dlsa.py:295); chunks never straddle
    a partition.  Peak memory = one chunk buffer + K blocks.  kind="gaussian32" (fp32 only) is the fp32-native stream: features and
    response in ONE launch (engine.synth_linear32), 8.5 ms per 2^22 x 2000 chunk where "gaussian" + synth_response take 41 -- config 5
    at its stated size 2.58 -> 2.11 s (97 -> 119 TF including generation).  overlap=True generates chunk i + 1 into a second buffer on
    a side stream while chunk i is consumed: the kernels do run side by side, but generator and Gram together take the SUM of their
    times (VALU work is not hidden under the fp32 matrix pipe on this chip, bench/overlap_probe.py: 120.1 + 41.2 -> 161.3 ms) for
    twice the chunk memory -- off by default.  Returns MappedBlocks (fp64 blocks)."""
    K = int(partition_num)
    n, p, chunk_rows = int(n), int(p), int(chunk_rows)
    if K < 1 or n < 0 or chunk_rows < 1:
        raise ValueError("fit_linear_streaming: need partition_num >= 1, n >= 0, chunk_rows >= 1")
    # kind "gaussian32": the fp32-native stream (engine.synth_linear32: features and response in ONE launch, a quarter of the
    # generator time of "gaussian" + synth_response at p = 2000) -- fp32 only
    native32 = isinstance(kind, str) and kind == "gaussian32"
    if native32 and dtype != torch.float32:
        raise ValueError("fit_linear_streaming: kind='gaussian32' is the fp32-native stream (dtype=torch.float32)")
    kind_id = 0 if native32 else ({"uniform": engine.SYNTH_UNIFORM, "gaussian": engine.SYNTH_GAUSSIAN}[kind] if isinstance(kind, str) else int(kind))
    pp = p + (1 if fit_intercept else 0)
    if names is None:
        names = ["x" + str(i) for i in range(p)]
    names = (["intercept"] if fit_intercept else []) + list(names)
    coef = torch.zeros((K, pp), dtype=torch.float64, device=device)
    smc = torch.zeros((K, pp), dtype=torch.float64, device=device)
    sig = torch.zeros((K, pp, pp), dtype=torch.float64, device=device)
    offs = [int(n * k / K) for k in range(K + 1)]
    rows_max = min(chunk_rows, max(1, max(offs[k + 1] - offs[k] for k in range(K))))
    chunks = [(k, r, min(rows_max, offs[k + 1] - r)) for k in range(K) for r in range(offs[k], offs[k + 1], rows_max)]
    # Two chunk buffers: chunk i + 1 is generated on a side stream while chunk i goes through the Gram and X'y kernels on the
    # caller's stream (generation is ~1/4 of a chunk's time at p = 2000: hidden instead of added).  HIP events order the hand-offs.
    nbuf = 2 if (overlap and len(chunks) > 1) else 1
    Xbuf = [engine.empty_rows(rows_max, p, dtype, device) for _ in range(nbuf)]
    ybuf = [torch.empty((rows_max,), dtype=dtype, device=device) for _ in range(nbuf)]
    main = torch.cuda.current_stream()
    side = torch.cuda.Stream() if nbuf == 2 else main
    ready = [torch.cuda.Event() for _ in range(nbuf)]
    done = [torch.cuda.Event() for _ in range(nbuf)]

    def generate(i):
        k, r, m = chunks[i]
        b = i % nbuf
        with torch.cuda.stream(side):
            if i >= nbuf:
                side.wait_event(done[b])             # the kernels that read this buffer two chunks ago have finished
            if native32:
                engine.synth_linear32(seed, row0 + r, m, p, sigma=sigma, out=Xbuf[b][:m], out_y=ybuf[b][:m])
            else:
                engine.synth(seed, row0 + r, m, p, kind=kind_id, labels=False, dtype=dtype, out=Xbuf[b][:m])
                engine.synth_response(seed, row0 + r, Xbuf[b][:m], sigma=sigma, out=ybuf[b][:m])
            ready[b].record(side)

    blocks = [None] * K
    if nbuf == 2:
        side.wait_stream(main)
    if chunks:
        generate(0)
    for i, (k, r, m) in enumerate(chunks):
        b = i % nbuf
        if i + 1 < len(chunks) and nbuf == 2:
            generate(i + 1)
        main.wait_event(ready[b])
        if blocks[k] is None:
            blocks[k] = _LinearBlock(p, fit_intercept, sig[k], smc[k], device)
        blocks[k].add(Xbuf[b][:m], ybuf[b][:m])
        done[b].record(main)
        if nbuf == 1 and i + 1 < len(chunks):
            generate(i + 1)
        if on_chunk is not None:
            on_chunk(k, r, m)
    if nbuf == 2:
        main.wait_stream(side)
    return _linear_finish(blocks, coef, smc, sig, names, n)


def fit_linear_chunks(chunks, p, partition_num=1, fit_intercept=False, dtype=None, names=None, device="cuda", sample_size=None):
    """The linear map step for rows that live OUTSIDE HBM (host memory, files read chunk by chunk: `dlsa_amd.ingest`):
    `chunks` yields `(k, X, y)` -- rows of partition k, X [m, p] and y [m] as numpy arrays or torch tensors, on the host
    (pinned memory makes the copy asynchronous) or already on the device, any m >= 1, any order of k.  Two device chunk
    buffers: the host -> HBM copy of chunk i + 1 runs on a copy stream (SDMA engines, no CU) while chunk i goes through the
    Gram kernel (accumulating) and the X'y pass on the caller's stream, so a PCIe-bound source costs max(copy, compute) per
    chunk instead of their sum.  A host chunk may be reused by the producer as soon as the NEXT item is requested (the copy
    out of it has completed by then).  Every chunk of one call has the same dtype (fp32 -> fp32 MFMA Gram summed in fp64,
    fp64 -> fp64 Gram); blocks are fp64.  Peak device memory = two buffers of the largest chunk + K blocks.
    Returns MappedBlocks (`loglik` slot = residual sum of squares per partition), as fit_linear_partitions does."""
    K = int(partition_num)
    p = int(p)
    if K < 1 or p < 1:
        raise ValueError("fit_linear_chunks: need partition_num >= 1 and p >= 1")
    if not torch.cuda.is_available():
        raise RuntimeError("fit_linear_chunks runs on the GPU only (no CPU fallback)")
    pp = p + (1 if fit_intercept else 0)
    if names is None:
        names = ["x" + str(i) for i in range(p)]
    names = (["intercept"] if fit_intercept else []) + list(names)
    coef = torch.zeros((K, pp), dtype=torch.float64, device=device)
    smc = torch.zeros((K, pp), dtype=torch.float64, device=device)
    sig = torch.zeros((K, pp, pp), dtype=torch.float64, device=device)
    blocks = [None] * K
    main = torch.cuda.current_stream()
    side = torch.cuda.Stream()
    Xbuf, ybuf = [None, None], [None, None]
    ready = [torch.cuda.Event(), torch.cuda.Event()]
    done = [None, None]
    rows_seen = 0
    side.wait_stream(main)
    for i, item in enumerate(chunks):
        k, Xc, yc = item
        k = int(k)
        if not 0 <= k < K:
            raise ValueError("fit_linear_chunks: partition index %d outside 0..%d" % (k, K - 1))
        Xc = torch.from_numpy(Xc) if isinstance(Xc, np.ndarray) else Xc
        yc = torch.from_numpy(yc) if isinstance(yc, np.ndarray) else yc
        if Xc.dim() != 2 or Xc.shape[1] != p or yc.dim() != 1 or yc.shape[0] != Xc.shape[0]:
            raise ValueError("fit_linear_chunks: chunk %d has X %s, y %s (need [m, %d] and [m])" % (i, tuple(Xc.shape), tuple(yc.shape), p))
        m = int(Xc.shape[0])
        if m == 0:
            continue
        if dtype is None:
            dtype = Xc.dtype if Xc.dtype in (torch.float32, torch.float64) else torch.float64
        b = i & 1
        if Xc.is_cuda and Xc.dtype == dtype and Xc.stride(1) == 1 and yc.is_cuda:
            # rows already in HBM: no staging copy (the caller keeps them alive until the call returns)
            Xd, yd = Xc, yc if yc.dtype == dtype else yc.to(dtype)
        else:
            with torch.cuda.stream(side):
                if done[b] is not None:
                    side.wait_event(done[b])             # the kernels that read this buffer two chunks ago have finished
                if Xbuf[b] is None or Xbuf[b].shape[0] < m:
                    if done[b] is not None:
                        done[b].synchronize()            # (growing a buffer: its old storage must be idle before it is freed)
                    Xbuf[b] = engine.empty_rows(m, p, dtype, device)
                    ybuf[b] = torch.empty((m,), dtype=dtype, device=device)
                Xbuf[b][:m].copy_(Xc, non_blocking=True)
                ybuf[b][:m].copy_(yc, non_blocking=True)
                ready[b].record(side)
            main.wait_event(ready[b])
            Xd, yd = Xbuf[b][:m], ybuf[b][:m]
        if blocks[k] is None:
            blocks[k] = _LinearBlock(p, fit_intercept, sig[k], smc[k], device)
        blocks[k].add(Xd, yd)
        done[b] = torch.cuda.Event()
        done[b].record(main)
        rows_seen += m
        if not Xc.is_cuda:
            ready[b].synchronize()                       # the producer may overwrite its host chunk from here on
    main.wait_stream(side)
    return _linear_finish(blocks, coef, smc, sig, names, rows_seen if sample_size is None else int(sample_size))


def linear_model(sample_df, Y_name, fit_intercept=False, dummy_info=[], dummy_factors_baseline=[], data_info=[]):
    """Frame-level sibling of logistic_model for a linear response: same arguments, same
    p x (3+p) output frame `par_id, coef, Sig_invMcoef, [intercept,] <features>`."""
    Xd, names = _device_design(sample_df, Y_name, fit_intercept, dummy_info, dummy_factors_baseline, data_info)
    if Xd is None:
        return pd.DataFrame(0, index=np.arange(len(names)), columns=["par_id", "coef", "Sig_invMcoef"] + names)
    yd = torch.from_numpy(np.ascontiguousarray(sample_df[Y_name].to_numpy(dtype=np.float64))).cuda()
    mb = fit_linear_partitions(Xd, yd, fit_intercept=False, names=names)      # the ones column is already in Xd
    out = mb.block_frame(0)
    if out.isna().values.any():
        warnings.warn("NAs appear in the final output")
    return out
