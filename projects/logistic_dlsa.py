#! /usr/bin/env python3
"""Counterpart of the reference's driver script (projects/logistic_dlsa.py) for GPU row-shards.

Same call sequence and result objects as the reference driver --
    map (logistic_model per partition, :305-316) -> dlsa_mapred (:337) -> dlsa (:341-344)
    -> out_par columns beta_byAIC, beta_byBIC, beta_byOLS, beta_byONESHOT (:353-355)
    -> log-likelihood evaluation (:357-363) -> out_time table (:393-407)
    -> pickle [Sig_inv_beta, out_dlsa, out_par, out_model_eval, out_time] (:411-412)
-- but the Spark partitions are contiguous row ranges of a device-resident shard and the reduce is
one all-reduce.  Settings are module-level constants in the reference (:66-175); here they are
command-line flags with the reference's "simulated_pdf" defaults (n=1e5, K=20, p=200).  `--csv FILE...` is the reference's
real-data branch (:100-175, 218-237): read the CSV(s), keep usecols_x + [Y], drop NAs, binarise Y > 0, select / load the dummy
levels (dummy_keep_top, 000_OTHERS), load / compute the standardisation table, partition_id = row % ceil(n / 1e6), and fit the
dummy design from level codes on the device (the one-hot matrix is never built on the host; the airline defaults are built in).

  python projects/logistic_dlsa.py                                   # 1 GPU
  python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 projects/logistic_dlsa.py ...
"""
import argparse
import os
import pickle
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np          # noqa: E402
import pandas as pd         # noqa: E402
import torch                # noqa: E402

import dlsa_amd             # noqa: E402
from dlsa_amd import distributed, engine  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sample-size", type=int, default=100000)      # sample_size_sub, reference :90
    ap.add_argument("--partition-num", type=int, default=20)        # partition_num_sub, reference :89
    ap.add_argument("--p", type=int, default=200)                   # reference :92
    ap.add_argument("--fit-intercept", action="store_true")         # reference :80 (True for the airline data)
    ap.add_argument("--gaussian", action="store_true", help="N(0,1/12) features instead of U(-0.5,0.5)")
    ap.add_argument("--seed", type=int, default=20260101)
    ap.add_argument("--csv", nargs="+", default=[], help="real-data mode: CSV file(s) with a header (reference file_path, :107)")
    ap.add_argument("--y-name", default="", help="response column, binarised as Y > 0 (default ArrDelay, :172)")
    ap.add_argument("--usecols-x", default="", help="comma-separated feature columns (default: the airline usecols_x, :109-110)")
    ap.add_argument("--dummy-columns", default=None, help="comma-separated categorical columns (default: the airline dummy_columns, :156)")
    ap.add_argument("--dummy-keep-top", default="", help="comma-separated cumulative shares kept per dummy column (default 1,1,0.8,0.9,0.9, :164)")
    ap.add_argument("--dummy-info", default="", help="pickle of the dummy_info dictionary: loaded when it exists, else created and saved there (:142-153)")
    ap.add_argument("--data-info", default="", help="CSV of describe() (count / mean / stddev rows): loaded when it exists, else created and saved (:263-275)")
    ap.add_argument("--sample-size-per-partition", type=int, default=1000000)      # reference :169
    ap.add_argument("--save", default="", help="pickle path for [Sig_inv_beta, out_dlsa, out_par, out_model_eval, out_time]")
    ap.add_argument("--coef-csv", default="", help="write the coefficient table Var, MLE, DLSA_AIC, DLSA_BIC, WLSE, ONE_SHOT "
                    "(projects/results/plot_coef.py:43-51); the MLE column is the global fit of all rows as one partition per rank, combined")
    args = ap.parse_args()

    rank, world = distributed.init_from_env()
    tic_init = time.perf_counter()
    tictoc = {}
    n, K, p = args.sample_size, args.partition_num, args.p
    Y_name = "label"

    if args.csv:
        # ---- real data (reference :218-237, 263-275): host parsing, then level codes + numerics on the device
        from dlsa_amd import dummies, ingest
        Y_name = args.y_name or ingest.AIRLINE_Y
        usecols_x = args.usecols_x.split(",") if args.usecols_x else list(ingest.AIRLINE_USECOLS_X)
        dummy_columns = (args.dummy_columns.split(",") if args.dummy_columns else []) if args.dummy_columns is not None \
            else [c for c in ingest.AIRLINE_DUMMY_COLUMNS if c in usecols_x]
        keep_top = [float(v) for v in args.dummy_keep_top.split(",")] if args.dummy_keep_top else [1, 1, 0.8, 0.9, 0.9][:len(dummy_columns)]
        if len(keep_top) != len(dummy_columns):
            raise SystemExit("--dummy-keep-top needs one share per dummy column")
        tictoc["repartition"] = [time.perf_counter()]
        frames = [ingest.read_csv_frame(os.path.expanduser(f), usecols_x, Y_name, dummy_columns) for f in args.csv]
        pdf = frames[0] if len(frames) == 1 else pd.concat(frames, ignore_index=True)
        del frames
        n = len(pdf)
        if dummy_columns:
            path = os.path.expanduser(args.dummy_info) if args.dummy_info else ""
            # every rank decides by itself: the table is a function of the input files, so a rank that finds rank 0's freshly written
            # copy (os.replace: complete or absent) loads what it would have computed
            if path and os.path.exists(path):
                with open(path, "rb") as f:
                    dummy_info = pickle.load(f)
            else:
                dummy_info = dummies.select_dummy_factors(dummies.dummy_factors_counts(pdf, dummy_columns), keep_top, "000_OTHERS",
                                                          pickle_file=(path if (path and rank == 0) else None))
            # baselines when fitting the intercept (:158-162): the folded level where there is one, else the first selected level
            baseline = [(c + "_000_OTHERS") if len(dummy_info["factor_dropped"][c]) > 0 else sorted(dummy_info["factor_selected_names"][c])[0]
                        for c in dummy_columns] if args.fit_intercept else []
        else:
            dummy_info, baseline = [], []
        numeric_cols = [c for c in usecols_x if c not in dummy_columns]
        path = os.path.expanduser(args.data_info) if args.data_info else ""
        if path and os.path.exists(path):
            data_info = pd.read_csv(path)
        else:
            data_info = ingest.data_info_from_frame(pdf, numeric_cols)
            if path and rank == 0:                     # (renamed into place: a rank that finds the table finds all of it)
                data_info.to_csv(path + ".tmp.%d" % os.getpid(), index=False)
                os.replace(path + ".tmp.%d" % os.getpid(), path)
        sh = ingest.shard_from_frame(pdf, Y_name, dummy_info, baseline, data_info, args.fit_intercept,
                                     sample_size_per_partition=args.sample_size_per_partition, world=world, rank=rank)
        del pdf
        K, spec = sh["partition_num"], sh["spec"]
        p = len(spec.names) - int(args.fit_intercept)
        torch.cuda.synchronize()
        tictoc["repartition"].append(time.perf_counter())
        memsize_total = n * (len(usecols_x) + 2) * 8
        tictoc["mapred"] = [time.perf_counter()]
        mapped = dlsa_amd.fit_logistic_design(sh["num"], sh["codes"], sh["y"], spec, part_offsets=sh["part_offsets"])
        bad = [s_ for s_ in mapped.status if s_ != 0]
        if bad and rank == 0:
            print("warning: partitions with status", mapped.status)
        names = list(spec.names)

        def evaluate(out_par):
            # model_eval.py:10-42 on the device: the design matrix of a row chunk (intercept column included) and one pass per 8 estimators
            par = torch.from_numpy(out_par.to_numpy(dtype=np.float64)).cuda()
            tot = torch.zeros(par.shape[1], dtype=torch.float64, device="cuda")
            m = int(sh["y"].numel())
            plan = spec.onehot_plan()
            if plan is not None and m > 0:
                # a qualifying dummy design: one structured logit pass per estimator on the raw numerics + level codes (no matrix)
                for c0 in range(par.shape[1]):
                    _, _, ll_c = engine.onehot_logit_pass(plan, sh["num"] if sh["num"].shape[1] else None, sh["codes"], sh["y"],
                                                          par[:, c0].contiguous(), want_w=False, want_g=False)
                    tot[c0] = ll_c.reshape(-1)[0]
                return distributed.allreduce_message(tot)
            for a in range(0, m, 1 << 21):
                b = min(m, a + (1 << 21))
                Xc, _ = spec.build(sh["num"][a:b] if sh["num"] is not None else None, sh["codes"][a:b])
                for c0 in range(0, par.shape[1], 8):
                    tot[c0:c0 + 8] += engine.loglik(Xc, sh["y"][a:b].contiguous(), par[:, c0:c0 + 8].contiguous())
            return distributed.allreduce_message(tot)

        def global_fit():
            return dlsa_amd.fit_logistic_design(sh["num"], sh["codes"], sh["y"], spec, part_offsets=[0, int(sh["y"].numel())])
    else:
        # ---- this rank's partitions: k % world == rank, each a contiguous row range (repartition, :295)
        tictoc["repartition"] = [time.perf_counter()]
        mine = distributed.owned_partitions(K, world, rank)
        sizes = [len(range(k, n, K)) for k in mine]                     # partition_id = i % K  (models.py:33)
        offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
        n_local = int(offs[-1])
        kind = engine.SYNTH_GAUSSIAN if args.gaussian else engine.SYNTH_UNIFORM
        # rows of partition k are the global rows k, k+K, k+2K, ...: the stream is generated in chunks of a multiple of K rows, so inside
        # a chunk partition k is the strided view Xc[k::K] -- one strided device copy per owned partition moves it behind the
        # partition's earlier rows (no index arithmetic on the device: tensors for storage and copies only)
        Xl = torch.empty((n_local, p), dtype=torch.float64, device="cuda")
        yl = torch.empty((n_local,), dtype=torch.float64, device="cuda")
        fill = [int(o) for o in offs[:-1]]
        chunk = max(K, (1 << 22) // max(1, p) // K * K)
        for r0 in range(0, n, chunk):
            m = min(chunk, n - r0)
            Xc, yc = engine.synth(args.seed, r0, m, p, kind=kind)
            for j, k in enumerate(mine):
                cnt = len(range(k, m, K))                                  # r0 is a multiple of K
                Xl[fill[j]: fill[j] + cnt].copy_(Xc[k::K])
                yl[fill[j]: fill[j] + cnt].copy_(yc[k::K])
                fill[j] += cnt
        torch.cuda.synchronize()
        tictoc["repartition"].append(time.perf_counter())
        memsize_total = n * (p + 2) * 8

        # ---- map + reduce (dlsa part 1)
        tictoc["mapred"] = [time.perf_counter()]
        names = ["x" + str(i) for i in range(p)]
        mapped = dlsa_amd.fit_logistic_partitions(Xl, yl, part_offsets=offs, fit_intercept=args.fit_intercept, names=names)
        bad = [s for s in mapped.status if s != 0]
        if bad and rank == 0:
            print("warning: partitions with status", mapped.status)

        def evaluate(out_par):
            return dlsa_amd.loglik_partitions(Xl, yl, out_par, fit_intercept=args.fit_intercept)      # model_eval.py:10-42, one all-reduce

        def global_fit():
            return dlsa_amd.fit_logistic_partitions(Xl, yl, part_offsets=[0, n_local], fit_intercept=args.fit_intercept, names=names)

    Sig_inv_beta = dlsa_amd.dlsa_mapred(mapped, num_partitions=K)      # K = partitions of the whole job (dlsa.py:51-52)
    torch.cuda.synchronize()
    tictoc["mapred"].append(time.perf_counter())

    # ---- shrinkage on every rank (dlsa part 2)
    tictoc["dlsa"] = [time.perf_counter()]
    out_dlsa = dlsa_amd.dlsa(Sig_inv_=Sig_inv_beta.iloc[:, 2:], beta_=Sig_inv_beta["beta_byOLS"],
                             sample_size=n, fit_intercept=args.fit_intercept)
    tictoc["dlsa"].append(time.perf_counter())
    tictoc["model_fit"] = [tic_init, time.perf_counter()]

    # ---- model evaluation: total log-likelihood of each estimator column (model_eval.py:10-42)
    tictoc["model_eval"] = [time.perf_counter()]
    out_par = out_dlsa.copy()
    out_par["beta_byOLS"] = Sig_inv_beta["beta_byOLS"]
    out_par["beta_byONESHOT"] = Sig_inv_beta["beta_byONESHOT"]
    ll = evaluate(out_par)
    out_model_eval = pd.DataFrame({c: [float(v)] for c, v in zip(out_par.columns, ll.cpu().numpy())})
    tictoc["model_eval"].append(time.perf_counter())

    time_mapred = tictoc["mapred"][1] - tictoc["mapred"][0]
    time_dlsa = tictoc["dlsa"][1] - tictoc["dlsa"][0]
    time_model_fit = tictoc["model_fit"][1] - tictoc["model_fit"][0]
    from dlsa_amd import results
    out_time = results.time_table(n, K, p + int(args.fit_intercept), memsize_total, tictoc["repartition"][1] - tictoc["repartition"][0],
                                  time_mapred, time_dlsa, time_model_fit, tictoc["model_eval"][1] - tictoc["model_eval"][0])
    if args.coef_csv:
        # the table's MLE column (plot_coef.py:20-41 takes it from a separate global fit): every rank fits ITS rows as one
        # partition, the WLS combine of those fits is the global estimate up to O(1/n^2)
        g = dlsa_amd.dlsa_mapred(global_fit())
        if rank == 0:
            results.write_coef_csv(args.coef_csv, out_par, list(Sig_inv_beta.columns[2:]), beta_byMLE=g["beta_byOLS"].to_numpy())
    if rank == 0:
        if args.save:
            results.save_results(args.save, Sig_inv_beta, out_dlsa, out_par, out_model_eval, out_time)
            print("Model results are saved to:\t" + args.save)
        print("\nModel Summary:\n")
        print(out_time.to_string(index=False))
        print("\nModel Evaluation:")
        print("\tlog likelihood:\n")
        print(out_model_eval.to_string(index=False))
        print("\nDLSA Coefficients:\n")
        print(out_par.head(12).to_string())
    if distributed.is_distributed():
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
