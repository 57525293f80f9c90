#!/usr/bin/env python3
"""bench.py -- rows/s through the X'WX Gram kernel at p=500 (BASELINE.json's metric).

  python bench.py --gpus N --steps K --warmup W
  (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Workload (config.workload): BASELINE config 3's per-GPU row shard -- synthetic Gaussian logistic
rows, 2.5e7 x 500 fp64 = 100 GB resident in HBM per GPU (weak scaling: every rank owns the rows
[rank*R, (rank+1)*R) of the same seeded stream).  One STEP = one pass of the weighted Gram
H = X' diag(w) X over the rank's shard (w = mu(1-mu) at the true coefficients) and, for N>1, the
algorithm's one-round communication: a single RCCL all-reduce of the [Sig_inv | Sig_inv.theta |
theta] message (p^2+2p doubles).  value = total rows of all ranks / max-over-ranks time.

The same JSON line carries `roofline` (Gram kernel vs the fp64 MFMA peak, HIP-event timed on
the launch stream) and `cpu_baseline` (the numpy oracle's Gram on the host cores, bounded
sample), plus `extra` with the HBM-bound logit pass and the end-to-end fit for reference.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FP64_MFMA_PEAK_TF = 78.6    # 256 CU x 4 SIMD x 32 flop/clk x 2.4 GHz (= fp32 vector 157.3 / 2)
HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--rows-per-gpu", type=int, default=25_000_000)
    ap.add_argument("--p", type=int, default=500)
    ap.add_argument("--seed", type=int, default=20260101)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true")
    ap.add_argument("--e2e", action="store_true",
                    help="also time the end-to-end fit (IRLS + combine + LARS); off by default so that every\n"
                         "gram_kernel launch of the default command has the benchmark's size (rocprof averages)")
    ap.add_argument("--cpu-sample-rows", type=int, default=400_000)
    return ap.parse_args()


def cpu_baseline(p, sample_rows, seed):
    """The oracle's Gram (numpy -> multithreaded BLAS dgemm, what models.py:130 runs) on a
    bounded sample of the same synthetic rows, timed on this box's host cores."""
    import numpy as np
    from oracle import dlsa_oracle as orc
    X = orc.synth_features(seed, 0, sample_rows, p, orc.SYNTH_GAUSSIAN)
    beta = orc.true_beta(p)
    w, _, _ = orc.logit_pass(X, np.zeros(sample_rows), beta)
    orc.gram(X[:20000], w[:20000])          # warm the BLAS threads
    reps, t_total = 0, 0.0
    while t_total < 10.0 and reps < 50:
        t0 = time.perf_counter()
        orc.gram(X, w)
        t_total += time.perf_counter() - t0
        reps += 1
    try:
        from threadpoolctl import threadpool_info
        cores = max([d.get("num_threads", 1) for d in threadpool_info()] + [1])
    except Exception:
        cores = os.cpu_count() or 1
    return {"value": sample_rows * reps / t_total, "unit": "rows/s", "cores": int(cores), "kind": "port",
            "sample": "oracle.gram (numpy/BLAS X'diag(w)X) on %d x %d fp64 synthetic Gaussian rows, %d passes, %.1f s"
                      % (sample_rows, p, reps, t_total)}


def main():
    args = parse()
    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    torch.cuda.set_device(local % max(1, torch.cuda.device_count()))
    dist = None
    if world > 1:
        import torch.distributed as dist_
        dist = dist_
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("DLSA_BENCH_BACKEND", "nccl")      # "gloo": lets two ranks share one GPU in a dry run
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    from dlsa_amd import engine

    p, R = args.p, args.rows_per_gpu
    free, _ = torch.cuda.mem_get_info()
    need = R * p * 8 * 1.08
    if need > free:
        R = int(free / 1.08 / (p * 8))
        print("[bench] shrinking rows-per-gpu to %d to fit %.0f GB free HBM" % (R, free / 1e9), file=sys.stderr)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- inputs resident in HBM before any timed region
    t_gen = time.perf_counter()
    X, y = engine.synth(args.seed, rank * R, R, p, kind=engine.SYNTH_GAUSSIAN, labels=True)
    beta_true = torch.zeros(p, dtype=torch.float64, device="cuda")
    beta_true[: int(p * 0.4)] = 1.0
    w, _, _ = engine.logit_pass(X, y, beta_true, want_g=False, want_loglik=False)
    torch.cuda.synchronize()
    t_gen = time.perf_counter() - t_gen

    msg = torch.zeros(p * p + 2 * p, dtype=torch.float64, device="cuda")
    H = msg[: p * p].view(p, p)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]

    def step(i=None):
        if i is not None:
            ev[i][0].record()
        engine.gram(X, w, out=H)
        if i is not None:
            ev[i][1].record()
        if dist is not None:
            dist.all_reduce(msg)

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    barrier()
    elapsed = time.perf_counter() - t0
    tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
    if dist is not None:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    elapsed = float(tmax.item())
    kern_ms = sum(a.elapsed_time(b) for a, b in ev) / args.steps

    out = None
    if rank == 0:
        rows_total = R * world * args.steps
        value = rows_total / elapsed
        flops_row = p * (p + 1) + p             # algorithmic: upper triangle outer product + w scaling
        bytes_row = 8 * (p + 1)                 # algorithmic: the X row + w_i
        ach_tf = R * flops_row / (kern_ms * 1e-3) / 1e12
        traffic, traffic_src = None, None
        try:    # HBM-side bytes per launch from the committed PMC pass (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE)
            prof = json.load(open(os.path.join(ROOT, "profiles", "pmc_latest.json")))
            assert prof.get("p", 500) == p
            kk = [k for k in prof["kernels"] if "gram_kernel<double" in k][0]
            per_row = (2.0 * prof["kernels"][kk]["FETCH_SIZE"] + prof["kernels"][kk]["WRITE_SIZE"]) * 1024.0 / prof["rows_per_gpu"]
            traffic, traffic_src = per_row * R, "profiles/pmc_latest.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, KB; FETCH x2 per MI355X_MICROARCH.md)"
        except Exception:
            pass
        out = {
            "metric": "rows/sec through X'WX kernel at p=%d" % p, "value": value, "unit": "rows/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "Logistic DLSA config 3 per-GPU row shard: synthetic Gaussian n=%d x p=%d fp64 "
                                   "per GPU (%.1f GB in HBM), weighted Gram X'WX pass%s" %
                                   (R, p, R * p * 8 / 1e9, " + 1 RCCL all-reduce of p^2+2p f64" if world > 1 else ""),
                       "rows_per_gpu": R, "p": p, "partitions_per_gpu": 1, "parallelism": "row-shards x%d" % world},
            "roofline": {"bound": "mfma", "achieved": ach_tf, "peak": FP64_MFMA_PEAK_TF, "unit": "TFLOP/s",
                         "frac": ach_tf / FP64_MFMA_PEAK_TF, "traffic": traffic, "traffic_unit": "bytes per launch",
                         "traffic_source": traffic_src, "algorithmic_bytes_per_launch": R * bytes_row,
                         "kernel": "dlsa::gram_kernel<double,true,2,0> (+gram_reduce_kernel, ~0.1 ms)", "kernel_ms": kern_ms,
                         "algorithmic_flops_per_row": flops_row, "algorithmic_bytes_per_row": bytes_row,
                         "hbm_GBps_algorithmic": R * bytes_row / (kern_ms * 1e-3) / 1e9,
                         "hbm_frac_of_8TBps": R * bytes_row / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
        }

    # ---- extras (rank 0, N=1 only): HBM-bound logit pass, end-to-end fit, config 2
    if rank == 0 and world == 1 and not args.no_extra:
        extra = {"gen_seconds": t_gen}
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        engine.logit_pass(X, y, beta_true)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(3):
            engine.logit_pass(X, y, beta_true)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 3
        extra["logit_pass"] = {"ms": ms, "rows_per_s": R / (ms * 1e-3),
                               "hbm_GBps": R * 8 * (p + 2) / (ms * 1e-3) / 1e9,
                               "hbm_frac_of_8TBps": R * 8 * (p + 2) / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
        if args.e2e:
            t1 = time.perf_counter()
            fit = engine.irls_fit(X, y, [0, R])
            msgv = engine.sum_blocks(fit["coef"], fit["Sig_invMcoef"], fit["Sig_inv"])
            S = msgv[: p * p].view(p, p)
            theta = engine.spd_solve(S, msgv[p * p: p * p + p])
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            path = engine.lars_path(S, theta, False, float(R))
            torch.cuda.synchronize()
            t3 = time.perf_counter()
            extra["end_to_end_fit"] = {"irls_iters": fit["n_iter"][0], "status": fit["status"][0],
                                       "map_plus_combine_s": t2 - t1, "lars_s": t3 - t2,
                                       "rows_per_s_whole_fit": R / (t3 - t1),
                                       "theta_err_vs_truth_linf": float((theta - beta_true).abs().max())}
            del fit
        out["extra"] = extra
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        del X, y, w
        torch.cuda.empty_cache()
        out["cpu_baseline"] = cpu_baseline(p, args.cpu_sample_rows, args.seed)
    elif rank == 0:
        out["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
