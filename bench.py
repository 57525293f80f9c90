#!/usr/bin/env python3
"""bench.py -- rows/s through the X'WX Gram kernel at p=500 (BASELINE.json's metric).

  python bench.py --gpus N --steps K --warmup W

N = 1 runs in this process.  N > 1 needs one process per GPU: when the script is started WITHOUT a
torch.distributed environment (no WORLD_SIZE) it launches the N ranks itself -- before anything in this
process touches the GPU -- as
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py ...
and relays rank 0's JSON line; started under torch.distributed.run (the driver's form) it is one of the ranks and
insists that WORLD_SIZE == --gpus.

Workload (config.workload): BASELINE config 3's per-GPU row shard -- synthetic Gaussian logistic rows,
2.5e7 x 500 fp64 = 100 GB resident in HBM per GPU (weak scaling: every rank owns the rows [rank*R, (rank+1)*R) of the
same seeded stream).  One STEP = one pass of the weighted Gram H = X' diag(w) X over the rank's shard (w = mu(1-mu)
at the true coefficients) and, for N>1, the algorithm's one round of communication (reference dlsa/dlsa.py:30-34):
a single all-reduce (RCCL over xGMI) of the [Sig_inv | Sig_inv.theta | theta] message (p^2+2p doubles).
value = total rows of all ranks / max-over-ranks time.

The same JSON line carries `roofline` (Gram kernel vs the fp64 MFMA peak, HIP-event timed on the launch stream),
`allreduce` (the collective, timed inside the steps and on its own), `cpu_baseline` (the numpy oracle of the hot
path on the host cores, bounded sample; N=1 only) and `extra` (the HBM-bound logit pass, optionally the whole fit).

Order of an N = 1 run: generate the shard -> W + K steps (`value`) -> the same launch until the GPU has been busy for
--sustain-seconds (`sustained_block`) -> the CPU baseline in freshly spawned interpreters -> the K steps once more
(`second_block`).  N > 1 starts with a PREFLIGHT per rank (devices, free HBM, communicator and one 8-byte all-reduce under
60 s watchdogs: a one-line reason and a non-zero exit, never a hang); `--preflight` runs only that.
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FP64_MFMA_PEAK_TF = 78.6    # 256 CU x 4 SIMD x 32 flop/clk x 2.4 GHz (= fp32 vector 157.3 / 2)
HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: 8.0 TB/s spec
STRONG_MIN_ROWS = 65536     # rows per launch below which the library leaves the cyclic p = 500 Gram kernel (gram.hip)


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--rows-per-gpu", type=int, default=25_000_000)
    ap.add_argument("--p", type=int, default=500)
    ap.add_argument("--seed", type=int, default=20260101)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true")
    ap.add_argument("--e2e", action="store_true", help="(default since round 6; kept so that older command lines still parse)")
    ap.add_argument("--no-e2e", action="store_true",
                    help="leave out the end-to-end fit legs (`extra.end_to_end_fit`): every gram_cyclic_kernel launch of the command then\n"
                         "has the benchmark's size, which is what bench/profile_round.sh wants for rocprofv3's per-kernel averages")
    ap.add_argument("--warmup-cap-seconds", type=float, default=5.0,
                    help="after the W warm-up steps keep launching untimed steps until two consecutive launches agree within 2 %%,\n"
                         "for at most this long (0: exactly W): a process started behind one that released tens of GB runs its first\n"
                         "seconds slower (lab notes r05 section 7); the extra launches are counted in `warmup_extra_steps`")
    ap.add_argument("--scaling", choices=("weak", "strong", "both"), default="both",
                    help="N > 1: `value` is always the weak-scaling figure (fixed rows per GPU); `strong` adds a leg with the\n"
                         "TOTAL rows fixed at --rows-per-gpu, split evenly over the ranks (SURVEY 8(d) scaling report)")
    ap.add_argument("--e2e-partitions", type=int, default=25,
                    help="partitions per rank of the end-to-end fit (25 x 1e6 rows: logistic_dlsa.py:170 on config 3's shard)")
    ap.add_argument("--preflight", action="store_true",
                    help="only the N-rank preflight (device count, free HBM for the shard, RCCL communicator + one 8-byte all-reduce, each\n"
                         "under a 60 s watchdog): exit code 0 and one line per rank, or non-zero with the reason -- never hangs")
    ap.add_argument("--sustain-seconds", type=float, default=12.0,
                    help="after the K timed steps keep launching the same Gram pass until the GPU has been busy this long (N = 1): the\n"
                         "driver's utilisation sampler sees the run, and the line carries the sustained ms_per_step")
    ap.add_argument("--watchdog-seconds", type=float, default=900.0,
                    help="a rank that is still running after this long prints every thread's Python stack and leaves with exit code 1\n"
                         "(a hang becomes a reason; the default run takes ~2.5 min at N = 1 with the CPU baseline, ~1 min at N = 8); 0: off")
    ap.add_argument("--cpu-rows-per-partition", type=int, default=0, help="0 = sized by oracle/cpu_baseline.py")
    ap.add_argument("--cpu-gram-rows", type=int, default=400_000)
    return ap.parse_args(argv)


# --------------------------------------------------------------------------------------------------
# N > 1 without a torch.distributed environment: start the ranks (this process never touches the GPU)
# --------------------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def rank_threads(ncpu, nranks):
    """Host threads a rank may use for OpenMP / torch intra-op work: the node's cores shared out over the ranks, at most 8 (the
    library's own partition chains are <= 4 std::threads per rank that sleep in hipStreamSynchronize: 8 ranks x 4 on any node)."""
    return max(1, min(8, int(ncpu) // max(1, int(nranks))))


def launch(args):
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: RCCL's intra-node transport needs it here
    env.setdefault("OMP_NUM_THREADS", str(rank_threads(os.cpu_count() or 8, args.gpus)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    print("[bench] launching %d ranks: %s" % (args.gpus, " ".join(cmd)), file=sys.stderr)
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in proc.stdout.splitlines():
        s = ln.strip()
        if s.startswith("{") and '"metric"' in s:
            line = s
        else:
            print(ln, file=sys.stderr)
    if args.preflight:                 # no JSON line: the ranks' own exit codes and one-line reasons are the result
        return proc.returncode
    if proc.returncode != 0 or line is None:
        print("[bench] the %d-rank run failed (exit code %d, %s JSON line)" %
              (args.gpus, proc.returncode, "no" if line is None else "a"), file=sys.stderr)
        return proc.returncode or 1
    out = json.loads(line)
    if out.get("n_gpus") != args.gpus:
        print("[bench] rank 0 reported n_gpus=%r, expected %d" % (out.get("n_gpus"), args.gpus), file=sys.stderr)
        return 1
    print(line)
    return 0


# --------------------------------------------------------------------------------------------------
def _sha16(path):
    with open(path, "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()[:16]


GRAM_SOURCES = ("gram.hip", "gram_cyclic.hip", "gram_cyclic_asm.inc", "common.h")      # what the p = 500 Gram launch is built from


def gram_sources_sha16():
    h = hashlib.sha256()
    for f in GRAM_SOURCES:
        with open(os.path.join(ROOT, "dlsa_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def traffic_from_profile(p, R):
    """HBM-side bytes per Gram launch from the committed PMC passes (profiles/pmc_latest.json: FETCH_SIZE x2, the
    gfx950 correction of MI355X_MICROARCH.md section HBM, + WRITE_SIZE, both in KB), scaled to R rows.  The profile
    records the hash of the kernel sources it was taken from: a profile of different sources gives null."""
    try:
        prof = json.load(open(os.path.join(ROOT, "profiles", "pmc_latest.json")))
        if prof.get("p", 500) != p:
            return None, "profiles/pmc_latest.json is for p=%s" % prof.get("p")
        have = gram_sources_sha16()
        if prof.get("gram_hip_sha16") != have:
            return None, "stale: profiles/pmc_latest.json was taken from Gram sources %s, the tree has %s" % (
                prof.get("gram_hip_sha16"), have)
        kk = [k for k in prof["kernels"] if ("gram_cyclic_kernel" in k or "gram_kernel<double" in k) and "FETCH_SIZE" in prof["kernels"][k]][0]
        per_row = (2.0 * prof["kernels"][kk]["FETCH_SIZE"] + prof["kernels"][kk]["WRITE_SIZE"]) * 1024.0 / prof["rows_per_gpu"]
        return per_row * R, ("profiles/pmc_latest.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes in KB; FETCH x2 per "
                             "MI355X_MICROARCH.md; Gram sources %s)" % have)
    except Exception as e:
        return None, "unavailable: %r" % (e,)


def _watchdog(seconds, what, rank):
    """Run-once timer: if `what` has not finished in time the rank says so and leaves with a non-zero code (no hang)."""
    import threading

    def fire():
        print("[bench] preflight FAILED on rank %d: %s did not finish within %.0f s" % (rank, what, seconds), file=sys.stderr, flush=True)
        os._exit(3)
    t = threading.Timer(seconds, fire)
    t.daemon = True
    t.start()
    return t


def preflight(args, rank, world, local, timeout_s=60.0):
    """What the first N-GPU run could trip over, checked up front with a one-line reason and a non-zero exit instead of a hang:
    enough devices, enough free HBM for this rank's shard, the communicator (RCCL unless DLSA_BENCH_BACKEND=gloo) and ONE 8-byte
    all-reduce.  Returns (torch.distributed or None, backend)."""
    import datetime
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("[bench] preflight FAILED on rank %d: no GPU visible" % rank)
    ndev = torch.cuda.device_count()
    backend = os.environ.get("DLSA_BENCH_BACKEND", "nccl")      # "gloo": lets ranks share one GPU in a dry run
    if backend == "nccl" and ndev < world:
        raise SystemExit("[bench] preflight FAILED on rank %d: %d ranks over RCCL need %d GPUs, this node shows %d "
                         "(DLSA_BENCH_BACKEND=gloo lets ranks share a GPU for a dry run)" % (rank, world, world, ndev))
    torch.cuda.set_device(local % max(1, ndev))
    sharing = max(1, (world + ndev - 1) // max(1, ndev))
    free, total = torch.cuda.mem_get_info()
    need = args.rows_per_gpu * args.p * 8 * 1.08
    if backend == "nccl" and need > free:
        raise SystemExit("[bench] preflight FAILED on rank %d: the shard needs %.1f GB of HBM, device %d has %.1f GB free of %.1f "
                         "(another process on the GPU?)" % (rank, need / 1e9, local, free / 1e9, total / 1e9))
    dist = None
    if world > 1:
        import torch.distributed as dist_
        dist = dist_
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        wd = _watchdog(timeout_s, "init_process_group(%s)" % backend, rank)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local),
                                    timeout=datetime.timedelta(seconds=timeout_s))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world, timeout=datetime.timedelta(seconds=timeout_s))
        wd.cancel()
        wd = _watchdog(timeout_s, "the first all-reduce (8 bytes)", rank)
        one = torch.ones(1, dtype=torch.float64, device="cuda")
        dist.all_reduce(one)
        torch.cuda.synchronize()
        wd.cancel()
        if int(one.item()) != world or dist.get_world_size() != world:
            raise SystemExit("[bench] preflight FAILED on rank %d: the 8-byte all-reduce returned %r over a communicator of %d ranks, expected %d"
                             % (rank, one.item(), dist.get_world_size(), world))
    print("[bench] preflight ok: rank %d/%d device %d of %d, %.1f GB free, backend %s%s" % (
        rank, world, local % max(1, ndev), ndev, free / 1e9, backend if world > 1 else "-", " (%d ranks per GPU)" % sharing if sharing > 1 else ""),
        file=sys.stderr, flush=True)
    if args.preflight and dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return dist, backend


def worker(args):
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d: start it as `python bench.py --gpus N` (it launches the "
                         "ranks) or under torch.distributed.run with --nproc-per-node equal to --gpus" % (args.gpus, world))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC (RCCL's intra-node transport needs it on this pool); set before torch loads
    if args.watchdog_seconds > 0:
        import faulthandler
        faulthandler.enable()
        sys.stderr.write("")                      # (faulthandler keeps the file descriptor)
        faulthandler.dump_traceback_later(args.watchdog_seconds, exit=True)
    p = args.p
    cpu = None        # the CPU baseline runs AFTER the first GPU block (its pool uses spawned interpreters: safe once the GPU is up)

    import torch
    torch.set_num_threads(rank_threads(os.cpu_count() or 8, world))
    dist, backend = None, None
    if world > 1 or args.preflight:
        dist, backend = preflight(args, rank, world, local)
        if args.preflight:
            return 0
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    ndev = torch.cuda.device_count()
    torch.cuda.set_device(local % max(1, ndev))
    from dlsa_amd import engine

    R = args.rows_per_gpu
    sharing = max(1, (world + ndev - 1) // max(1, ndev))        # ranks per GPU (1 except in the gloo dry run)
    free, _ = torch.cuda.mem_get_info()
    need = R * p * 8 * 1.08
    if need > free / sharing:
        R = int(free / sharing / 1.08 / (p * 8))
        print("[bench] shrinking rows-per-gpu to %d to fit %.0f GB free HBM" % (R, free / 1e9), file=sys.stderr)
    if dist is not None:            # every rank must own the same number of rows (weak scaling)
        rmin = torch.tensor([R], dtype=torch.int64, device="cuda")
        dist.all_reduce(rmin, op=dist.ReduceOp.MIN)
        R = int(rmin.item())

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
    mk = lambda: torch.cuda.Event(enable_timing=True)

    warm_info = {}

    def timed_steps(Xs, ws, steps, warmup, settle_cap_s=0.0):
        """W untimed + K timed steps of (Gram pass over Xs + the all-reduce), bracketed by barrier + synchronize on both
        sides; returns (max-over-ranks wall seconds, mean Gram ms, mean all-reduce ms, this rank's per-step Gram ms).
        settle_cap_s > 0: after the W warm-up steps more untimed steps follow until two consecutive launches agree within
        2 % ON EVERY RANK (one 16-byte all-reduce per extra step keeps the ranks' step counts equal), for at most that long."""
        ev = [(mk(), mk(), mk()) for _ in range(steps)]

        def step(i=None):
            if i is not None:
                ev[i][0].record()
            engine.gram(Xs, ws, out=H)
            if i is not None:
                ev[i][1].record()
            if dist is not None:
                dist.all_reduce(msg)
            if i is not None:
                ev[i][2].record()

        for _ in range(warmup):
            step()
        if settle_cap_s > 0:
            t_w, prev, trail = time.perf_counter(), None, []
            while True:
                a, b = mk(), mk()
                a.record()
                step()
                b.record()
                torch.cuda.synchronize()
                ms = a.elapsed_time(b)
                trail.append(ms)
                unstable = 1.0 if (prev is None or abs(ms - prev) > 0.02 * min(ms, prev)) else 0.0
                prev = ms
                flag = torch.tensor([unstable, time.perf_counter() - t_w], dtype=torch.float64, device="cuda")
                if dist is not None:
                    dist.all_reduce(flag, op=dist.ReduceOp.MAX)
                unstable_any, waited = (float(v) for v in flag.tolist())
                if unstable_any == 0.0 or waited >= settle_cap_s or len(trail) >= 200:
                    break
            warm_info.update({"warmup_extra_steps": len(trail), "warmup_extra_seconds": time.perf_counter() - t_w,
                              "warmup_settled": unstable_any == 0.0, "warmup_extra_step_ms": [round(v, 3) for v in trail[:12]]})
        barrier()
        t0 = time.perf_counter()
        for i in range(steps):
            step(i)
        barrier()
        el = time.perf_counter() - t0
        tmax = torch.tensor([el], dtype=torch.float64, device="cuda")
        if dist is not None:
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        per_step = [a.elapsed_time(b) for a, b, _ in ev]
        return (float(tmax.item()), sum(per_step) / steps,
                sum(b.elapsed_time(c) for _, b, c in ev) / steps, per_step)

    def per_rank(v):
        """min / max / mean over the ranks of a per-rank scalar (skew between the GPUs)."""
        t = torch.tensor([v], dtype=torch.float64, device="cuda")
        if dist is None:
            return {"min": v, "max": v, "mean": v}
        allv = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(allv, t)
        vals = [float(x.item()) for x in allv]
        return {"min": min(vals), "max": max(vals), "mean": sum(vals) / len(vals), "per_rank": vals}

    elapsed, kern_ms, comm_ms, step_ms = timed_steps(X, w, args.steps, args.warmup, settle_cap_s=args.warmup_cap_seconds)
    last_ms = step_ms[-1]
    median_ms = sorted(step_ms)[len(step_ms) // 2]
    first_ranks = per_rank(step_ms[0])
    # what the library dispatched, and the clock the chip held in the last timed launch: shader cycles of wave 0 of
    # workgroup 0 (s_memtime delta written by the kernel) / that launch's HIP-event time
    kernel_name, cycles = engine.gram_last_kernel(want_cycles=True)
    clock_ghz = cycles / (last_ms * 1e-3) / 1e9 if cycles else None
    kern_ranks = per_rank(kern_ms)
    clock_ranks = per_rank(clock_ghz or 0.0)
    if world > 1 and rank == 0 and kern_ranks["min"] > 0 and kern_ranks["max"] / kern_ranks["min"] > 1.05:
        slow = max(range(world), key=lambda r: kern_ranks["per_rank"][r])
        print("[bench] kernel time differs by %.1f %% over the ranks: rank %d is the slowest (%.2f ms, shader clock %.2f GHz; fastest %.2f ms)"
              % ((kern_ranks["max"] / kern_ranks["min"] - 1) * 100, slow, kern_ranks["max"], clock_ranks["per_rank"][slow], kern_ranks["min"]),
              file=sys.stderr, flush=True)

    # ---- N = 1: the same launches again until the GPU has been busy for --sustain-seconds (the driver's utilisation sampler sees the
    # run; the rate a long job holds), THEN the CPU baseline (fresh spawned interpreters, the GPU idle), then the K steps a second time
    sustained = second_block = None
    if world == 1 and args.sustain_seconds > 0:
        nsus = int(max(0.0, args.sustain_seconds - elapsed) / max(1e-6, kern_ms * 1e-3))
        if nsus > 0:
            s_el, s_kern, _, _ = timed_steps(X, w, nsus, 0)
            sustained = {"steps": nsus, "seconds": s_el, "ms_per_step": s_el / nsus * 1e3, "kernel_ms": s_kern, "value": R * nsus / s_el}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # a REPORTED reference point: its failure (a worker, the pool's barrier, a missing gcc / liboracle_synth, host memory)
        # must not cost the primary GPU line
        try:
            from oracle import cpu_baseline
            cpu = cpu_baseline.run(p, args.seed, rows_per_partition=args.cpu_rows_per_partition or None,
                                   gram_rows=args.cpu_gram_rows)
        except BaseException as e:      # incl. SystemExit / BrokenBarrierError from the pool
            if isinstance(e, KeyboardInterrupt):
                raise
            print("[bench] cpu_baseline failed: %r" % (e,), file=sys.stderr)
            cpu = {"error": repr(e)}
        b_el, b_kern, _, _ = timed_steps(X, w, args.steps, 1)
        second_block = {"steps": args.steps, "ms_per_step": b_el / args.steps * 1e3, "kernel_ms": b_kern, "value": R * args.steps / b_el,
                        "note": "the same K steps once more after the CPU baseline (GPU cold again): the line's `value` is the FIRST block"}

    # ---- strong scaling leg (N > 1): the SAME total rows as one GPU's shard, split evenly over the ranks
    strong = None
    if dist is not None and args.scaling in ("strong", "both") and R // world < STRONG_MIN_ROWS:
        # fewer rows per rank than the p = 500 Gram kernel of the weak-scaling line takes: another kernel would run, and the leg would
        # compare kernels, not GPUs -- say so instead of changing kernels silently
        strong = {"scaling": "strong", "skipped": "rows_per_gpu / N = %d is under the %d rows per launch the metric's kernel needs "
                                                  "(gram_cyclic_kernel); run with a larger --rows-per-gpu" % (R // world, STRONG_MIN_ROWS)}
        if rank == 0:
            print("[bench] strong-scaling leg skipped: " + strong["skipped"], file=sys.stderr, flush=True)
    elif dist is not None and args.scaling in ("strong", "both"):
        Rs = R // world
        s_el, s_kern, s_comm, _ = timed_steps(X[:Rs], w[:Rs], args.steps, args.warmup)
        s_name, _ = engine.gram_last_kernel()
        strong = {"scaling": "strong", "total_rows": Rs * world, "rows_per_gpu": Rs, "value": Rs * world * args.steps / s_el,
                  "unit": "rows/s", "ms_per_step": s_el / args.steps * 1e3, "kernel_ms": per_rank(s_kern),
                  "allreduce_ms_in_step": s_comm, "kernel": s_name,
                  "note": "every rank takes the first total_rows / N rows of its own resident shard (same distribution, "
                          "same seeded stream); speed-up vs N=1 = value / the N=1 run's value"}

    # ---- the collective on its own (all ranks): 20 back-to-back all-reduces of the message
    allreduce = None
    if dist is not None:
        reps = 20
        for _ in range(3):
            dist.all_reduce(msg)
        barrier()
        t1 = time.perf_counter()
        for _ in range(reps):
            dist.all_reduce(msg)
        barrier()
        iso = torch.tensor([(time.perf_counter() - t1) / reps * 1e3], dtype=torch.float64, device="cuda")
        dist.all_reduce(iso, op=dist.ReduceOp.MAX)
        nbytes = msg.numel() * 8
        allreduce = {"backend": "rccl" if backend == "nccl" else backend, "ranks": world, "payload_bytes": nbytes,
                     "ms_in_step": comm_ms, "ms_isolated": float(iso.item()),
                     "busbw_GBps_isolated": 2.0 * (world - 1) / world * nbytes / (float(iso.item()) * 1e-3) / 1e9}

    out = None
    if rank == 0:
        rows_total = R * world * args.steps
        value = rows_total / elapsed
        flops_row = p * (p + 1) + p             # algorithmic: upper triangle outer product + w scaling
        bytes_row = 8 * (p + 1)                 # algorithmic: the X row + w_i
        ach_tf = R * flops_row / (kern_ms * 1e-3) / 1e12
        traffic, traffic_src = traffic_from_profile(p, R)
        out = {
            "metric": "rows/sec through X'WX kernel at p=%d" % p, "value": value, "unit": "rows/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            # ranks of the COMMUNICATOR whose all-reduce the preflight verified (sum of ones == ranks), not the environment's word for it
            "rccl_ranks": (dist.get_world_size() if dist is not None else 1) if backend == "nccl" else 0,
            "comm_ranks": dist.get_world_size() if dist is not None else 1,
            "value_per_gpu": value / world,          # weak scaling: compare with the N = 1 line's value
            # guards against a slow start landing in `value`: the untimed steps beyond W, and the first timed
            # launch against the median one (kernel time on the launch stream, HIP events)
            "warmup_extra_steps": warm_info.get("warmup_extra_steps", 0), "warmup_extra_seconds": warm_info.get("warmup_extra_seconds", 0.0),
            "warmup_settled": warm_info.get("warmup_settled"), "warmup_extra_step_ms": warm_info.get("warmup_extra_step_ms"),
            "first_step_ms": step_ms[0], "median_step_ms": median_ms, "first_over_median": step_ms[0] / median_ms,
            "first_step_ms_over_ranks": first_ranks if world > 1 else None,
            "sustained_block": sustained, "second_block": second_block,
            "config": {"workload": "Logistic DLSA config 3 per-GPU row shard: synthetic Gaussian n=%d x p=%d fp64 "
                                   "per GPU (%.1f GB in HBM), weighted Gram X'WX pass%s" %
                                   (R, p, R * p * 8 / 1e9, " + 1 all-reduce of p^2+2p f64 (%s)" %
                                    ("RCCL" if backend == "nccl" else backend) if world > 1 else ""),
                       "rows_per_gpu": R, "p": p, "partitions_per_gpu": 1, "parallelism": "row-shards x%d" % world},
            "roofline": {"bound": "mfma", "achieved": ach_tf, "peak": FP64_MFMA_PEAK_TF, "unit": "TFLOP/s",
                         "frac": ach_tf / FP64_MFMA_PEAK_TF, "traffic": traffic, "traffic_unit": "bytes per launch",
                         "traffic_source": traffic_src, "algorithmic_bytes_per_launch": R * bytes_row,
                         # what the library dispatched in the timed launches (dlsa_gram_last_kernel), not a guess from p
                         "kernel": "dlsa::%s (+gram_reduce_kernel, ~0.03 ms)" % kernel_name, "kernel_ms": kern_ms,
                         "kernel_ms_over_ranks": kern_ranks,
                         # the clock the chip held (DVFS): shader cycles of wave 0 of workgroup 0 in the last timed launch /
                         # that launch's HIP-event time; the peak assumes 2.4 GHz, so frac <= clock / 2.4
                         "shader_clock_GHz": clock_ghz, "shader_clock_GHz_over_ranks": clock_ranks if world > 1 else None,
                         "frac_at_sustained_clock": (ach_tf / (FP64_MFMA_PEAK_TF * clock_ghz / 2.4)) if clock_ghz else None,
                         "algorithmic_flops_per_row": flops_row, "algorithmic_bytes_per_row": bytes_row,
                         "hbm_GBps_algorithmic": R * bytes_row / (kern_ms * 1e-3) / 1e9,
                         "hbm_frac_of_8TBps": R * bytes_row / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
            "allreduce": allreduce,
            "strong_scaling": strong,
        }

    # ---- extras: the HBM-bound logit pass (rank 0, N = 1), the end-to-end fit legs (every rank takes part when N > 1; --no-e2e leaves them out)
    extra = {"gen_seconds": t_gen}
    if rank == 0 and world == 1 and not args.no_extra:
        e0, e1 = mk(), mk()
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
    if not args.no_e2e and not args.no_extra:
        # The whole path on every rank's shard (the C3 geometry of logistic_dlsa.py:170: partitions of 1e6 rows): per-partition
        # exact-MLE fits + Hessians -> local block sum -> ONE all-reduce of p^2 + 2p + 1 doubles -> WLS solve + LARS on every
        # rank (redundant: cheaper than a broadcast).  Wall time = max over ranks, barrier-bracketed.  A rank whose fit raises does
        # not leave the others waiting in the collective: the ranks agree on an "ok" flag first, and the leg is dropped with the reason.
        def iters(v):
            return {"min": min(v), "max": max(v), "mean": sum(v) / len(v)}

        def all_ok(ok):
            f = torch.tensor([1.0 if ok else 0.0], dtype=torch.float64, device="cuda")
            if dist is not None:
                dist.all_reduce(f, op=dist.ReduceOp.MIN)
            return float(f.item()) == 1.0

        def whole_path(Kp):
            offs = [int(R * k / Kp) for k in range(Kp + 1)]
            barrier()
            t1 = time.perf_counter()
            fit, err = None, None
            try:
                fit = engine.irls_fit(X, y, offs)
                fit_path = engine.irls_last_fit_path()
                msgv = torch.cat([engine.sum_blocks(fit["coef"], fit["Sig_invMcoef"], fit["Sig_inv"]),
                                  torch.tensor([float(Kp)], dtype=torch.float64, device="cuda")])
                torch.cuda.synchronize()
            except Exception as e:
                err = repr(e)
            t_map = time.perf_counter()
            if not all_ok(err is None):
                return {"error": err or "another rank's fit failed", "partitions_per_rank": Kp}
            if dist is not None:
                dist.all_reduce(msgv)
            S = msgv[: p * p].view(p, p)
            theta, wls_rank = engine.wls_solve(S, msgv[p * p: p * p + p].contiguous())
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            path = engine.lars_path(S, theta, False, float(R * world))
            barrier()
            t3 = time.perf_counter()
            tt = torch.tensor([t_map - t1, t2 - t_map, t3 - t2, t3 - t1], dtype=torch.float64, device="cuda")
            if dist is not None:
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            tt = [float(v) for v in tt.tolist()]
            return {"ranks": world, "partitions_per_rank": Kp, "partitions_total": int(round(float(msgv[-1].item()))),
                    "rows_per_partition": R // Kp, "irls_iters_rank0": fit["n_iter"], "passes_per_partition": iters(fit["n_iter"]),
                    "fit_path": fit_path, "status_ok": all(v == 0 for v in fit["status"]),
                    "map_s": tt[0], "reduce_plus_wls_s": tt[1], "lars_s": tt[2], "total_s": tt[3],
                    "rows_per_s_map": R * world / tt[0], "rows_per_s_whole_fit": R * world / tt[3], "wls_rank": wls_rank,
                    "lars_steps": int(path["beta"].shape[0]) - 1,
                    "theta_err_vs_truth_linf": float((theta - beta_true).abs().max())}

        def best_of(Kp, reps):
            """One untimed call (workspace, chains' streams, the first touch of the rows by these kernels), then `reps` timed
            ones: the record of the fastest map step, with every call's map seconds beside it."""
            first = whole_path(Kp)
            if "error" in first:
                return first
            runs = [whole_path(Kp) for _ in range(reps)]
            if any("error" in r for r in runs):
                return [r for r in runs if "error" in r][0]
            best = min(runs, key=lambda r: r["map_s"])
            best["map_s_all_calls"] = [first["map_s"]] + [r["map_s"] for r in runs]
            best["note"] = "fastest of %d calls after one untimed call; map_s = per-partition exact-MLE fits + Hessians + local block sum" % reps
            return best

        Kp = max(1, min(args.e2e_partitions, R // 1000))
        e2e = best_of(Kp, 2)
        extra["end_to_end_fit"] = e2e
        if world == 1 and "error" not in e2e:
            # north_star's literal geometry, one shard = one partition per GPU (dlsa_irls_last_fit_path = 0: chains)
            try:
                e2e["k1"] = best_of(1, 2)
            except Exception as e:
                e2e["k1"] = {"error": repr(e)}
            # MANY partitions of a narrow design: the lock-step driver (csrc/irls_batch.hip) on 1000 x 2e4 x 100 (16 GB beside the shard)
            try:
                import dlsa_amd
                Kl, nl, pl = 1000, 20000, 100
                free_now, _ = torch.cuda.mem_get_info()
                if free_now < 1.5 * Kl * nl * pl * 8:
                    raise RuntimeError("%.1f GB free beside the shard, the lock-step leg needs 24" % (free_now / 1e9))
                Xl, yl = engine.synth(args.seed + 1, 0, Kl * nl, pl, kind=engine.SYNTH_GAUSSIAN)
                offl = [k * nl for k in range(Kl + 1)]
                dlsa_amd.fit_logistic_partitions(Xl, yl, part_offsets=offl)
                ts = []
                for _ in range(3):
                    torch.cuda.synchronize()
                    t = time.perf_counter()
                    mb = dlsa_amd.fit_logistic_partitions(Xl, yl, part_offsets=offl)
                    torch.cuda.synchronize()
                    ts.append(time.perf_counter() - t)
                e2e["lock_step"] = {"partitions": Kl, "rows_per_partition": nl, "p": pl, "fit_s": min(ts), "fit_s_all_calls": ts,
                                    "fit_path": engine.irls_last_fit_path(), "passes_per_partition": iters(list(mb.n_iter)),
                                    "status_ok": all(v == 0 for v in mb.status), "rows_per_s": Kl * nl / min(ts),
                                    "note": "dlsa_amd.fit_logistic_partitions (models.py:110-131 for every partition), fit_path 2 = lock step; "
                                            "passes = Newton passes over all rows (the pooled start and the gradient-only passes come before them)"}
                del Xl, yl, mb
            except Exception as e:
                e2e["lock_step"] = {"error": repr(e)}
    if rank == 0:
        out["extra"] = extra
    if rank == 0:
        out["cpu_baseline"] = cpu
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch(args)
    return worker(args)


if __name__ == "__main__":
    sys.exit(main())
