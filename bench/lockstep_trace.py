"""One lock-step fit under rocprofv3 --kernel-trace: the kernels of an iteration in launch order with their durations and the gaps
between them (where the time of an iteration goes besides the fused pass).
   rocprofv3 --kernel-trace -d gpurun_out/ls -o ls --output-format csv -- python3 bench/lockstep_trace.py run [K nk p]
   python3 bench/lockstep_trace.py show gpurun_out/ls/*/ls_kernel_trace.csv"""
import csv, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if sys.argv[1] == "run":
    import torch
    import dlsa_amd
    from dlsa_amd import engine
    K, nk, p = (int(float(v)) for v in (sys.argv[2:5] if len(sys.argv) >= 5 else (1000, 20000, 100)))
    X, y = engine.synth(20260101, 0, K * nk, p, kind=engine.SYNTH_GAUSSIAN)
    offs = [k * nk for k in range(K + 1)]
    for _ in range(3):
        torch.cuda.synchronize()
        mb = dlsa_amd.fit_logistic_partitions(X, y, part_offsets=offs, batched=True, small=False)
        torch.cuda.synchronize()
    print("iters", mb.n_iter[:3])
    if os.environ.get("LS_SINGLE"):          # the same rows through the single-partition form of the fused pass, for the per-row rate
        beta = mb.coef[0].clone()
        for _ in range(5):
            engine.irls_pass(X, y, beta)
        torch.cuda.synchronize()
else:
    rows = list(csv.DictReader(open(sys.argv[2])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # the last fit: from the last batch_restart / first batched pass after the last synth... simply the last N kernels after the last gather / memset gap
    names = [r["Kernel_Name"] for r in rows]
    last_unpack = max(i for i, n in enumerate(names) if "batch_update_kernel" in n)
    # walk back to the start of the last fit: a gap of > 2 ms between kernels ends a fit at these sizes? use the count of update kernels per fit instead
    upd = [i for i, n in enumerate(names) if "batch_update_kernel" in n]
    per_fit = len(upd) // 3
    first = upd[-per_fit]
    while first > 0 and "irls_pass" not in names[first]:
        first -= 1
    prev_end = None
    tot = {}
    for r in rows[first:last_unpack + 1]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        gap = (s - prev_end) / 1e3 if prev_end else 0.0
        nm = r["Kernel_Name"].split("(")[0][:60]
        print("%8.1f us  gap %7.1f us  %s" % ((e - s) / 1e3, gap, nm))
        t = tot.setdefault(nm, [0, 0.0]); t[0] += 1; t[1] += (e - s) / 1e3
        t = tot.setdefault("(gaps)", [0, 0.0]); t[0] += 1; t[1] += gap
        prev_end = e
    single = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows[last_unpack + 1:] if "irls_pass_narrow" in r["Kernel_Name"]]
    if single: print("---- single-partition form over the same rows: %s us" % ["%.1f" % v for v in single])
    print("---- totals of the last fit")
    for nm, (c, us) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
        print("%9.1f us  %4d x  %s" % (us, c, nm))
