"""Kernel timeline of ONE fit (the last of several) under rocprofv3 --kernel-trace: every kernel in launch order with its duration
and the gap before it.   rocprofv3 --kernel-trace -d gpurun_out/ft -o ft --output-format csv -- python3 bench/fit_timeline.py run n p [K [chains]]      (chains = 1: one partition chain, no stream sharing)
   python3 bench/fit_timeline.py show gpurun_out/ft/*/ft_kernel_trace.csv"""
import csv, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if sys.argv[1] == "run":
    import torch
    from dlsa_amd import engine
    n, p = int(float(sys.argv[2])), int(sys.argv[3])
    K = int(sys.argv[4]) if len(sys.argv) > 4 else 1
    X, y = engine.synth(20260101, 0, n, p, kind=engine.SYNTH_GAUSSIAN)
    offs = [n * k // K for k in range(K + 1)]
    import contextlib, time
    opt = engine.irls_options(chains=int(sys.argv[5])) if len(sys.argv) > 5 else contextlib.nullcontext()
    with opt:
        for _ in range(4):
            t = time.perf_counter(); engine.irls_fit(X, y, offs); torch.cuda.synchronize()
            print("fit %.2f ms (under the tracer)" % ((time.perf_counter() - t) * 1e3), file=sys.stderr)
        marker = torch.zeros(7, device="cuda") + 1.0; torch.cuda.synchronize()       # (a torch kernel separates the fits in the trace)
        engine.irls_fit(X, y, offs); torch.cuda.synchronize()
else:
    rows = sorted(csv.DictReader(open((sys.argv[2] if os.path.exists(sys.argv[2]) else __import__("glob").glob(sys.argv[2])[0]))), key=lambda r: int(r["Start_Timestamp"]))
    last_torch = max(i for i, r in enumerate(rows) if "dlsa" not in r["Kernel_Name"] and "rocclr" not in r["Kernel_Name"])
    rows = rows[last_torch + 1:]
    prev = None
    tot = {}
    t0 = int(rows[0]["Start_Timestamp"])
    for r in rows:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        nm = r["Kernel_Name"].split("(")[0][:64]
        gap = (s - prev) / 1e3 if prev else 0.0
        print("%9.1f us  at %9.1f  gap %7.1f  %s" % ((e - s) / 1e3, (s - t0) / 1e3, gap, nm))
        t = tot.setdefault(nm, [0, 0.0]); t[0] += 1; t[1] += (e - s) / 1e3
        if gap > 0: g = tot.setdefault("(gaps)", [0, 0.0]); g[0] += 1; g[1] += gap
        prev = max(prev or 0, e)
    print("---- span %.1f us" % ((prev - t0) / 1e3))
    for nm, (c, us) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
        print("%9.1f us %4d x  %s" % (us, c, nm))
