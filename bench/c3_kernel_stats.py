"""Config 3's shard as 25 partitions under rocprofv3 --kernel-trace --stats: which kernels the map fit's time is in.
   rocprofv3 --kernel-trace --stats -d gpurun_out/c3s -o c3s --output-format csv -- python3 bench/c3_kernel_stats.py run
   python3 bench/c3_kernel_stats.py show gpurun_out/c3s/*/c3s_kernel_stats.csv"""
import csv, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if sys.argv[1] == "run":
    import torch
    from dlsa_amd import engine
    K, nk, p = 25, 1_000_000, 500
    X, y = engine.synth(20260101, 0, K * nk, p, kind=engine.SYNTH_GAUSSIAN)
    offs = [i * nk for i in range(K + 1)]
    for _ in range(4):
        engine.irls_fit(X, y, offs); torch.cuda.synchronize()
else:
    rows = list(csv.DictReader(open(sys.argv[2])))
    tot = sum(float(r["TotalDurationNs"]) for r in rows if "synth" not in r["Name"])
    print("kernel time of 4 fits (synth excluded): %.1f ms" % (tot / 1e6))
    for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:28]:
        print("%9.2f ms/fit  %6d calls/fit  avg %8.1f us  %s" % (float(r["TotalDurationNs"]) / 4e6, int(r["Calls"]) // 4, float(r["AverageNs"]) / 1e3, r["Name"][:90]))
