#!/bin/bash
# per-kernel totals of one configuration's map fit: bench/fit_profile.sh C3   (writes gpurun_out/fitprof_<cfg>/)
CFG=${1:-C3}; OUT=gpurun_out/fitprof_$CFG; mkdir -p $OUT; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT -o fit --output-format csv -- python3 bench/bench_configs.py $CFG > $OUT/out.jsonl 2> $OUT/err.log
python3 - <<PY
import csv
rows = list(csv.DictReader(open("$OUT/fit_kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:22]:
    print("%-90s calls %5s  total %9.3f ms  avg %9.3f ms  %5.1f%%" % (r["Name"][:90], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e6, float(r["Percentage"])))
print("total kernel time %.1f ms" % (tot / 1e6))
PY
