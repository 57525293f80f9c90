#!/bin/bash
# kernel time vs wall time of a map fit: bench/fit_timeline.sh n p [K]   (gpurun_out/fit_timeline/)
OUT=gpurun_out/fit_timeline; mkdir -p $OUT; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT -o tl --output-format csv -- python3 bench/irls_trace.py "$@" > $OUT/out.txt 2>&1
grep -E "^fit" $OUT/out.txt | tail -2
python3 - <<PY
import csv
rows = list(csv.DictReader(open("$OUT/tl_kernel_trace.csv")))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the LAST fit: from the last synth-free stretch; find the last 'irls' run: take kernels after the 4th-from-last fit boundary is hard: report totals / 5 fits
names = {}
for r in rows:
    k = r["Kernel_Name"].split("(")[0][:60]
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    names.setdefault(k, [0, 0.0]); names[k][0] += 1; names[k][1] += d
tot = sum(v[1] for k, v in names.items() if "synth" not in k)
print("kernel time excluding synth: %.1f us over 5 fits = %.1f us per fit" % (tot, tot / 5))
for k, v in sorted(names.items(), key=lambda kv: -kv[1][1])[:14]:
    print("  %-60s calls/fit %5.1f  us/fit %8.1f" % (k, v[0] / 5, v[1] / 5))
PY
