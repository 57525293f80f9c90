#!/bin/bash
# kernel time vs wall time of the structured airline-shaped map fit: bench/oh_timeline.sh [rows] [K]
OUT=gpurun_out/oh_timeline; mkdir -p $OUT; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT -o tl --output-format csv -- python3 bench/oh_trace.py "$@" > $OUT/out.txt 2>&1
grep -E "^structured" $OUT/out.txt | tail -2
python3 - <<PY
import csv
rows = list(csv.DictReader(open("$OUT/tl_kernel_trace.csv")))
names = {}
for r in rows:
    k = r["Kernel_Name"].split("(")[0][:70]
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    names.setdefault(k, [0, 0.0]); names[k][0] += 1; names[k][1] += d
skip = ("multinomial", "randn", "distribution", "elementwise", "Fill", "index", "cumsum", "scan", "sort", "copy", "cat")
tot = sum(v[1] for k, v in names.items() if not any(s in k for s in skip))
print("kernel time (fit kernels) %.1f us over 4 fits = %.1f us per fit" % (tot, tot / 4))
for k, v in sorted(names.items(), key=lambda kv: -kv[1][1])[:18]:
    print("  %-70s calls/fit %6.1f  us/fit %8.1f  avg %7.1f us" % (k, v[0] / 4, v[1] / 4, v[1] / v[0]))
PY
