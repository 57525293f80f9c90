#!/bin/bash
# FETCH_SIZE / L2 hit counters of probe binaries: bench/probe/pmc_probe.sh "<args>" tag...
ARGS="$1"; shift; export TMPDIR=/tmp
for t in "$@"; do
  O=gpurun_out/pp_$t; mkdir -p $O
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/f -o f --output-format csv -- build/probe/probe_$t $ARGS > $O/f.log 2>&1
  rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum -d $O/t -o t --output-format csv -- build/probe/probe_$t $ARGS > $O/t.log 2>&1
  python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob("$O/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if 'gram_plan' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
print("PMC $t", {k: sum(v) / len(v) for k, v in agg.items()})
PY
done
