#!/bin/bash
# FETCH/TCC counters of gram_quick at one shape with the CURRENT libdlsa_hip.so: bench/pmc_gram_quick.sh "<gram_quick args>" tag
ARGS="$1"; TAG=${2:-x}
OUT=gpurun_out/pmc_gq_$TAG; mkdir -p $OUT; export TMPDIR=/tmp
run() { name=$1; shift; rocprofv3 --kernel-trace --pmc "$@" -d $OUT/$name -o $name --output-format csv -- python3 bench/gram_quick.py $ARGS > $OUT/$name.log 2>&1; }
run fetch FETCH_SIZE
run tcc TCC_HIT_sum TCC_MISS_sum
run sq SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_LDS
python3 - <<PY
import csv, collections, glob
for f in sorted(glob.glob("$OUT/*/*_counter_collection.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'][:30]; agg[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[(k, r['Counter_Name'])] += 1
    for k, d in agg.items():
        if 'gram_kernel' in k:
            for c, v in sorted(d.items()): print('$TAG %-32s %-26s %.5g' % (k, c, v / cnt[(k, c)]))
PY
