#!/bin/bash
# PMC passes for the Gram kernel (separate rocprofv3 runs per counter group; kernel-trace only).
# usage: bench/pmc_gram.sh <outdir> [rows]
OUT=${1:-gpurun_out/pmc}
ROWS=${2:-5000000}
mkdir -p $OUT
export TMPDIR=/tmp
run() { name=$1; shift; rocprofv3 --kernel-trace --pmc "$@" -d $OUT/$name -o $name --output-format csv -- python3 bench.py --steps 2 --warmup 1 --rows-per-gpu $ROWS --no-cpu-baseline --no-extra > $OUT/$name.log 2>&1; }
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES
run sq2 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD
run tcc1 TCC_HIT_sum TCC_MISS_sum
run tcc2 FETCH_SIZE
run tcc3 WRITE_SIZE
python3 - <<PY
import csv, collections, glob, os
for f in sorted(glob.glob("$OUT/*/*_counter_collection.csv")):
    rows = list(csv.DictReader(open(f)))
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in rows:
        k = r['Kernel_Name'][:48]; agg[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[(k, r['Counter_Name'])] += 1
    for k, d in agg.items():
        if 'gram_kernel' in k or 'logit_kernel' in k:
            for c, v in sorted(d.items()): print('%-50s %-26s %.5g /dispatch (%d)' % (k, c, v / cnt[(k, c)], cnt[(k, c)]))
for f in sorted(glob.glob("$OUT/sq1/*_kernel_trace.csv")):
    rows = list(csv.DictReader(open(f)))
    for r in rows:
        if 'gram_kernel' in r['Kernel_Name']: print('gram_kernel duration ms', (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6)
PY
