#!/bin/bash
OUT=gpurun_out/pmc_f32; mkdir -p $OUT; export TMPDIR=/tmp
run() { name=$1; shift; rocprofv3 --kernel-trace --pmc "$@" -d $OUT/$name -o $name --output-format csv -- python3 bench/gram_quick.py 6000000 2000 2 f32 > $OUT/$name.log 2>&1; }
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES
run tcc1 TCC_HIT_sum TCC_MISS_sum
run fetch FETCH_SIZE
run sq2 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU
python3 - <<PY
import csv, collections, glob
for f in sorted(glob.glob("$OUT/*/*_counter_collection.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'][:40]; agg[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[(k, r['Counter_Name'])] += 1
    for k, d in agg.items():
        if 'gram' in k:
            for c, v in sorted(d.items()): print('%-42s %-26s %.5g /dispatch (%d)' % (k, c, v / cnt[(k, c)], cnt[(k, c)]))
for f in glob.glob("$OUT/sq1/*_kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        if 'gram' in r['Kernel_Name']: print(r['Kernel_Name'][:40], 'ms', (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6)
PY
