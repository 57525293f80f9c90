#!/bin/bash
# PMC passes of the logit pass at one shape: bench/pmc_logit.sh rows p
ROWS=${1:-10000000}; P=${2:-100}
OUT=gpurun_out/pmc_logit_$P; mkdir -p $OUT; export TMPDIR=/tmp
run() { name=$1; shift; rocprofv3 --kernel-trace --pmc "$@" -d $OUT/$name -o $name --output-format csv -- python3 bench/logit_one.py $ROWS $P > $OUT/$name.log 2>&1; }
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES
run sq2 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM
run sq3 SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_INSTS_FLAT SQ_BUSY_CU_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT
python3 - <<PY
import csv, collections, glob
for f in sorted(glob.glob("$OUT/*/*_counter_collection.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'][:34]; agg[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[(k, r['Counter_Name'])] += 1
    for k, d in agg.items():
        if 'logit_kernel' in k:
            for c, v in sorted(d.items()): print('%-36s %-24s %.5g' % (k, c, v / cnt[(k, c)]))
for f in glob.glob("$OUT/sq1/*_kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        if 'logit_kernel' in r['Kernel_Name']: print('logit_kernel ms', (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6)
PY
