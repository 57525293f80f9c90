#!/bin/bash
# average duration of qn_step_kernel by workgroup size (DLSA_QN_THREADS), from rocprofv3 kernel stats
export TMPDIR=/tmp
for cfg in "1e7 100 10" "1.4e7 260 14" "2.5e7 500 25"; do
  for t in 1024 512 256; do
    OUT=gpurun_out/qn_ab; rm -rf $OUT; mkdir -p $OUT
    DLSA_QN_THREADS=$t rocprofv3 --kernel-trace --stats -d $OUT -o q --output-format csv -- python3 bench/irls_trace.py $cfg > $OUT/out.txt 2>&1
    echo -n "cfg=$cfg threads=$t: $(grep '^fit' $OUT/out.txt | tail -1)  qn_step avg us: "
    python3 -c "
import csv
for r in csv.DictReader(open('$OUT/q_kernel_stats.csv')):
    if 'qn_step' in r['Name']: print(round(float(r['AverageNs'])/1e3,1), 'calls', r['Calls'])
"
  done
done
