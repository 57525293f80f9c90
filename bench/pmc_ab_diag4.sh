#!/bin/bash
# MFMA-pipe counters of the narrow Gram for the shipped library and the -DDLSA_NARROW_DIAG4=0 variant: bench/pmc_ab_diag4.sh "p ..." 
OUT=gpurun_out/pmc_ab_diag4; mkdir -p $OUT; export TMPDIR=/tmp
cp dlsa_amd/libdlsa_hip.so /tmp/libdlsa_orig.so
for v in orig nodiag4; do
  [ $v = orig ] && cp /tmp/libdlsa_orig.so dlsa_amd/libdlsa_hip.so || cp build/var/libdlsa_$v.so dlsa_amd/libdlsa_hip.so
  for p in $1; do
    rows=10000000; [ $p -le 64 ] && rows=20000000
    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT -d $OUT/${v}_sq_$p -o sq --output-format csv -- python3 bench/gram_quick.py $rows $p 3 > $OUT/${v}_sq_$p.log 2>&1
    rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE -d $OUT/${v}_grbm_$p -o grbm --output-format csv -- python3 bench/gram_quick.py $rows $p 3 > $OUT/${v}_grbm_$p.log 2>&1
  done
done
cp /tmp/libdlsa_orig.so dlsa_amd/libdlsa_hip.so
python3 - <<PY
import csv, collections, glob, os
for d in sorted(glob.glob("$OUT/*_*_*")):
    if not os.path.isdir(d): continue
    for f in glob.glob(d + "/*_counter_collection.csv"):
        agg = collections.defaultdict(float); cnt = collections.Counter()
        for r in csv.DictReader(open(f)):
            if 'gram_narrow' in r['Kernel_Name']:
                agg[r['Counter_Name']] += float(r['Counter_Value']); cnt[r['Counter_Name']] += 1
        dur = []
        for g in glob.glob(d + "/*_kernel_trace.csv"):
            for r in csv.DictReader(open(g)):
                if 'gram_narrow' in r['Kernel_Name']: dur.append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6)
        print(os.path.basename(d), "ms %.3f" % (sorted(dur)[len(dur)//2] if dur else -1), " ".join("%s=%.5g" % (c, v / cnt[c]) for c, v in sorted(agg.items())))
PY
