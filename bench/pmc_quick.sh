#!/bin/bash
# PMC passes of gram_quick at one shape with the CURRENT libdlsa_hip.so (separate rocprofv3 runs, kernel-trace only):
#   bench/pmc_quick.sh "<gram_quick args>" <tag> [kernel-name pattern]
ARGS="$1"; TAG=${2:-x}; PAT=${3:-gram}
OUT=gpurun_out/pmc_$TAG; mkdir -p $OUT; export TMPDIR=/tmp
run() { name=$1; shift; rocprofv3 --kernel-trace --pmc "$@" -d $OUT/$name -o $name --output-format csv -- python3 ${DRIVER:-bench/gram_quick.py} $ARGS > $OUT/$name.log 2>&1; }
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES
run sq2 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD
run grbm GRBM_GUI_ACTIVE
run tcc TCC_HIT_sum TCC_MISS_sum
run fetch FETCH_SIZE
run write WRITE_SIZE
python3 - <<PY
import csv, collections, glob, json
res = {}
for f in sorted(glob.glob("$OUT/*/*_counter_collection.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0]; agg[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[(k, r['Counter_Name'])] += 1
    for k, d in agg.items():
        if '$PAT' in k and 'reduce' not in k:
            for c, v in sorted(d.items()): res.setdefault(k, {})[c] = v / cnt[(k, c)]
dur = collections.defaultdict(list)
for f in glob.glob("$OUT/sq1/*_kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0]
        if '$PAT' in k and 'reduce' not in k: dur[k].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6)
for k, d in res.items():
    if k in dur: d['duration_ms_under_pmc'] = sorted(dur[k])[len(dur[k]) // 2]
    print('$TAG', k)
    for c, v in sorted(d.items()): print('    %-28s %.6g' % (c, v))
json.dump({"args": "$ARGS", "kernels": res}, open("$OUT/summary.json", "w"), indent=1, sort_keys=True)
PY
