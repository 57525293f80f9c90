#!/bin/bash
# Collect the round's profiling evidence on the GPU box into gpurun_out/<tag>/ :
#   kernel-trace --stats of the default bench command, PMC passes (separate runs), micro-benchmarks.
# usage: bench/profile_round.sh <tag> [rows-per-gpu]
TAG=${1:-r01}
ROWS=${2:-25000000}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
./bench/ubench_f64 > $OUT/ubench_f64.txt 2>&1
./bench/ubench_f64_4x4 > $OUT/ubench_f64_4x4.txt 2>&1
./bench/ubench_gram_inner > $OUT/ubench_gram_inner.txt 2>&1
./bench/probe_mfma4x4 > $OUT/probe_mfma4x4.txt 2>&1
# 1. per-kernel time of the default bench command
rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats --output-format csv -- python3 bench.py --rows-per-gpu $ROWS --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err
# 2. PMC passes (own runs, kernel-trace only)
pmc() { name=$1; shift; rocprofv3 --kernel-trace --pmc "$@" -d $OUT/$name -o $name --output-format csv -- python3 bench.py --steps 2 --warmup 1 --rows-per-gpu $ROWS --no-cpu-baseline > $OUT/$name.log 2>&1; }
pmc pmc_sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES
pmc pmc_sq2 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD
pmc pmc_grbm GRBM_GUI_ACTIVE
pmc pmc_tcc1 TCC_HIT_sum TCC_MISS_sum
pmc pmc_fetch FETCH_SIZE
pmc pmc_write WRITE_SIZE
python3 - <<PY
import csv, collections, glob, json, os
import hashlib
h = hashlib.sha256()
for fsrc in ("gram.hip", "gram_cyclic.hip", "gram_cyclic_asm.inc", "common.h"):
    h.update(open("dlsa_amd/csrc/" + fsrc, "rb").read())
out = {"rows_per_gpu": $ROWS, "p": 500, "gram_hip_sha16": h.hexdigest()[:16], "kernels": {}}
for f in sorted(glob.glob("$OUT/pmc_*/*_counter_collection.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0]
        agg[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[(k, r['Counter_Name'])] += 1
    for k, d in agg.items():
        if any(s in k for s in ('gram_kernel', 'gram_cyclic_kernel', 'gram_plan_kernel', 'logit_kernel', 'gram_reduce')):
            for c, v in d.items():
                out["kernels"].setdefault(k, {})[c] = v / cnt[(k, c)]
# kernel durations from the stats run
dur = collections.defaultdict(list)
for f in glob.glob("$OUT/stats/*_kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        dur[r['Kernel_Name'].split('(')[0]].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6)
out["kernel_ms"] = {k: {"calls": len(v), "avg_ms": sum(v) / len(v), "min_ms": min(v), "max_ms": max(v)} for k, v in dur.items()}
json.dump(out, open("$OUT/pmc_summary.json", "w"), indent=1, sort_keys=True)
print(json.dumps(out["kernel_ms"], indent=1))
for k, d in out["kernels"].items():
    print(k); [print("   %-28s %.6g" % (c, v)) for c, v in sorted(d.items())]
PY
ls $OUT/stats
