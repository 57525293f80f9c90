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
# 0. the driver's command, unprofiled, in the SAME call (same box, same minute) as the profiled run below: the round's committed
#    default line (profiles/<tag>_bench_default.json) and the reference point of the 3 % guard in tests/test_bench_cpu.py
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err
# 1. per-kernel time of the same command under rocprofv3 (--no-e2e: without the fit legs every gram_cyclic_kernel launch has the
#    benchmark's size, so rocprofv3's own per-kernel average IS the metric kernel's; --no-cpu-baseline: host work only)
rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats --output-format csv -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --rows-per-gpu $ROWS --no-cpu-baseline --no-e2e > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err
# 2. PMC passes (own runs, kernel-trace only)
pmc() { name=$1; shift; rocprofv3 --kernel-trace --pmc "$@" -d $OUT/$name -o $name --output-format csv -- python3 bench.py --steps 2 --warmup 1 --rows-per-gpu $ROWS --no-cpu-baseline --no-e2e > $OUT/$name.log 2>&1; }
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
# the same-call cross-check: rocprofv3's average for the dispatched kernel against the HIP-event kernel time of the profiled run itself
# and of the UNPROFILED default run made just before it
def line(path):
    try:
        return json.loads([l for l in open(path) if l.startswith("{")][-1])
    except Exception as e:
        return None
un, pr = line("$OUT/bench_default.json"), line("$OUT/bench_under_rocprof.json")
if un and pr:
    kname = un["roofline"]["kernel"].split("dlsa::")[1].split(" ")[0]
    csv_avg = [v["avg_ms"] for k, v in out["kernel_ms"].items() if kname.replace(" ", "") in k.replace(" ", "")]
    out["same_call"] = {"kernel": kname, "unprofiled_kernel_ms": un["roofline"]["kernel_ms"], "unprofiled_clock_GHz": un["roofline"]["shader_clock_GHz"],
                        "unprofiled_ms_per_step": un["ms_per_step"], "profiled_kernel_ms_hip_events": pr["roofline"]["kernel_ms"],
                        "profiled_clock_GHz": pr["roofline"]["shader_clock_GHz"], "rocprof_csv_avg_ms": csv_avg[0] if csv_avg else None,
                        "rocprof_over_unprofiled": (csv_avg[0] / un["roofline"]["kernel_ms"]) if csv_avg else None,
                        "note": "one gpurun call: python3 bench.py --gpus 1 --steps 20 --warmup 5, then the same command with "
                                "--no-cpu-baseline --no-e2e under rocprofv3 --kernel-trace --stats"}
    print(json.dumps(out["same_call"], indent=1))
json.dump(out, open("$OUT/pmc_summary.json", "w"), indent=1, sort_keys=True)
print(json.dumps(out["kernel_ms"], indent=1))
for k, d in out["kernels"].items():
    print(k); [print("   %-28s %.6g" % (c, v)) for c, v in sorted(d.items())]
PY
ls $OUT/stats
