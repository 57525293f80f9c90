#!/bin/bash
# Round-6 evidence in ONE gpurun call (same box): bench/evidence_r06.sh [tag]
#   default line + rocprofv3 stats of the same command + PMC passes (profile_round.sh), stale kernel evidence, every configuration's
#   shard, the eight-rank gloo dry run with the one-rank line at the same rows per GPU.
TAG=${1:-r06}
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
bench/profile_round.sh $TAG > $OUT/profile_round.log 2>&1
python3 bench/pmc_evidence.py $TAG narrow_p100 > $OUT/pmc_evidence.log 2>&1
cp profiles/${TAG}_pmc_narrow_p100.json $OUT/ 2>/dev/null
python3 bench/bench_configs.py > $OUT/configs.jsonl 2> $OUT/configs.err
DLSA_BENCH_BACKEND=gloo python3 bench.py --gpus 8 --rows-per-gpu 2000000 > $OUT/bench_gloo8_dryrun.json 2> $OUT/bench_gloo8_dryrun.err
python3 bench.py --gpus 1 --rows-per-gpu 2000000 --no-cpu-baseline > $OUT/bench_rows2e6_n1.json 2> $OUT/bench_rows2e6_n1.err
tail -3 $OUT/profile_round.log; tail -2 $OUT/pmc_evidence.log; wc -l $OUT/configs.jsonl; tail -c 300 $OUT/bench_gloo8_dryrun.err
