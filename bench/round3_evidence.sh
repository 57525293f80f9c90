#!/bin/bash
# Round-3 evidence in one call: gpurun_out/r03/*
OUT=gpurun_out/r03; mkdir -p $OUT; export TMPDIR=/tmp
python bench.py --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err
python bench.py --steps 5 --warmup 2 --e2e --no-cpu-baseline > $OUT/bench_e2e.json 2> $OUT/bench_e2e.err
DLSA_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 3 --warmup 1 --rows-per-gpu 4000000 --e2e > $OUT/bench_gloo2_dryrun.json 2> $OUT/bench_gloo2.err
bash bench/profile_round.sh r03 25000000 > $OUT/profile_round.log 2>&1
python bench/bench_configs.py > $OUT/configs.jsonl 2> $OUT/configs.err
python bench/bench_configs.py C5s >> $OUT/configs.jsonl 2>> $OUT/configs.err
bash bench/gram_widths.sh 10000000 > $OUT/gram_widths.txt 2>&1
bash bench/fit_profile.sh C2 > $OUT/fit_C2.txt 2>&1
bash bench/fit_profile.sh C4 > $OUT/fit_C4.txt 2>&1
DRIVER=bench/fused_driver.py bash bench/pmc_quick.sh "1e7 100" fused_p100 irls_pass > $OUT/pmc_fused_p100.txt 2>&1
python bench/fused_quick.py 1e7 50 64 80 100 112 > $OUT/fused_quick.txt 2>&1
python bench/logit_quick.py > $OUT/logit_shapes.txt 2>&1
ls $OUT
