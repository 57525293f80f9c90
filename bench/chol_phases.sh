#!/bin/bash
# GPU time of chol_small_kernel with phases left out (build/var/libdlsa_cs<mask>.so: bench/build_variant.sh cs<mask> chol.hip -DCS_SKIP=<mask>)
export TMPDIR=/tmp; P=${1:-100}
for v in "" 1 2 4 8 16 31; do
  lib=dlsa_amd/libdlsa_hip.so; [ -n "$v" ] && lib=build/var/libdlsa_cs$v.so
  O=gpurun_out/cs_$v; rm -rf $O; mkdir -p $O
  rocprofv3 --kernel-trace --stats -d $O -o t --output-format csv -- python3 bench/chol_quick.py $P $lib > $O/out.txt 2>&1
  echo "skip=${v:-0}: $(grep chol_small $O/t_kernel_stats.csv | awk -F, '{printf "calls %s avg %.1f us", $2, $4/1000}')"
done
