#!/bin/bash
# GPU time of the small-system kernel (chol.hip) with phases left out -- build/var/libdlsa_cs<mask>.so from
#   bench/build_variant.sh cs<mask> chol.hip -DCS_SKIP=<mask>
# (the LDS-Cholesky version this round started with had five phases, masks 1 .. 16: factor loop 88 us, inverse 59, L / Linv stores 7,
#  H^-1 46, solve 3 of 220 us at p = 100; the sweep kernel that replaced it has one, mask 1 = no sweeps)
export TMPDIR=/tmp; P=${1:-100}
for v in "" 1; do
  lib=dlsa_amd/libdlsa_hip.so; [ -n "$v" ] && lib=build/var/libdlsa_cs$v.so
  O=gpurun_out/cs_$v; rm -rf $O; mkdir -p $O
  rocprofv3 --kernel-trace --stats -d $O -o t --output-format csv -- python3 bench/chol_quick.py $P $lib > $O/out.txt 2>&1
  echo "skip=${v:-0}: $(grep chol_small $O/t_kernel_stats.csv | awk -F, '{printf "calls %s avg %.1f us", $2, $4/1000}')"
done
