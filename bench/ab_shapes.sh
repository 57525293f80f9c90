#!/bin/bash
# A/B builds of libdlsa_hip.so over several Gram shapes on one box: bench/ab_shapes.sh "<rows> <reps>" "<p list>" name...
set -- "$@"; RR="$1"; PS="$2"; shift 2
cp dlsa_amd/libdlsa_hip.so /tmp/libdlsa_orig.so
for rep in 1 2; do
for v in "$@"; do
  cp build/var/libdlsa_$v.so dlsa_amd/libdlsa_hip.so
  for p in $PS; do set -- $RR; echo "== $v: $(python bench/gram_quick.py $1 $p $2 | grep DBG)"; done
done
done
cp /tmp/libdlsa_orig.so dlsa_amd/libdlsa_hip.so
