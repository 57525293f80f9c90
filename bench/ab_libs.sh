#!/bin/bash
# A/B two builds of libdlsa_hip.so on the same box: bench/ab_libs.sh "<gram_quick args>" old new ...
ARGS="$1"; shift
cp dlsa_amd/libdlsa_hip.so /tmp/libdlsa_orig.so
for rep in 1 2; do
for v in "$@"; do
  cp build/var/libdlsa_$v.so dlsa_amd/libdlsa_hip.so
  echo "== $v: $(python bench/gram_quick.py $ARGS | grep DBG)"
done
done
cp /tmp/libdlsa_orig.so dlsa_amd/libdlsa_hip.so
