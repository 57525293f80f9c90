#!/bin/bash
# A/B builds of the narrow Gram kernel on one box: bench/ab_narrow.sh <suffix> ...  (build/var/libdlsa_narrow<suffix>.so)
cp dlsa_amd/libdlsa_hip.so /tmp/libdlsa_orig.so
for rep in 1 2; do
for v in "$@"; do
  cp build/var/libdlsa_narrow$v.so dlsa_amd/libdlsa_hip.so
  for p in 100 50 112 64; do echo "== $v: $(python bench/gram_quick.py 10000000 $p 7 | grep DBG)"; done
done
done
cp /tmp/libdlsa_orig.so dlsa_amd/libdlsa_hip.so
