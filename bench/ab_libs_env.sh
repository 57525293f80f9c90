#!/bin/bash
# like ab_libs.sh but with DLSA_GRAM_DBG passed through: bench/ab_libs_env.sh DBG "<args>" libs...
D="$1"; ARGS="$2"; shift; shift
cp dlsa_amd/libdlsa_hip.so /tmp/libdlsa_orig.so
for v in "$@"; do
  cp build/var/libdlsa_$v.so dlsa_amd/libdlsa_hip.so
  echo "== $v: $(DLSA_GRAM_DBG=$D python bench/gram_quick.py $ARGS | grep DBG)"
done
cp /tmp/libdlsa_orig.so dlsa_amd/libdlsa_hip.so
