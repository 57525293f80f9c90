#!/bin/bash
# A/B builds of libdlsa_hip.so on the logit pass: bench/ab_logit.sh old new ...
cp dlsa_amd/libdlsa_hip.so /tmp/libdlsa_orig.so
for v in "$@"; do
  cp build/var/libdlsa_$v.so dlsa_amd/libdlsa_hip.so
  echo "== $v"; python bench/logit_quick.py
done
cp /tmp/libdlsa_orig.so dlsa_amd/libdlsa_hip.so
