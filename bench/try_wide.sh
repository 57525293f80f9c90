#!/bin/bash
# time the wide fp32 Gram with alternative builds (build/var/libdlsa_w_<KC>_<STAGES>.so)
cp dlsa_amd/libdlsa_hip.so /tmp/libdlsa_orig.so
for v in "$@"; do
  cp build/var/libdlsa_w_$v.so dlsa_amd/libdlsa_hip.so
  echo "== variant $v"; python bench/gram_quick.py 6000000 2000 3 f32 | grep DBG
done
cp /tmp/libdlsa_orig.so dlsa_amd/libdlsa_hip.so
