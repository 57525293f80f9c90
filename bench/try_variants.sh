#!/bin/bash
# time the Gram kernel with alternative builds of libdlsa_hip.so (build/var/libdlsa_<KC>_<OCC>.so)
cp dlsa_amd/libdlsa_hip.so /tmp/libdlsa_orig.so
for v in "$@"; do
  cp build/var/libdlsa_$v.so dlsa_amd/libdlsa_hip.so
  echo "== variant $v"; python bench/gram_quick.py 25000000 500 5 | grep DBG
done
cp /tmp/libdlsa_orig.so dlsa_amd/libdlsa_hip.so
