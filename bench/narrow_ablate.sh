#!/bin/bash
# What the narrow Gram's launch time is made of (knobs build, wrong results by design): DLSA_GRAM_DBG=1 no DMA after the prologue,
# 128 no MFMAs, 129 neither.   needs `make knobs` (bench/libdlsa_hip_knobs.so)
cp dlsa_amd/libdlsa_hip.so /tmp/libdlsa_orig.so
cp bench/libdlsa_hip_knobs.so dlsa_amd/libdlsa_hip.so
for rep in 1 2; do
for p in 50 64 100 112; do
  rows=10000000; [ $p -le 64 ] && rows=20000000
  for d in 0 1 128 129; do echo "== $(DLSA_GRAM_DBG=$d python bench/gram_quick.py $rows $p 7 2>/dev/null | grep DBG)"; done
done
done
cp /tmp/libdlsa_orig.so dlsa_amd/libdlsa_hip.so
