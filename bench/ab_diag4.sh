#!/bin/bash
# Same-box A/B of the narrow Gram with diagonal tiles on 4x4x4 MFMAs (the shipped library) against one 16x16x4 per diagonal tile
# (build/var/libdlsa_nodiag4.so = bench/build_variant.sh nodiag4 gram_narrow.hip -DDLSA_NARROW_DIAG4=0 -Wno-inline-asm)
cp dlsa_amd/libdlsa_hip.so /tmp/libdlsa_orig.so
for rep in 1 2; do
for v in orig nodiag4; do
  [ $v = orig ] && cp /tmp/libdlsa_orig.so dlsa_amd/libdlsa_hip.so || cp build/var/libdlsa_$v.so dlsa_amd/libdlsa_hip.so
  for p in 50 64 80 96 100 112 120; do rows=10000000; [ $p -le 64 ] && rows=20000000; echo "== $v: $(python bench/gram_quick.py $rows $p 7 | grep DBG)"; done
done
done
cp /tmp/libdlsa_orig.so dlsa_amd/libdlsa_hip.so
