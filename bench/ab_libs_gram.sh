#!/bin/bash
# Same-box A/B of Gram timings for variant libraries: bench/ab_libs_gram.sh "<p list>" <variant> ...   (orig = the shipped library)
PS="$1"; shift
cp dlsa_amd/libdlsa_hip.so /tmp/libdlsa_orig.so
for rep in 1 2; do
for v in orig "$@"; do
  [ $v = orig ] && cp /tmp/libdlsa_orig.so dlsa_amd/libdlsa_hip.so || cp build/var/libdlsa_$v.so dlsa_amd/libdlsa_hip.so
  for p in $PS; do rows=10000000; [ $p -le 64 ] && rows=20000000; echo "== $v: $(python bench/gram_quick.py $rows $p 7 2>/dev/null | grep DBG)"; done
done
done
cp /tmp/libdlsa_orig.so dlsa_amd/libdlsa_hip.so
