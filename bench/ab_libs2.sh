#!/bin/bash
# same-box A/B of whole-library builds through LD path swap: bench/ab_libs2.sh "<gram_quick args>" name...   (build/var/libdlsa_<name>.so)
ARGS="$1"; shift
cp dlsa_amd/libdlsa_hip.so /tmp/libdlsa_orig.so
for rep in 1 2; do for v in "$@"; do cp build/var/libdlsa_$v.so dlsa_amd/libdlsa_hip.so; echo "== $v: $(python bench/gram_quick.py $ARGS 2>&1 | grep -E 'DBG|rror')"; done; done
cp /tmp/libdlsa_orig.so dlsa_amd/libdlsa_hip.so
