#!/bin/bash
# same-box A/B of cyclic-kernel builds against the panel kernel: bench/ab_cyc.sh "<p list>" rows reps name...
PS="$1"; ROWS=$2; REPS=$3; shift 3
cp dlsa_amd/libdlsa_hip.so /tmp/libdlsa_orig.so
for rep in 1 2; do
  for p in $PS; do
    echo "== panel: $(DLSA_GRAM_DBG=8 python bench/gram_pitch.py $ROWS $p $p $REPS 2>&1 | grep -E 'PITCH|rror')"
    for v in "$@"; do cp build/var/libdlsa_$v.so dlsa_amd/libdlsa_hip.so; echo "== $v: $(python bench/gram_pitch.py $ROWS $p $p $REPS 2>&1 | grep -E 'PITCH|rror')"; done
  done
done
cp /tmp/libdlsa_orig.so dlsa_amd/libdlsa_hip.so
