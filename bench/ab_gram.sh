#!/bin/bash
# same-box A/B of library builds (build/var/libdlsa_<name>.so from bench/build_variant.sh) against the shipped one:
#   bench/ab_gram.sh "<p list>" rows reps name...      two rounds, the shipped library first in each
PS="$1"; ROWS=$2; REPS=$3; shift 3
for round in 1 2; do
  for p in $PS; do
    python bench/gram_time.py $ROWS $p $REPS 2>&1 | grep -E "GRAM|rror"
    for v in "$@"; do python bench/gram_time.py $ROWS $p $REPS build/var/libdlsa_$v.so 2>&1 | grep -E "GRAM|rror"; done
  done
done
