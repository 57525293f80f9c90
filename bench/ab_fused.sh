#!/bin/bash
# same-box A/B of the fused Newton pass / narrow Gram / ring logit pass: the shipped library against build/var/libdlsa_<name>.so
#   bench/ab_fused.sh rows "<p list>" name...
ROWS=$1; PS="$2"; shift 2
for round in 1 2; do
  python bench/fused_quick.py $ROWS $PS 2>&1 | grep -E "^p=|rror"
  for v in "$@"; do DLSA_AB_LIB=build/var/libdlsa_$v.so python bench/fused_quick.py $ROWS $PS 2>&1 | grep -E "^p=|rror"; done
done
