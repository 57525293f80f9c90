#!/bin/bash
# same library, kernels selected by the valid-result DLSA_GRAM_DBG switches: bench/gram_ab_env.sh "<rows> <p> <reps>" dbg...
ARGS="$1"; shift
for rep in 1 2; do for d in "$@"; do echo "== dbg $d: $(DLSA_GRAM_DBG=$d python bench/gram_quick.py $ARGS | grep DBG)"; done; done
