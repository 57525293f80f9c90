#!/bin/bash
# same-box A/B of the partition chains (csrc/irls.hip): chain count x seeding, on the configurations with K > 1
for cfg in "1 1" "2 0" "2 1" "4 0" "4 1"; do set -- $cfg; echo "chains=$1 seed=$2"
  export DLSA_IRLS_CHAINS=$1 DLSA_IRLS_SEED=$2
  python bench/oh_trace.py 2>&1 | tail -1
  python bench/irls_trace.py 1e7 100 10 2>&1 | grep "^fit" | tail -1
  python bench/irls_trace.py 1.4e7 260 14 2>&1 | grep "^fit" | tail -1
  python bench/irls_trace.py 2.5e7 500 25 2>&1 | grep "^fit" | tail -1
done
