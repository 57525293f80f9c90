#!/bin/bash
for f in 0.3 1.0 3 10 100; do echo "freeze=$f"; for cfg in "1e7 100 1" "1e7 100 10" "1.4e7 260 14" "2.5e7 500 1"; do echo -n "  $cfg: "; DLSA_IRLS_FREEZE=$f python bench/irls_trace.py $cfg 2>&1 | grep "^fit" | tail -1; done; done
