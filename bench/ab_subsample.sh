#!/bin/bash
# cold-start subsample divisor (DLSA_IRLS_SUBSAMPLE) on the single-partition configurations
for d in 4 8 16 32 64; do echo "divisor=$d"; DLSA_IRLS_SUBSAMPLE=$d python bench/irls_trace.py 1e7 100 1 2>&1 | grep "^fit" | tail -1; DLSA_IRLS_SUBSAMPLE=$d python bench/irls_trace.py 2.5e7 500 1 2>&1 | grep "^fit" | tail -1; done
