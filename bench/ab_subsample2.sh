#!/bin/bash
for d in 48 64 96 128 200; do for f in 4 6 8; do echo -n "sub=$d fac=$f: "; DLSA_IRLS_SUBSAMPLE=$d DLSA_IRLS_FACTOR_DIV=$f python bench/irls_trace.py 2.5e7 500 1 2>&1 | grep "^fit" | tail -1; done; done
for d in 16 64; do echo -n "K=25 sub=$d: "; DLSA_IRLS_SUBSAMPLE=$d python bench/irls_trace.py 2.5e7 500 25 2>&1 | grep "^fit" | tail -1; echo -n "C4 dense sub=$d: "; DLSA_IRLS_SUBSAMPLE=$d python bench/irls_trace.py 1.4e7 260 14 2>&1 | grep "^fit" | tail -1; done
