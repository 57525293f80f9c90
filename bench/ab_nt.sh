#!/bin/bash
# same-box A/B of the non-temporal row stream (DLSA_STREAM_AUX=2 / DLSA_LOGIT_NT=1 variants built by bench/build_variant.sh)
for r in 1 2; do
  python bench/fused_quick.py 1e7 64 100 112
  DLSA_AB_LIB=build/var/libdlsa_nt_irls_pass.so python bench/fused_quick.py 1e7 64 100 112
  python bench/fused_quick.py 2.5e6 500 256
  DLSA_AB_LIB=build/var/libdlsa_nt_logit.so python bench/fused_quick.py 2.5e6 500 256
  for p in 64 100; do
    python bench/gram_time.py 1e7 $p 7 2>&1 | grep GRAM
    python bench/gram_time.py 1e7 $p 7 build/var/libdlsa_nt_gram_narrow.so 2>&1 | grep GRAM
  done
done
