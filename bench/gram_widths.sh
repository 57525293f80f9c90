#!/bin/bash
# fp64 Gram over widths: the dispatched kernel (DLSA_GRAM_DBG=0) against the round-1 panel kernel everywhere (264 = 8 + 256; + 64 for p <= 120)
ROWS=${1:-10000000}
for p in 50 64 100 112 118 124 130 160 200 230 260 284 290 320 350 380 410 440 470 480 496 500 508 530 560 572; do
  d=264; [ $p -le 120 ] && d=328
  echo "p=$p  new: $(DLSA_GRAM_DBG=0 python bench/gram_quick.py $ROWS $p 5 | grep -o 'median.*')"
  echo "p=$p  panel: $(DLSA_GRAM_DBG=$d python bench/gram_quick.py $ROWS $p 5 | grep -o 'median.*')"
done
