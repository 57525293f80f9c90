#!/bin/bash
# Round-5 width table: the fp64 Gram (dispatched kernel) and the logit pass at 26 widths, 1e7 rows (the narrow ones 2e7) -- one box.
for p in 50 64 100 112 118 124 130 160 200 230 260 284 290 320 350 380 410 440 470 480 496 500 508 530 560 572; do
  rows=10000000; [ $p -le 64 ] && rows=20000000
  g=$(python bench/gram_quick.py $rows $p 5 | grep -o 'median.*')
  l=$(python bench/logit_quick.py $rows $p 2>/dev/null | tail -1)
  echo "p=$p rows=$rows  gram: $g  | logit: $l"
done
