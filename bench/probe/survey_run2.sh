#!/bin/bash
# bench/probe/survey_run2.sh "<nt list>" tagA tagB : pairs of probe builds per NT, ~10 ms runs
for nt in $1; do p=$((16 * nt)); rows=$((2500000000 / (p * p / 250 + 1) / 1000 * 1000)); [ $rows -gt 30000000 ] && rows=30000000
  for rep in 1 2; do for v in $2 $3; do timeout 60 build/probe/probe_$v$nt $rows $p 5; done; done; done
