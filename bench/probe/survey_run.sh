#!/bin/bash
# runs the probe pairs of survey_build.sh: rows scaled so that every run takes ~10 ms
for nt in $1; do p=$((16 * nt)); rows=$((1000000000 / (p * p / 250 + 1) / 1000 * 1000)); [ $rows -gt 20000000 ] && rows=20000000
  for rep in 1 2; do for v in b s; do timeout 60 build/probe/probe_$v$nt $rows $p 5; done; done; done
