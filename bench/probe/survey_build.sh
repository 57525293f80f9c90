#!/bin/bash
# builds probe pairs (segmented / burst DMA issue) for a list of NT: bench/probe/survey_build.sh "10 12 ..."
for nt in $1; do
  while [ $(jobs -r | wc -l) -ge 8 ]; do sleep 1; done
  ( bench/probe/build_probe.sh s$nt $nt > build/probe_s$nt.log 2>&1 || echo "FAIL s$nt" ) &
  while [ $(jobs -r | wc -l) -ge 8 ]; do sleep 1; done
  ( bench/probe/build_probe.sh b$nt $nt -DPLAN_DMA_BURST=1 > build/probe_b$nt.log 2>&1 || echo "FAIL b$nt" ) &
done
wait
