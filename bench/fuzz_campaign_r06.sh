#!/bin/bash
# Round-6 fuzz campaign outside the suite (the suite runs each checker with a few dozen cases): bench/fuzz_campaign_r06.sh > profiles/r06_fuzz.txt
for spec in "bench/fit_fuzz.py 1500 61" "bench/pass_fuzz.py 4000 62" "bench/gram_fuzz.py 2500 63" "bench/lockstep_fuzz.py 400 64" "tests/lars_fuzz.py 300 65" \
            "tests/lars_fuzz.py 200 66 1" "tests/lars_fuzz.py 80 67 2" "bench/onehot_fuzz.py 1500 68" "bench/reduce_fuzz.py 600 69" "bench/eval_fuzz.py 800 70" "bench/linear_fuzz.py 400 71"; do
  set -- $spec
  t0=$(date +%s)
  out=$(timeout 900 python "$@" 2>&1 | grep -E " ok:|Error|assert|Traceback" | tail -3)
  echo "== python $spec  ($(( $(date +%s) - t0 )) s)"; echo "$out"
done
