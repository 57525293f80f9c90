#!/bin/bash
# plan kernel (dbg 0) against the older kernels (dbg 256) over widths: bench/ab_plan.sh "<p> <p> ..." [rows]
ROWS=${2:-4000000}
for p in $1; do for d in 0 256 0 256; do echo "== p $p dbg $d: $(DLSA_GRAM_DBG=$d python bench/gram_quick.py $ROWS $p 5 | grep DBG)"; done; done
