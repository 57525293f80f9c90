"""One lock-step fit with the driver's trace on: the relative step sizes of every pass (pooled / gradient-only / Newton).
   python bench/lockstep_steps.py [K nk p]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dlsa_amd
from dlsa_amd import engine
K, nk, p = (int(float(v)) for v in (sys.argv[1:4] if len(sys.argv) >= 4 else (1000, 20000, 100)))
X, y = engine.synth(20260101, 0, K * nk, p, kind=engine.SYNTH_GAUSSIAN)
mb = dlsa_amd.fit_logistic_partitions(X, y, part_offsets=[k * nk for k in range(K + 1)], batched=True, small=False, trace=True)
print("n_iter", mb.n_iter[:4])
