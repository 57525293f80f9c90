"""Where the column-split LARS kernel (lars_c.hip) overtakes lars_q.hip: python bench/lars_crossover.py [p ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dlsa_amd import engine
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from lars_c_check import problem, rel_inf, timed
for p in [int(v) for v in sys.argv[1:]] or [300, 400, 500, 640, 768, 900, 1020]:
    S, b, n = problem(p, 0.5, 777 + p)
    St, bt = torch.from_numpy(S).cuda(), torch.from_numpy(b).cuda()
    msq, rq = timed(lambda: engine.lars_path(St, bt, False, float(n), type="lasso"), 5)
    out = "p=%d: default %.2f ms" % (p, msq)
    with engine.kernel_options(lars_q=2):
        msc, rc = timed(lambda: engine.lars_path(St, bt, False, float(n), type="lasso"), 5)
        out += " | column split %.2f ms" % msc
        for w in [int(v) for v in os.environ.get("LARS_C_WGS", "").split()]:
            with engine.kernel_options(lars_q=2, lars_wgs=w):
                msw, rw = timed(lambda: engine.lars_path(St, bt, False, float(n), type="lasso"), 5)
            out += " (%d wgs %.2f)" % (w, msw)
    assert rq["beta"].shape == rc["beta"].shape and rel_inf(rq["beta"].cpu().numpy(), rc["beta"].cpu().numpy()) < 1e-8
    print(out, flush=True)
