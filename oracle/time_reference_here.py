"""TEST INFRASTRUCTURE ONLY (build container only) -- times the REFERENCE's own functions beside the oracle's restatement of them,
on the same seeded rows, so that bench.py's `cpu_baseline` (kind "port": the oracle, oracle/cpu_baseline.py) says what it is a
baseline OF (last row; VERDICT r4 "missing 2").

  logistic_model as shipped   dlsa/models.py:42-147 -- sklearn newton-cg, tol = 1e-4 (models.py:110-113), pandas in / pandas out
  logistic_model, exact MLE   the same call under the harness's tol = 1e-15 shim (what the parity fixtures are generated with)
  oracle.logistic_model_block oracle/dlsa_oracle.py -- dense Newton to |step| <= 1e-13 + the Hessian at the MLE (what bench.py times)
  lars_lsa                    dlsa/lsa.py:90-212 against oracle.lars_lsa, p = 500

each with one BLAS thread (the reference's one-core Spark executors, projects/bash/run_spark_dlsa.sh:15,42) and with all threads.
Imports /root/reference through oracle/ref_shims.py (sys.dont_write_bytecode: nothing is written into the reference tree); refuses
to run where the reference is absent.  Output: profiles/r05_reference_vs_port.json + a table on stdout.

    python3 oracle/time_reference_here.py [--quick]
"""
import json
import os
import sys
import time
import warnings

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

import numpy as np                                  # noqa: E402
import pandas as pd                                 # noqa: E402
from threadpoolctl import threadpool_limits         # noqa: E402

from oracle import dlsa_oracle as orc, fast_synth   # noqa: E402
from oracle.ref_shims import load_reference, reference_available   # noqa: E402


def best_of(fn, reps):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        out = fn()
        ts.append(time.perf_counter() - t0)
    return min(ts), out


def main():
    if not reference_available():
        print("time_reference_here: /root/reference is not present (this script runs in the build container only)", file=sys.stderr)
        return 2
    quick = "--quick" in sys.argv
    shapes = [(20000, 500), (200000, 100)] + ([] if quick else [(200000, 500)])
    ref_shipped, ref_mle = load_reference(tol=None), load_reference(tol=1e-15)
    cores = int(os.cpu_count() or 1)
    rows = []
    for n, p in shapes:
        X, y = fast_synth.synth_logistic(20260101, 0, n, p, orc.SYNTH_GAUSSIAN)
        names = ["x%d" % i for i in range(p)]
        df = pd.DataFrame(np.column_stack([np.zeros(n), y, X]), columns=["partition_id", "label"] + names)
        for threads in (1, cores):
            with threadpool_limits(limits=threads), warnings.catch_warnings():
                warnings.simplefilter("ignore")
                reps = 2 if n * p <= 2e7 else 1
                t_ship, out_ship = best_of(lambda: ref_shipped.models.logistic_model(df, "label", fit_intercept=False), reps)
                t_mle, out_mle = best_of(lambda: ref_mle.models.logistic_model(df, "label", fit_intercept=False), reps)
                t_port, out_port = best_of(lambda: orc.logistic_model_block(X, y), reps)
            coef_mle = out_mle["coef"].to_numpy(dtype=float)
            rec = {"rows": n, "p": p, "blas_threads": threads,
                   "reference_shipped_s": t_ship, "reference_exact_mle_s": t_mle, "oracle_port_s": t_port,
                   "reference_shipped_rows_per_s": n / t_ship, "reference_exact_mle_rows_per_s": n / t_mle, "oracle_port_rows_per_s": n / t_port,
                   "port_over_shipped": t_port / t_ship, "port_over_exact_mle": t_port / t_mle,
                   "coef_port_vs_reference_exact_mle_rel_linf": float(np.max(np.abs(out_port[0] - coef_mle)) / np.max(np.abs(coef_mle))),
                   "coef_shipped_vs_exact_mle_rel_linf": float(np.max(np.abs(out_ship["coef"].to_numpy(dtype=float) - coef_mle)) / np.max(np.abs(coef_mle)))}
            rows.append(rec)
            print("n=%-7d p=%-4d threads=%-2d  reference as shipped %8.2f s (%9.0f rows/s)   reference exact MLE %8.2f s   oracle port %8.2f s (%9.0f rows/s)"
                  "   port / shipped = %.2f   port / exact = %.2f" % (n, p, threads, t_ship, n / t_ship, t_mle, t_port, n / t_port, t_port / t_ship, t_port / t_mle), flush=True)
        del df, X, y
    # LARS at p = 500 on a Hessian of the model (the reduce stage's serial part)
    p, n = (200, 20000) if quick else (500, 60000)
    X, y = fast_synth.synth_logistic(7, 0, n, p, orc.SYNTH_GAUSSIAN)
    coef, smc, S = orc.logistic_model_block(X, y)
    lars = []
    for threads in (1, cores):
        with threadpool_limits(limits=threads), warnings.catch_warnings():
            warnings.simplefilter("ignore")
            t_ref, r_ref = best_of(lambda: ref_shipped.lsa.lars_lsa(np.matrix(S), coef, False, n), 1)
            t_port, r_port = best_of(lambda: orc.lars_lsa(S, coef, False, n), 1)
        err = float(np.max(np.abs(np.asarray(r_ref["beta"]) - np.asarray(r_port["beta"]))))
        lars.append({"p": p, "blas_threads": threads, "reference_s": t_ref, "oracle_port_s": t_port, "port_over_reference": t_port / t_ref,
                     "beta_path_abs_diff": err})
        print("lars_lsa p=%d threads=%-2d  reference %.2f s   oracle port %.2f s   port / reference = %.2f   |path diff| %.1e" % (p, threads, t_ref, t_port, t_port / t_ref, err), flush=True)
    one = [r for r in rows if r["blas_threads"] == 1]
    doc = {"host_cores": cores, "map": rows, "lars": lars,
           "port_over_shipped_one_thread": {"%dx%d" % (r["rows"], r["p"]): round(r["port_over_shipped"], 3) for r in one},
           "port_over_exact_mle_one_thread": {"%dx%d" % (r["rows"], r["p"]): round(r["port_over_exact_mle"], 3) for r in one},
           "note": "seconds are best-of; 'reference as shipped' stops at sklearn's tol = 1e-4 (models.py:110-113), the port and the "
                   "tol = 1e-15 shim run to the exact MLE; the reference's time includes its pandas frame handling (models.py:50-147)"}
    out = os.path.join(ROOT, "profiles", "r05_reference_vs_port.json")
    with open(out, "w") as f:
        json.dump(doc, f, indent=1, sort_keys=True)
    print("wrote", out)
    return 0


if __name__ == "__main__":
    sys.exit(main())
