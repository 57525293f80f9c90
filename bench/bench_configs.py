"""Per-GPU-shard timings of every BASELINE.json configuration on ONE MI355X (evidence table for
DESIGN.md; the headline metric lives in bench.py).  Synthetic inputs as SURVEY.md section 8(d)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np   # noqa: E402
import torch         # noqa: E402

import dlsa_amd      # noqa: E402
from dlsa_amd import engine  # noqa: E402


FP64_MFMA_PEAK_TF, FP32_MFMA_PEAK_TF, HBM_PEAK_GBS = 78.6, 157.3, 8000.0      # as bench.py / MI355X_MICROARCH.md


def traffic_from_evidence(tag, p, n):
    """HBM bytes of one launch over n rows from the committed counter passes of that kernel (profiles/r0N_pmc_<tag>.json,
    bench/pmc_evidence.py): per row of the evidence run, scaled; None when the evidence is of another width or of other sources."""
    try:
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        import pmc_evidence as ev
        doc = json.load(open(ev.latest_evidence_path(tag)))
        if doc["p"] != p or doc["sources_sha16"] != ev.sources_sha16(ev.KERNELS[tag]["src"]):
            return None
        k = next(iter(doc["kernels"].values()))
        return k["hbm_bytes"] / doc["rows"] * n
    except Exception:
        return None


def roofline(kind, n, p, seconds, tag=None):
    """{"bound", "achieved", "peak", "unit", "frac", "traffic"} of one launch: kind "gram64" / "gram32" (algorithmic flops of the
    upper triangle), "fused" (the Gram's flops: the pass's logistic terms are not counted as useful work), "logit" (bytes)."""
    if kind in ("gram64", "fused"):
        ach, peak, unit, bound = n * (p * (p + 1) + p) / seconds / 1e12, FP64_MFMA_PEAK_TF, "TFLOP/s", "mfma"
    elif kind == "gram32":
        ach, peak, unit, bound = n * p * (p + 1) / seconds / 1e12, FP32_MFMA_PEAK_TF, "TFLOP/s", "mfma"
    else:
        ach, peak, unit, bound = n * 8 * (p + 2) / seconds / 1e9, HBM_PEAK_GBS, "GB/s", "hbm"
    return {"bound": bound, "achieved": ach, "peak": peak, "unit": unit, "frac": ach / peak, "kernel": engine.gram_last_kernel()[0] if kind != "logit" else None,
            "traffic": traffic_from_evidence(tag, p, n) if tag else None, "ms": seconds * 1e3}


def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t = time.perf_counter(); out = fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
    return sorted(ts)[len(ts) // 2], out


def logistic_config(name, n, p, K, kind=engine.SYNTH_GAUSSIAN, Xy=None):
    extra = {}
    if Xy is not None:
        X, y, extra = Xy
    else:
        X, y = engine.synth(20260101, 0, n, p, kind=kind)
    p = X.shape[1]
    offs = [int(n * k / K) for k in range(K + 1)]
    beta = torch.zeros(p, dtype=torch.float64, device="cuda"); beta[: int(0.4 * p)] = 1.0
    w, _, _ = engine.logit_pass(X, y, beta)
    t_gram, H = timed(lambda: engine.gram(X, w))
    kname = engine.gram_last_kernel()[0]
    tag = "cyclic_p500" if "cyclic" in kname else "plan_p260" if "plan" in kname else "narrow_p100" if "narrow" in kname else None
    rl = {"gram": roofline("gram64", n, p, t_gram, tag)}
    t_logit, _ = timed(lambda: engine.logit_pass(X, y, beta))
    rl["logit"] = roofline("logit", n, p, t_logit, "logit_p500" if p == 500 else "logit_p50" if p == 50 else None)
    if 49 <= p <= 120 and p % 2 == 0 and n >= 8192:
        t_fused, _ = timed(lambda: engine.irls_pass(X, y, beta))
        rl["fused"] = roofline("fused", n, p, t_fused, "fused_p100")
        if p == 100:
            # A second fraction: fp64 VALU and fp64 MFMA share ONE pipe on gfx950 (bench/ubench_gap.hip: every VALU
            # instruction between MFMAs costs ~4.7 pipe cycles), so the fused pass's ceiling is not "the Gram's MFMAs with the logistic
            # terms hidden" but MFMA cycles + 4.7 x VALU per trip: 112 MFMAs (5866 cycles) + 177 VALU = 6698 cycles per 64 rows and CU
            # (irls_pass.hip's p = 100 shape), at 2.4 GHz and at the ~2.1 GHz the chip holds under this load
            cyc = (n / 256.0 / 64.0) * (5866.0 + 4.7 * 177.0)
            rl["fused"]["shared_pipe"] = {"ceiling_ms_at_2.4GHz": cyc / 2.4e6, "frac_at_2.4GHz": cyc / 2.4e6 / (t_fused * 1e3),
                                          "ceiling_ms_at_2.1GHz": cyc / 2.1e6, "frac_at_2.1GHz": cyc / 2.1e6 / (t_fused * 1e3),
                                          "what": "time of the pass's own MFMA + 4.7 x VALU pipe cycles / measured time"}
    def whole():
        mb = dlsa_amd.fit_logistic_partitions(X, y, part_offsets=offs)
        out = dlsa_amd.dlsa_mapred(mb)
        sel = dlsa_amd.dlsa(out.iloc[:, 2:], out["beta_byOLS"], n)
        return mb, out, sel
    t_all, (mb, out, sel) = timed(whole, reps=2)
    t_fit, _ = timed(lambda: engine.irls_fit(X, y, offs), reps=2)
    fl = p * (p + 1) + p
    return {**extra, "config": name, "n": n, "p": p, "K": K, "dtype": "f64",
            "gram_ms": t_gram * 1e3, "gram_rows_per_s": n / t_gram, "gram_TF_alg": n * fl / t_gram / 1e12,
            "gram_GBps_alg": n * 8 * (p + 1) / t_gram / 1e9,
            "logit_ms": t_logit * 1e3, "logit_GBps": n * 8 * (p + 2) / t_logit / 1e9,
            "map_fit_s": t_fit, "map_reduce_lars_s": t_all, "irls_iters": mb.n_iter[:3], "status_ok": all(s == 0 for s in mb.status),
            # the launch a fresh Hessian costs in this configuration's fit, and the other passes' own fractions
            "roofline": rl.get("fused", rl["gram"]), "rooflines": rl}


def airline_shaped(n, seed=7):
    """Config 4 surrogate (bench/surrogates.py): timings of the design kernel and of the structured passes."""
    from surrogates import airline_shaped as make
    c = make(n, seed, dense=False)
    num, codes, spec, plan, y, beta, p = c["num"], c["codes"], c["spec"], c["plan"], c["y"], c["beta"], c["p"]
    Xbuf = torch.empty((n, p), dtype=torch.float64, device="cuda")
    t_design, (X, seen) = timed(lambda: engine.design(num, codes, *spec, out=Xbuf))
    assert int(seen.sum()) == p
    info = {"design_ms": t_design * 1e3, "design_write_GBps": n * p * 8 / t_design / 1e9,
            "design_input_bytes_per_row": 7 * 8 + 5 * 4, "design_output_bytes_per_row": p * 8}
    w, _, _ = engine.logit_pass(X, y, beta)
    t_og, Hs = timed(lambda: engine.onehot_gram(plan, num, codes, w))
    t_ol, _ = timed(lambda: engine.onehot_logit_pass(plan, num, codes, y, beta))
    Hd = engine.gram(X, w)
    K = 14
    offs = [int(n * k / K) for k in range(K + 1)]
    t_of, rs = timed(lambda: engine.onehot_irls_fit(plan, num, codes, y, offs), reps=2)
    rd = engine.irls_fit(X, y, offs)
    info.update({"structured_roles": plan.roles, "structured_gram_ms": t_og * 1e3, "structured_logit_ms": t_ol * 1e3,
                 "structured_raw_GBps_gram": n * (76 + 8) * plan.roles / t_og / 1e9, "structured_map_fit_s": t_of,
                 "structured_vs_dense_H_relerr": float((Hs - Hd).abs().max() / Hd.abs().max()),
                 "structured_vs_dense_coef_relerr": float((rs["coef"] - rd["coef"]).abs().max() / rd["coef"].abs().max()),
                 "structured_status_ok": all(v == 0 for v in rs["status"])})
    return X, y, info


def linear_config(name, n, p, K):
    X, _ = engine.synth(20260101, 0, n, p, kind=engine.SYNTH_GAUSSIAN, labels=False, dtype=torch.float32)
    beta = torch.zeros(p, dtype=torch.float32, device="cuda"); beta[: int(0.4 * p)] = 1.0
    y = torch.randn(n, dtype=torch.float32, device="cuda")
    for r in range(0, n, 1_000_000):                 # chunked: one giant sgemv (4.8e10 elements) returned NaNs under memory pressure
        y[r:r + 1_000_000] += X[r:r + 1_000_000] @ beta
    if os.environ.get("DLSA_DIAG"):
        print("y finite", bool(torch.isfinite(y).all()), file=sys.stderr, flush=True)
        offs = [int(n * k / K) for k in range(K + 1)]
        for k in range(K):
            lo, hi = offs[k], offs[k + 1]
            H = engine.gram(X[lo:hi], None); g, vv = engine.xtv(X[lo:hi], y[lo:hi])
            Hd, gd = H.double(), g.double()
            msg = "part %d H finite %s g finite %s Hmax %.3e" % (k, bool(torch.isfinite(H).all()), bool(torch.isfinite(g).all()), float(H.abs().max()))
            try:
                engine.spd_solve(Hd, gd); msg += " solve ok"
            except Exception as e:
                msg += " SOLVE FAIL %s" % e
                try:
                    ev = torch.linalg.eigvalsh(Hd); msg += " eig [%.3e, %.3e]" % (float(ev.min()), float(ev.max()))
                except Exception as e2:
                    msg += " eig fail"
                bad = (~torch.isfinite(H)).nonzero()
                msg += " nonfinite count %d first %s" % (bad.shape[0], bad[:3].tolist())
                asym = float((H - H.T).abs().max()); msg += " asym %.3e" % asym
            print(msg, file=sys.stderr, flush=True)
    t_gram, _ = timed(lambda: engine.gram(X, None))
    rl = roofline("gram32", n, p, t_gram, "wide_f32_p2000")
    t_xty, _ = timed(lambda: engine.xtv(X, y))
    offs = [int(n * k / K) for k in range(K + 1)]
    t_all, mb = timed(lambda: dlsa_amd.fit_linear_partitions(X, y, part_offsets=offs), reps=2)
    out = dlsa_amd.dlsa_mapred(mb)

    def whole():
        o = dlsa_amd.dlsa_mapred(dlsa_amd.fit_linear_partitions(X, y, part_offsets=offs))
        return dlsa_amd.dlsa(o.iloc[:, 2:], o["beta_byOLS"], n)
    t_whole, _ = timed(whole, reps=2)
    fl = p * (p + 1)
    return {"config": name, "n": n, "p": p, "K": K, "dtype": "f32",
            "gram_ms": t_gram * 1e3, "gram_rows_per_s": n / t_gram, "gram_TF_alg": n * fl / t_gram / 1e12,
            "gram_GBps_alg": n * 4 * p / t_gram / 1e9, "xty_ms": t_xty * 1e3, "xty_GBps": n * 4 * (p + 1) / t_xty / 1e9,
            "map_fit_s": t_all, "map_reduce_lars_s": t_whole, "theta_err_linf": float((torch.from_numpy(out["beta_byOLS"].to_numpy()).cuda() - beta.double()).abs().max()),
            "roofline": rl}


def linear_streaming_config(name, n, p, K, chunk_rows):
    """Config 5 at its STATED per-GPU size: 6.25e7 x 2000 fp32 = 500 GB of rows streamed through one chunk buffer
    (generated on the device chunk by chunk); per-phase times from a dry pass over the same chunks."""
    dt = torch.float32
    rows_max = min(chunk_rows, n // K)
    Xc = engine.empty_rows(rows_max, p, dt, "cuda"); yc = torch.empty(rows_max, dtype=dt, device="cuda")
    H = torch.zeros((p, p), dtype=torch.float64, device="cuda")
    kind = os.environ.get("DLSA_C5_KIND", "gaussian32")       # "gaussian": the fp64-defined stream rounded to fp32 + a separate response pass (round 3)
    if kind == "gaussian32":
        t_syn, _ = timed(lambda: engine.synth_linear32(20260101, 0, rows_max, p, out=Xc, out_y=yc))
    else:
        t_syn, _ = timed(lambda: (engine.synth(20260101, 0, rows_max, p, kind=engine.SYNTH_GAUSSIAN, labels=False, dtype=dt, out=Xc),
                                  engine.synth_response(20260101, 0, Xc, out=yc)))
    t_gram, _ = timed(lambda: engine.gram_acc64(Xc, None, out=H, accumulate=True))
    kern = engine.gram_last_kernel()[0]
    rl = roofline("gram32", rows_max, p, t_gram, "wide_f32_p2000")
    t_stats, _ = timed(lambda: engine.xtv_stats(Xc, yc, want_colsum=True))
    del Xc, yc, H
    torch.cuda.reset_peak_memory_stats(); base = torch.cuda.memory_allocated()
    t0 = time.perf_counter()
    overlap = os.environ.get("DLSA_C5_OVERLAP", "0") != "0"
    mb = dlsa_amd.fit_linear_streaming(n, p, partition_num=K, chunk_rows=chunk_rows, fit_intercept=True, dtype=dt, overlap=overlap, kind=kind)
    torch.cuda.synchronize(); t_map = time.perf_counter() - t0
    peak = torch.cuda.max_memory_allocated() - base
    t1 = time.perf_counter()
    out = dlsa_amd.dlsa_mapred(mb)
    sel = dlsa_amd.dlsa(out.iloc[:, 2:], out["beta_byOLS"], n, fit_intercept=True)
    torch.cuda.synchronize(); t_rest = time.perf_counter() - t1
    truth = np.concatenate([[0.0], np.ones(int(0.4 * p)), np.zeros(p - int(0.4 * p))])
    fl = p * (p + 1)
    return {"config": name, "n": n, "p": p, "K": K, "dtype": "f32", "chunk_rows": rows_max, "gram_kernel": kern,
            "chunk_synth_ms": t_syn * 1e3, "chunk_gram_ms": t_gram * 1e3, "chunk_stats_ms": t_stats * 1e3,
            "gram_rows_per_s": rows_max / t_gram, "gram_TF_alg": rows_max * fl / t_gram / 1e12,
            "stats_GBps": rows_max * 4 * (p + 1) / t_stats / 1e9, "map_fit_s": t_map, "map_rows_per_s": n / t_map,
            "map_TF_alg_incl_generation": n * fl / t_map / 1e12, "reduce_lars_s": t_rest, "peak_device_bytes": peak,
            "status_ok": all(v == 0 for v in mb.status), "generation_overlapped": overlap, "stream_kind": kind, "roofline": rl,
            "theta_err_linf": float(np.max(np.abs(out["beta_byOLS"].to_numpy() - truth)))}


def main():
    class _P(list):
        def append(self, r):
            print(json.dumps(r), flush=True)
    rows = _P()
    sel = set(sys.argv[1:]) or {"C1", "C2", "C2b", "C3", "C3b", "C3c", "C4", "C5"}
    if "C1" in sel:
        rows.append(logistic_config("C1 n=1e5 p=50 K=20 (uniform, reference simulated_pdf)", 100_000, 50, 20, kind=engine.SYNTH_UNIFORM))
    if "C2" in sel:
        rows.append(logistic_config("C2 n=1e7 p=100 K=1", 10_000_000, 100, 1))
    if "C2b" in sel:
        rows.append(logistic_config("C2 n=1e7 p=100 K=10", 10_000_000, 100, 10))
    if "C3" in sel:
        rows.append(logistic_config("C3 shard n=2.5e7 p=500 K=1 (one shard per GPU)", 25_000_000, 500, 1))
    if "C3b" in sel:
        rows.append(logistic_config("C3 shard n=2.5e7 p=500 K=25 (1e6 rows per partition)", 25_000_000, 500, 25))
    if "C3c" in sel:
        # the reference-faithful call at config-3 scale: partition_id = i % 25 (models.py:33) and fit_intercept (logistic_dlsa.py:80),
        # on the shard as it lies: strided partition views + implicit intercept column, no copy
        n, p, K = 25_000_000, 500, 25
        X, y = engine.synth(20260101, 0, n, p, kind=engine.SYNTH_GAUSSIAN)
        torch.cuda.synchronize(); torch.cuda.reset_peak_memory_stats()
        t_fit, mb = timed(lambda: dlsa_amd.fit_logistic_partitions(X, y, partition_num=K, fit_intercept=True), reps=2)
        t_all, _ = timed(lambda: dlsa_amd.dlsa(*(lambda o: (o.iloc[:, 2:], o["beta_byOLS"]))(dlsa_amd.dlsa_mapred(
            dlsa_amd.fit_logistic_partitions(X, y, partition_num=K, fit_intercept=True))), n, fit_intercept=True), reps=2)
        rows.append({"config": "C3 shard n=2.5e7 p=500 K=25 as i %% 25 strided views + implicit intercept (no copy of the shard)", "n": n, "p": p + 1,
                     "K": K, "dtype": "f64", "map_fit_s": t_fit, "map_reduce_lars_s": t_all, "irls_iters": mb.n_iter[:3],
                     "status_ok": all(v == 0 for v in mb.status), "peak_mem_GB": torch.cuda.max_memory_allocated() / 1e9})
        del X, y, mb
        torch.cuda.empty_cache()
    if "C4" in sel:
        n4 = 14_000_000
        rows.append(logistic_config("C4 shard airline-shaped synthetic n=1.4e7 (113.9M/8) p=260 (7 numeric + 5 factors one-hot on device) K=14", n4, 0, 14, Xy=airline_shaped(n4)))
    torch.cuda.empty_cache()
    if "C5s" in sel:
        rows.append(linear_streaming_config("C5 shard at its stated size, STREAMED: linear n=6.25e7 p=2000 fp32 K=8, 2^22-row chunks generated on the device",
                                            int(os.environ.get("DLSA_C5_ROWS", "62500000")), 2000, 8, 1 << 22))
    if "C5" in sel:
        rows.append(linear_config("C5 shard (capped) linear n=2.4e7 p=2000 fp32 K=8", 24_000_000, 2000, 8))


if __name__ == "__main__":
    main()
