"""Wide Newton pass (dlsa_newton_wide_pass_f64): g / loglik / w against the logit pass, H~ against the fp64 Gram, per width; timing
against the logit pass it replaces.  usage: wide_quick.py [n] [p ...]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dlsa_amd import engine


def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000
for spec in (sys.argv[2:] or ["500", "500i", "499", "512", "260", "130", "384i", "256i"]):
    icpt = spec.endswith("i")
    p = int(spec.rstrip("i"))
    ldx = p + (p & 1)
    Xfull, y = engine.synth(20260101, 0, n, ldx, kind=engine.SYNTH_GAUSSIAN)
    X = Xfull[:, :p]
    pe = p + (1 if icpt else 0)
    beta = torch.zeros(pe, dtype=torch.float64, device="cuda"); beta[(1 if icpt else 0): (1 if icpt else 0) + int(0.4 * p)] = 0.9
    if icpt:
        beta[0] = 0.3
    # labels from the model at these coefficients keep the gradient small-ish; here they come from the generator's own beta
    w0, g0, ll0 = engine.logit_pass(X, y, beta, fit_intercept=icpt)
    H0 = engine.gram_icpt(X, w0) if icpt else engine.gram(X, w0)
    H, g, ll, w = engine.newton_wide_pass(X, y, beta, fit_intercept=icpt, want_w=True)
    torch.cuda.synchronize()
    d = torch.sqrt(torch.diag(H0))
    errH = float(((H - H0).abs() / (d[:, None] * d[None, :])).max())
    # spectral quality as a preconditioner: eigenvalues of H0^-1 H~ - I
    L = torch.linalg.cholesky(H0)
    M = torch.linalg.solve_triangular(L, torch.linalg.solve_triangular(L, H, upper=False).T, upper=False)
    ev = torch.linalg.eigvalsh((M + M.T) / 2)
    errg = float((g - g0).abs().max() / g0.abs().max())
    errw = float((w - w0).abs().max())
    errl = abs(float(ll) - float(ll0)) / abs(float(ll0))
    sym = bool(torch.equal(H, H.T))
    tl = t(lambda: engine.logit_pass(X, y, beta, fit_intercept=icpt))
    tw = t(lambda: engine.newton_wide_pass(X, y, beta, fit_intercept=icpt, want_w=True))
    tw0 = t(lambda: engine.newton_wide_pass(X, y, beta, fit_intercept=icpt))
    print("p=%4s n=%.0e  logit %.3f ms (%.2f TB/s)  wide %.3f ms (no w: %.3f; %.2fx)  H~ entry err %.2e  spectrum of H^-1 H~ in [%.6f, %.6f]  g %.1e  w %.1e  ll %.1e  sym %s" % (
        spec, n, tl, n * p * 8 / tl * 1e-9, tw, tw0, tw / tl, errH, float(ev[0]), float(ev[-1]), errg, errw, errl, sym), flush=True)
    del Xfull, X, y, w0, H0, H, w
