"""TEST INFRASTRUCTURE ONLY -- generate tests/golden/*.npz by RUNNING THE REFERENCE here.

Run in the build container (needs /root/reference):  python oracle/make_golden.py
The fixtures are data: seeded inputs + the reference's outputs.  No reference source is
copied.  Fixture families:

  F1  logistic_model (dlsa/models.py:42-147) per partition, two tiers:
        "shipped": sklearn default tol=1e-4 as the reference ships (sanity tier, ~1e-3)
        "mle":     the same call with a tol=1e-15 shim (exact MLE; the parity tier)
      cases: uniform synthetic n=2000 p=5 K=4; n=20000 p=50 K=4; games-expand.csv
      (projects/results/data/games-expand.csv, 22905x10) K=1 and K=5, with/without intercept
  F2  dlsa_mapred (dlsa/dlsa.py:21-61) on the F1 blocks and on random SPD blocks
  F3  lars_lsa (dlsa/lsa.py:90-212) full paths, 'lar' and 'lasso', p in {6, 50, 100}
"""
import os
import sys
import warnings

import numpy as np
import pandas as pd

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import dlsa_oracle as orc          # noqa: E402
from oracle.ref_shims import FakeSparkDF, load_reference, REFERENCE_ROOT  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def run_logistic_model(ref, X, y, K, fit_intercept, names):
    n = X.shape[0]
    pid = np.arange(n) % K                      # models.py:33
    df = pd.DataFrame(np.column_stack([pid, y, X]), columns=["partition_id", "label"] + names)
    outs = []
    for k in range(K):
        sub = df[df.partition_id == k].reset_index(drop=True)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            outs.append(ref.models.logistic_model(sub, "label", fit_intercept=fit_intercept))
    return outs


def blocks_to_arrays(outs):
    coef = np.stack([o["coef"].to_numpy() for o in outs])
    smc = np.stack([o["Sig_invMcoef"].to_numpy() for o in outs])
    sig = np.stack([o.iloc[:, 3:].to_numpy() for o in outs])
    return coef, smc, sig


def main():
    os.makedirs(OUT, exist_ok=True)
    ref_ship = load_reference(tol=None)
    games = pd.read_csv(os.path.join(REFERENCE_ROOT, "projects/results/data/games-expand.csv"))
    print("games columns:", list(games.columns)[:12], games.shape)

    cases = []
    for (n, p, K, seed) in [(2000, 5, 4, 11), (20000, 50, 4, 12)]:
        X, y = orc.synth_logistic(seed, 0, n, p, orc.SYNTH_UNIFORM)
        cases.append(("synth_n%d_p%d_K%d" % (n, p, K), X, y, K, False, seed))
        if p == 5:
            cases.append(("synth_n%d_p%d_K%d_icpt" % (n, p, K), X, y, K, True, seed))

    ycol = "label"
    gX = games.drop(columns=[ycol]).to_numpy(dtype=np.float64)
    gy = games[ycol].to_numpy(dtype=np.float64)
    for K in (1, 5):
        for icpt in (False, True):
            cases.append(("games_K%d%s" % (K, "_icpt" if icpt else ""), gX, gy, K, icpt, -1))
    # games-expand input itself is a data file the reference holds: keep a copy as fixture
    np.savez_compressed(os.path.join(OUT, "games_expand_input.npz"),
                        X=gX.astype(np.int8), y=gy.astype(np.int8))

    for tier, tol in (("shipped", None), ("mle", 1e-15)):
        ref = load_reference(tol=tol)
        for name, X, y, K, icpt, seed in cases:
            names = ["x%d" % i for i in range(X.shape[1])]
            outs = run_logistic_model(ref, X, y, K, icpt, names)
            coef, smc, sig = blocks_to_arrays(outs)
            mapped = pd.concat(outs, ignore_index=True)
            mr = ref.dlsa.dlsa_mapred(FakeSparkDF(mapped, K))
            np.savez_compressed(
                os.path.join(OUT, "F1_%s_%s.npz" % (name, tier)),
                seed=seed, n=X.shape[0], p=X.shape[1], K=K, fit_intercept=icpt,
                coef=coef, Sig_invMcoef=smc, Sig_inv=sig,
                columns=np.array(list(outs[0].columns)),
                mapred_columns=np.array(list(mr.columns)),
                beta_byOLS=mr["beta_byOLS"].to_numpy(),
                beta_byONESHOT=mr["beta_byONESHOT"].to_numpy(),
                Sig_inv_sum=mr.iloc[:, 2:].to_numpy())
            print("F1", name, tier, "coef[0][:3]=", coef[0][:3])

    # F2: dlsa_mapred on random SPD blocks (K=3, p=4)
    ref = ref_ship
    rng = np.random.default_rng(2024)
    K, p = 3, 4
    frames, coefs, smcs, sigs = [], [], [], []
    for k in range(K):
        A = rng.standard_normal((p + 3, p))
        S = A.T @ A
        c = rng.standard_normal(p)
        coefs.append(c); sigs.append(S); smcs.append(S @ c)
        f = pd.DataFrame(np.column_stack([np.arange(p), c, S @ c, S]),
                         columns=["par_id", "coef", "Sig_invMcoef"] + ["x%d" % i for i in range(p)])
        frames.append(f)
    mr = ref.dlsa.dlsa_mapred(FakeSparkDF(pd.concat(frames, ignore_index=True), K))
    np.savez_compressed(os.path.join(OUT, "F2_spd_K3_p4.npz"),
                        coef=np.stack(coefs), Sig_invMcoef=np.stack(smcs), Sig_inv=np.stack(sigs),
                        beta_byOLS=mr["beta_byOLS"].to_numpy(),
                        beta_byONESHOT=mr["beta_byONESHOT"].to_numpy(),
                        Sig_inv_sum=mr.iloc[:, 2:].to_numpy())

    # F3: lars_lsa paths.  Sigma = a logistic Hessian-like SPD matrix, b = noisy sparse vector
    for p in (6, 50, 100):
        rng = np.random.default_rng(300 + p)
        n = 40 * p
        X = rng.random((n, p)) - 0.5
        w = rng.random(n) * 0.25
        S = X.T @ (w[:, None] * X)
        b = orc.true_beta(p) + 0.3 * rng.standard_normal(p) / np.sqrt(n / 50)
        for typ in ("lar", "lasso"):
            r = ref.lsa.lars_lsa(np.matrix(S), b, False, n, type=typ)
            np.savez_compressed(os.path.join(OUT, "F3_lars_p%d_%s.npz" % (p, typ)),
                                Sigma=S, b=b, n=n,
                                AIC=np.asarray(r["AIC"]).ravel(), BIC=np.asarray(r["BIC"]).ravel(),
                                beta=np.asarray(r["beta"]), beta0=np.asarray(r["beta0"]).ravel())
            print("F3", p, typ, "steps=", np.asarray(r["beta"]).shape[0] - 1)
    # lasso cases engineered to produce drops (steps > p): low-rank-plus-noise designs
    for seed in (4, 22):
        rng = np.random.default_rng(seed)
        p, n = 10, 60
        Z = rng.standard_normal((n, 2))
        X = Z @ rng.standard_normal((2, p)) + 0.2 * rng.standard_normal((n, p))
        S = X.T @ X
        b = rng.standard_normal(p)
        for typ in ("lar", "lasso"):
            r = ref.lsa.lars_lsa(np.matrix(S), b, False, n, type=typ)
            np.savez_compressed(os.path.join(OUT, "F3_lars_drop%d_p%d_%s.npz" % (seed, p, typ)),
                                Sigma=S, b=b, n=n,
                                AIC=np.asarray(r["AIC"]).ravel(), BIC=np.asarray(r["BIC"]).ravel(),
                                beta=np.asarray(r["beta"]), beta0=np.asarray(r["beta0"]).ravel())
            print("F3 drop", seed, typ, "steps=", np.asarray(r["beta"]).shape[0] - 1)


if __name__ == "__main__":
    main()
