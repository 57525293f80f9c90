"""Synthetic stand-ins for the BASELINE.json configurations whose data is not available offline.

`airline_shaped` -- config 4 (the airline on-time data of the paper, projects/README.md:27-31; settings at
projects/logistic_dlsa.py:157-165): 7 numeric columns + 5 categorical factors with Zipf level frequencies and
level counts (11, 6, 20, 110, 110) mirroring dummy_keep_top = [1, 1, 0.8, 0.9, 0.9]; baseline level 0 of every
factor dropped, intercept first: p = 1 + 7 + 10 + 5 + 19 + 109 + 109 = 260.
Everything is generated on the device; the dense matrix is only built on request (dlsa_design_f64)."""
import torch

from dlsa_amd import engine

AIRLINE_LEVELS = (11, 6, 20, 110, 110)
AIRLINE_NUMERIC = 7


def airline_shaped(n, seed=7, dense=True, device="cuda"):
    """Returns a dict: num [n,7] fp64, codes [n,5] int32, y [n], beta [p], spec (kind, src, level, shift, scale device
    arrays for engine.design), plan (engine.OnehotPlan), p, and -- with dense=True -- X [n,p] built by the design kernel."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    q, levels = AIRLINE_NUMERIC, AIRLINE_LEVELS
    num = torch.randn((n, q), dtype=torch.float64, device=device, generator=g) * 3.0 + 1.5
    codes = torch.empty((n, len(levels)), dtype=torch.int32, device=device)
    for fi, L in enumerate(levels):
        pr = 1.0 / torch.arange(1, L + 1, dtype=torch.float64, device=device)
        codes[:, fi] = torch.multinomial(pr / pr.sum(), n, replacement=True, generator=g).int()
    kind, src, level, shift, scale = [0], [0], [0], [0.0], [1.0]
    for j in range(q):
        kind.append(1); src.append(j); level.append(0); shift.append(1.5); scale.append(3.0)
    for fi, L in enumerate(levels):
        for lv in range(1, L):                    # level 0 = baseline
            kind.append(2); src.append(fi); level.append(lv); shift.append(0.0); scale.append(1.0)
    p = len(kind)
    d = lambda a, t: torch.tensor(a, dtype=t, device=device)
    spec = (d(kind, torch.int32), d(src, torch.int32), d(level, torch.int32), d(shift, torch.float64), d(scale, torch.float64))
    dense_idx = [j for j in range(p) if kind[j] in (0, 1)]
    level_col, pos = [], 1 + q
    for L in levels:
        level_col += [-1] + list(range(pos, pos + L - 1))
        pos += L - 1
    plan = engine.OnehotPlan(p, [kind[j] for j in dense_idx], [src[j] for j in dense_idx], [shift[j] for j in dense_idx],
                             [scale[j] for j in dense_idx], dense_idx, list(levels), level_col)
    beta = torch.randn(p, dtype=torch.float64, device=device, generator=g) * 0.15
    # labels from the structured representation: eta = dense part + one gathered coefficient per factor
    eta = beta[0] + ((num - 1.5) / 3.0) @ beta[1:1 + q]
    pos = 1 + q
    for fi, L in enumerate(levels):
        tab = torch.cat([torch.zeros(1, dtype=torch.float64, device=device), beta[pos:pos + L - 1]])
        eta = eta + tab[codes[:, fi].long()]
        pos += L - 1
    y = (torch.rand(n, dtype=torch.float64, device=device, generator=g) < torch.sigmoid(eta)).double()
    out = {"num": num, "codes": codes, "y": y, "beta": beta, "spec": spec, "plan": plan, "p": p, "levels": levels}
    if dense:
        X, seen = engine.design(num, codes, *spec)
        out["X"], out["seen"] = X, seen
    return out
