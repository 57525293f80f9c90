"""Level selection for categorical factors -- host mirror of the reference's dlsa/dummies.py.

`dummy_factors_counts` (dummies.py:10-33), `cumsum_dicts` (:36-49), `select_dummy_factors` (:52-108) and
`select_dummy_factors_from_file` (:111-147) keep the reference's names, arguments and the `dummy_info` dictionary
(`factor_set`, `factor_selected`, `factor_dropped`, `factor_selected_names`) that `logistic_model` consumes
(models.py:56-77).  This is string bookkeeping and stays on the host; `level_counts_from_codes` is the device-side
counter for a shard whose categorical columns already live in HBM as level codes."""
import os
import pickle
from collections import Counter

import numpy as np
import pandas as pd


def dummy_factors_counts(pdf, dummy_columns):
    """Level -> count for every dummy column of a frame (dummies.py:10-33).  `dummy_columns` holds column names or
    (all-integer) column positions.  Counts are ordered most frequent first, as value_counts() gives them."""
    cols = list(pdf.columns)
    if all(isinstance(c, int) for c in dummy_columns):
        names = [cols[c] for c in dummy_columns]
    else:
        names = list(dummy_columns)
    out = {}
    for c in names:
        vc = pdf[c].value_counts()
        out[c] = vc[vc > 0].to_dict()           # (a categorical column also lists levels that no row holds any more)
    return out


def cumsum_dicts(dict1, dict2):
    """Merge two {column: {level: count}} dictionaries, adding the counts of equal levels (dummies.py:36-49)."""
    if len(dict1) == 0:
        return dict2
    if len(dict2) == 0:
        return dict1
    return {c: dict(Counter(dict1[c]) + Counter(dict2[c])) for c in dict1.keys()}


def select_dummy_factors(dummy_dict, keep_top, replace_with, pickle_file=None):
    """Keep, per factor, the leading levels whose cumulative share of the rows stays <= keep_top[i] (in the order the
    counts are given: most frequent first when they come from dummy_factors_counts); the rest are `dropped` and will be
    folded into the level `replace_with` (dummies.py:52-108).  Returns the reference's dummy_info dictionary and, when
    `pickle_file` is given, pickles it there as the reference does (:105-106)."""
    factor_set, factor_selected, factor_dropped, factor_selected_names = {}, {}, {}, {}
    for i, col in enumerate(list(dummy_dict)):
        levels = list(dummy_dict[col].keys())
        counts = np.asarray(list(dummy_dict[col].values()), dtype=np.float64)
        share = np.cumsum(counts) / np.sum(counts)
        keep = share <= keep_top[i]
        arr = np.array(levels)
        factor_set[col] = levels
        factor_selected[col] = list(arr[keep])
        factor_dropped[col] = list(arr[~keep])
        new = ([replace_with] if (~keep).any() else []) + factor_selected[col]
        factor_selected_names[col] = [col + "_" + str(x) for x in new]
    dummy_info = {"factor_set": factor_set, "factor_selected": factor_selected, "factor_dropped": factor_dropped,
                  "factor_selected_names": factor_selected_names}
    if pickle_file:
        # written next to its place and renamed: another rank (or job) that finds the file finds all of it
        path = os.path.expanduser(pickle_file)
        tmp = "%s.tmp.%d" % (path, os.getpid())
        with open(tmp, "wb") as f:
            pickle.dump(dummy_info, f)
        os.replace(tmp, path)
        print("dummy_info saved in:\t" + pickle_file)
    return dummy_info


def select_dummy_factors_from_file(file, header, dummy_columns, keep_top, replace_with, pickle_file=None,
                                   chunk_bytes=1024000):
    """Memory-bounded level selection from a large CSV file: count levels buffer by buffer, then select
    (dummies.py:111-147).  Fields are split on ',' and kept as strings, as in the reference."""
    dummy_dict = {}
    names = None
    with open(os.path.expanduser(file)) as f:
        first = True
        while True:
            buf = f.readlines(chunk_bytes)
            if not buf:
                break
            rows = [x.strip().split(",") for x in buf]
            if first and header is True:
                names, rows = rows[0], rows[1:]
            first = False
            pdf = pd.DataFrame(rows)
            if names is not None:
                pdf.columns = names
            dummy_dict = cumsum_dicts(dummy_dict, dummy_factors_counts(pdf, dummy_columns))
    return select_dummy_factors(dummy_dict, keep_top, replace_with, pickle_file)


def level_counts_from_codes(codes, levels):
    """Counts of a device-resident shard: codes [n, f] int32 (negative = unknown level), levels = {factor: [level, ...]}
    in code order.  Returns {factor: {level: count}} ordered most frequent first (ties: lower code first), ready for
    select_dummy_factors; across ranks, all-reduce the bincounts (or merge with cumsum_dicts)."""
    out = {}
    host = codes.cpu().numpy() if hasattr(codes, "cpu") else np.asarray(codes)      # counting is host work (tensors are storage)
    for t, (col, lv) in enumerate(levels.items()):
        c = host[:, t]
        cnt = np.bincount(c[c >= 0].astype(np.int64), minlength=len(lv))
        order = np.lexsort((np.arange(len(lv)), -cnt))
        out[col] = {lv[j]: int(cnt[j]) for j in order if cnt[j] > 0}
    return out
