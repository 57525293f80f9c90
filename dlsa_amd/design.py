"""Design-matrix construction on the GPU.

The reference builds each chunk's design matrix with pandas inside `logistic_model`
(dlsa/models.py:56-104): fold the dropped levels of every factor into "000_OTHERS", one-hot encode
(`pd.get_dummies`), drop the baseline dummies, standardise the numeric columns with the global
mean / std (`data_info`, rows 1 and 2 of Spark's describe()), `reindex` to the canonical column
order (sorted numeric names, then each factor's sorted dummy names), and -- for the Hessian -- prepend
a ones column (:121-122).  Here the host only turns the categorical strings into integer level codes;
the n x p matrix itself is written by `dlsa_design_f64` on the device, so what crosses PCIe is
n x (q numeric + f codes) instead of n x p, and a shard whose codes are already in HBM never touches
the host at all.
"""
import warnings

import numpy as np
import pandas as pd
import torch

from . import engine

OTHERS = "000_OTHERS"      # models.py:60


def _row_major(t):
    """[n, q] with strides (q, 1) exactly (`.contiguous()` keeps whatever stride a size-1 dimension has)."""
    if t.stride(1) == 1 and t.stride(0) == t.shape[1]:
        return t
    out = torch.empty(t.shape, dtype=t.dtype, device=t.device)
    out.copy_(t)
    return out


class DesignSpec:
    """Column plan of the design matrix.  `names` are the output columns in the reference's order
    (`["intercept"] +` usecols_x of models.py:70-77); numeric_cols / factors say where each comes from."""

    def __init__(self, numeric_cols, factors, levels, names, kind, src, level, shift, scale, dummy_cols):
        self.numeric_cols = list(numeric_cols)      # sorted raw numeric column names (usecols_x0, models.py:70)
        self.factors = list(factors)                # categorical columns in dummy_info["factor_selected"] order
        self.levels = levels                        # factor -> list of level strings; code = position
        self.names = list(names)
        self.kind, self.src, self.level = kind, src, level
        self.shift, self.scale = shift, scale
        self.dummy_cols = list(dummy_cols)          # output positions of the dummy columns
        self._dev = {}

    @property
    def p(self):
        return len(self.names)

    @classmethod
    def from_reference(cls, columns, Y_name, fit_intercept=False, dummy_info=[], dummy_factors_baseline=[],
                       data_info=[]):
        """Same arguments as logistic_model (models.py:42); `columns` = the chunk frame's columns."""
        columns = [c for c in columns if c not in ("partition_id", Y_name)]
        factors = list(dummy_info["factor_selected"].keys()) if len(dummy_info) > 0 else []
        if len(dummy_info) > 0:
            numeric_cols = sorted(set(columns) - set(factors))                       # models.py:70
            out_numeric = list(numeric_cols)
        else:
            numeric_cols = list(columns)                                             # models.py:93-95 (frame order)
            out_numeric = [c for c in numeric_cols if c not in dummy_factors_baseline]
        levels = {}
        for fct in factors:
            lv = [str(x) for x in dummy_info["factor_selected"][fct]]
            if len(dummy_info["factor_dropped"][fct]) > 0:
                lv = [OTHERS] + lv
            levels[fct] = lv
        names, kind, src, level, shift, scale, dummy_cols = [], [], [], [], [], [], []

        def push(name, k, s, l, sh=0.0, sc=1.0):
            names.append(name); kind.append(k); src.append(s); level.append(l); shift.append(sh); scale.append(sc)

        if fit_intercept:
            push("intercept", 0, 0, 0)
        for c in out_numeric:
            if len(data_info) > 0:                                                   # models.py:99-101
                push(c, 1, numeric_cols.index(c), 0, float(data_info[c][1]), float(data_info[c][2]))
            else:
                push(c, 1, numeric_cols.index(c), 0)
        for fi, fct in enumerate(factors):
            for nm in sorted(dummy_info["factor_selected_names"][fct]):              # models.py:73-75
                if nm in dummy_factors_baseline:
                    continue
                lv = nm[len(fct) + 1:]
                if lv not in levels[fct]:
                    raise ValueError("dummy column %r has no level in factor %r" % (nm, fct))
                dummy_cols.append(len(names))
                push(nm, 2, fi, levels[fct].index(lv))
        return cls(numeric_cols, factors, levels, names,
                   np.asarray(kind, np.int32), np.asarray(src, np.int32), np.asarray(level, np.int32),
                   np.asarray(shift, np.float64), np.asarray(scale, np.float64), dummy_cols)

    def _numeric_host(self, sample_df):
        """The numeric columns as one fp64 host array [n, q], in whatever order pandas hands it over (to_numpy() of a
        single-dtype frame is a column-major view of its block: engine.rows_to_device() uploads that and transposes in HBM)."""
        if not self.numeric_cols:
            return np.zeros((len(sample_df), 0))
        return sample_df[self.numeric_cols].to_numpy(dtype=np.float64)

    def encode(self, sample_df, dummy_info=[], numeric=True):
        """Host step: numeric columns as one fp64 array, categorical columns as int32 level codes
        (dropped levels fold into the OTHERS code, models.py:60; a level that is neither selected nor
        dropped gets -1 and `unknown` is set -- get_dummies would have produced an unexpected column)."""
        n = len(sample_df)
        num = np.ascontiguousarray(self._numeric_host(sample_df)) if numeric else None      # row-major for host users (ingest)
        # One hash pass per factor (pd.factorize), then the FEW distinct values are turned into strings and looked up: the level
        # of a value is what the reference's astype(str) / replace / get_dummies chain gives it, without a string operation per row
        # (1e6 rows x 5 factors: 242 -> ~60 ms).
        cols, unknown = [], False
        for fct in self.factors:
            if isinstance(sample_df[fct].dtype, pd.CategoricalDtype):
                # already dictionary-encoded: the codes are there, only the categories need a string each (NaN = one more entry)
                uniq = list(sample_df[fct].cat.categories) + [np.nan]
                fc = sample_df[fct].cat.codes.to_numpy().astype(np.int64)
                fc = np.where(fc < 0, len(uniq) - 1, fc)
            else:
                fc, uniq = pd.factorize(sample_df[fct], use_na_sentinel=False)
            dropped = set(str(x) for x in dummy_info["factor_dropped"][fct]) if len(dummy_info) > 0 else set()
            index = {lv: i for i, lv in enumerate(self.levels[fct])}
            lut = np.empty(max(len(uniq), 1), dtype=np.int32)
            lut[:] = -1
            if sample_df[fct].dtype == object and len(uniq) and bool(pd.isna(np.asarray(uniq, dtype=object)).any()):
                # None / NaN in an object column keep their own spellings under astype(str) ('None', 'nan'); factorize merges them
                col = sample_df[fct].astype(str)
                if dropped:
                    col = col.where(~col.isin(dropped), OTHERS)
                c = pd.Categorical(col, categories=self.levels[fct]).codes.astype(np.int32)
            else:
                for u, raw in enumerate(uniq):
                    sv = str(raw)
                    lut[u] = index.get(OTHERS if sv in dropped else sv, -1)
                c = lut[fc]
            unknown |= bool((c < 0).any())
            cols.append(c)
        codes = np.stack(cols, axis=1) if cols else np.zeros((n, 0), dtype=np.int32)
        return num, codes, unknown

    def onehot_plan(self):
        """engine.OnehotPlan for the structured passes (gather / histogram instead of dense rows), or None when the
        design does not qualify: more than 8 dense columns or 8 factors, or so many levels that the dense-by-level block alone
        exceeds LDS (pair tables larger than LDS are cut into row bands by the plan)."""
        if getattr(self, "_oh_plan", False) is not False:
            return self._oh_plan
        self._oh_plan = None
        dense = [j for j in range(self.p) if self.kind[j] in (0, 1)]
        if len(dense) > 8 or len(self.factors) > 8 or len(self.factors) == 0:
            return None
        nlevels = [len(self.levels[f]) for f in self.factors]
        level_col = []
        for t, fct in enumerate(self.factors):
            cols = [-1] * nlevels[t]
            for j in self.dummy_cols:
                if self.src[j] == t:
                    cols[int(self.level[j])] = j
            level_col += cols
        try:
            self._oh_plan = engine.OnehotPlan(self.p, [int(self.kind[j]) for j in dense], [int(self.src[j]) for j in dense],
                                              [float(self.shift[j]) for j in dense], [float(self.scale[j]) for j in dense],
                                              dense, nlevels, level_col)
        except Exception:
            self._oh_plan = None
        return self._oh_plan

    def numeric_to_device(self, sample_df, device="cuda"):
        """The numeric columns as a row-major fp64 device tensor [n, q] (None when q == 0) without a host-side gather: a
        frame that arrives column by column (Arrow / read_csv: every column a contiguous array) is uploaded column by column
        into a [q, n] buffer and transposed in HBM; other layouts take encode()'s array through engine.rows_to_device()."""
        q, n = len(self.numeric_cols), len(sample_df)
        if q == 0:
            return None
        if n * q >= (1 << 16):
            cols = [sample_df[c].to_numpy() for c in self.numeric_cols]            # per-column views, no copy
            if all(c.dtype == np.float64 and c.ndim == 1 and c.flags.c_contiguous for c in cols):
                buf = torch.empty((q, n), dtype=torch.float64, device=device)
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")                                   # (read-only views under copy-on-write)
                    for j, c in enumerate(cols):
                        buf[j].copy_(torch.from_numpy(c))
                return _row_major(buf.t())
            # A frame built FROM a row-major array (pd.DataFrame(ndarray), the reference's simulate_logistic): every column is a
            # strided view of the same [n, width] parent.  The window of parent columns that holds them goes up as one strided
            # tensor (torch gathers it with all host threads; pandas' column subset is a single-threaded copy, 73 ms per 1e6 x 100)
            # and the wanted columns are picked in HBM.
            strides = {c.strides[0] for c in cols}
            if len(strides) == 1 and all(c.dtype == np.float64 and c.ndim == 1 for c in cols):
                S = strides.pop()
                addr = [c.ctypes.data for c in cols]
                a0 = min(addr)
                w = (max(addr) - a0) // 8 + 1
                if S > 8 and S % 8 == 0 and w <= S // 8 and all((a - a0) % 8 == 0 for a in addr):
                    window = np.lib.stride_tricks.as_strided(cols[addr.index(a0)], shape=(n, w), strides=(S, 8), writeable=False)
                    with warnings.catch_warnings():
                        warnings.simplefilter("ignore")
                        Wd = torch.from_numpy(window).to(device)
                    idx = [(a - a0) // 8 for a in addr]
                    if idx == list(range(q)) and w == q:
                        return _row_major(Wd)
                    return _row_major(Wd.index_select(1, torch.tensor(idx, dtype=torch.int64, device=device)))
        return engine.rows_to_device(self._numeric_host(sample_df), device)

    def missing_levels(self, codes):
        """Names of the dummy columns whose level does not occur in `codes` (host array or device tensor [n, f])."""
        if torch.is_tensor(codes):
            present = [set(torch.unique(codes[:, t]).cpu().tolist()) for t in range(codes.shape[1])]
        else:
            present = []
            for t in range(codes.shape[1]):
                c = codes[:, t]
                present.append(set(np.nonzero(np.bincount(c[c >= 0], minlength=1))[0].tolist()) | ({-1} if (c < 0).any() else set()))
        return [self.names[j] for j in self.dummy_cols if int(self.level[j]) not in present[int(self.src[j])]]

    def device_arrays(self, device):
        key = str(device)
        if key not in self._dev:
            self._dev[key] = tuple(torch.from_numpy(a).to(device) for a in
                                   (self.kind, self.src, self.level, self.shift, self.scale))
        return self._dev[key]

    def build(self, num, codes, dtype=torch.float64, out=None):
        """Device step: num [n,q] / codes [n,f] (device tensors, either may be None when q or f is 0)
        -> (X [n,p], missing: names of dummy columns without any non-zero entry in this chunk)."""
        dev = num.device if num is not None else codes.device
        kind, src, level, shift, scale = self.device_arrays(dev)
        if num is not None and num.shape[1] == 0:
            num = None
        if codes is not None and codes.shape[1] == 0:
            codes = None
        X, seen = engine.design(num, codes, kind, src, level, shift, scale, dtype=dtype, out=out)
        seen = seen.cpu().numpy()
        missing = [self.names[j] for j in self.dummy_cols if not seen[j]]
        return X, missing


def design_matrix(num, codes, spec, dtype=torch.float64):
    """Tensor fast path: device-resident raw columns -> dense design matrix (see DesignSpec.build)."""
    return spec.build(num, codes, dtype=dtype)
