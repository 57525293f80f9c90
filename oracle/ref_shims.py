"""TEST INFRASTRUCTURE ONLY -- harness that imports the *reference* (feng-li/dlsa at
/root/reference) in THIS container so golden vectors can be generated from it.

Nothing here ships: it refuses to run when /root/reference is absent (the GPU box), it
never writes into the reference tree (sys.dont_write_bytecode), and it copies no
reference source.  It only installs harness-side shims so that the unmodified reference
modules import and run on numpy 2 / pandas 2 / scikit-learn 1.7:

  * np.float / np.NAN aliases            (dlsa/lsa.py:12,28,90 use the removed names)
  * stub pyspark modules in sys.modules  (dlsa/dlsa.py:14-15 import them; no JVM here)
  * dlsa.models.LogisticRegression       penalty="none" -> None, optional tol override
                                         (dlsa/models.py:110-112)
  * dlsa.models.pd proxy                 pd.concat(objs, 1) positional axis
                                         (dlsa/models.py:142)
  * a pandas-backed fake Spark DataFrame for dlsa_mapred (dlsa/dlsa.py:30-34,51-52)
"""
import os
import sys
import types

REFERENCE_ROOT = "/root/reference"


def reference_available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "dlsa"))


def load_reference(tol=None):
    """Import the reference package with shims.  tol=None keeps sklearn's default
    (1e-4, what the reference ships with); tol=1e-15 drives newton-cg to the exact MLE."""
    if not reference_available():
        raise RuntimeError("reference tree not present; golden vectors can only be "
                           "generated in the build container")
    sys.dont_write_bytecode = True
    import numpy as np
    import pandas as pd
    if not hasattr(np, "float"):
        np.float = float
    if not hasattr(np, "NAN"):
        np.NAN = np.nan
    if not hasattr(np, "row_stack"):
        np.row_stack = np.vstack

    for name in ("pyspark", "pyspark.sql", "pyspark.sql.types", "pyspark.sql.functions"):
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__all__ = []
            sys.modules[name] = m
    sys.modules["pyspark.sql.functions"].pandas_udf = lambda *a, **k: (lambda f: f)
    sys.modules["pyspark.sql.functions"].PandasUDFType = types.SimpleNamespace(GROUPED_MAP=0)

    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import dlsa.models as ref_models
    import dlsa.lsa as ref_lsa
    import dlsa.dlsa as ref_dlsa
    import dlsa.dummies as ref_dummies
    from sklearn.linear_model import LogisticRegression as _LR

    class _ShimLR(_LR):
        def __init__(self, penalty=None, **kw):
            if penalty == "none":
                penalty = None
            if tol is not None:
                kw["tol"] = tol
                kw["max_iter"] = max(kw.get("max_iter", 100), 10000)
            super().__init__(penalty=penalty, **kw)

    class _PdProxy:
        def __getattr__(self, k):
            return getattr(pd, k)

        @staticmethod
        def concat(objs, axis=0, **kw):
            return pd.concat(objs, axis=axis, **kw)

    ref_models.LogisticRegression = _ShimLR
    ref_models.pd = _PdProxy()
    return types.SimpleNamespace(models=ref_models, lsa=ref_lsa, dlsa=ref_dlsa, dummies=ref_dummies)


class FakeSparkDF:
    """Pandas-backed stand-in for the Spark DataFrame dlsa_mapred consumes
    (dlsa/dlsa.py:30-34 groupby('par_id').sum(*cols).toPandas(); :52 rdd.getNumPartitions)."""

    def __init__(self, pdf, num_partitions):
        self._pdf = pdf
        self.columns = list(pdf.columns)
        self.rdd = types.SimpleNamespace(getNumPartitions=lambda: num_partitions)

    def groupby(self, key):
        outer = self

        class _G:
            def sum(self, *cols):
                g = outer._pdf.groupby(key)[list(cols)].sum().reset_index()
                g.columns = [key] + ["sum(%s)" % c for c in cols]
                return types.SimpleNamespace(toPandas=lambda: g)

        return _G()
