"""dlsa_amd -- MI355X-native engine for the DLSA hot path (feng-li/dlsa):
per-partition logistic fit -> X'WX / X'W theta -> one-round sum -> WLS combine -> LARS shrinkage.

The compute path is hand-written HIP for gfx950 behind the C ABI in include/dlsa_hip.h
(libdlsa_hip.so); this package is the Python host side mirroring the reference's operator
interface (dlsa/models.py, dlsa/dlsa.py, dlsa/lsa.py).  There is no CPU fallback.
"""
__version__ = "0.1.0"

from .models import (MappedBlocks, fit_linear_chunks, fit_linear_partitions, fit_linear_streaming, fit_logistic_design, fit_logistic_partitions, linear_model,   # noqa: E402,F401
                     logistic_model, logistic_model_eval, simulate_logistic)
from .design import DesignSpec, design_matrix                                                    # noqa: E402,F401
from .dlsa import dlsa, dlsa_fit, dlsa_mapred, dlsa_mapreduce                                   # noqa: E402,F401
from .lsa import lars_lsa                                                                        # noqa: E402,F401
from .dummies import (cumsum_dicts, dummy_factors_counts, select_dummy_factors,                  # noqa: E402,F401
                      select_dummy_factors_from_file)
from .model_eval import logistic_model_eval_sdf, loglik_partitions                               # noqa: E402,F401
from .results import coef_table, write_coef_csv                                                  # noqa: E402,F401
