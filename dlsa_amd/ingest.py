"""CSV ingestion for row shards: the reference's real-data branch reads the airline CSV, keeps
`usecols_x + [Y_name]`, drops rows with missing values, binarises the response (ArrDelay > 0) and deals the rows to
partitions with partition_id = row % K (projects/logistic_dlsa.py:108-110,218-237; the pandas variant is
dlsa/utils.py:8-45 `clean_airlinedata`).  Here the same steps produce what the device path consumes: numeric columns as
one fp64 array, categorical columns as int32 LEVEL CODES (DesignSpec.encode: dropped levels fold into 000_OTHERS,
models.py:60), labels, and the rows of this rank's partitions grouped contiguously.  Parsing is pandas' C reader on the
host; everything after it (one-hot, standardise, fit) runs on the GPU from the codes."""
import math

import numpy as np
import pandas as pd

from .design import DesignSpec

# the airline schema of logistic_dlsa.py:108-140 restricted to usecols_x + Y
AIRLINE_USECOLS_X = ["Year", "Month", "DayofMonth", "DayOfWeek", "DepTime", "CRSDepTime", "CRSArrTime", "UniqueCarrier",
                     "ActualElapsedTime", "Origin", "Dest", "Distance"]
AIRLINE_DUMMY_COLUMNS = ["Month", "DayOfWeek", "UniqueCarrier", "Origin", "Dest"]          # logistic_dlsa.py:156
AIRLINE_Y = "ArrDelay"                                                                     # logistic_dlsa.py:172


def read_csv_frame(path, usecols_x, Y_name, dummy_columns=(), binarize=True, nrows=None, engine="auto"):
    """select(usecols_x + [Y_name]).dropna() and the 0/1 response (logistic_dlsa.py:222-231).  Categorical columns are read
    as strings (integer-looking levels keep their text, e.g. Month '1'), the rest as float64.
    engine: "pyarrow" parses with Arrow's multi-threaded CSV reader and hands the categorical columns over DICTIONARY-ENCODED
    (pandas `category`: DesignSpec.encode then takes their codes as they are, no per-row string hashing); "pandas" is the
    single-threaded C parser with object columns; "auto" = pyarrow when importable and nrows is None.  Same rows, same values,
    same level strings either way (tests/test_host_api_cpu.py)."""
    cols = list(usecols_x) + [Y_name]
    use_arrow = engine == "pyarrow"
    if engine == "auto" and nrows is None:
        try:
            import pyarrow  # noqa: F401
            use_arrow = True
        except ImportError:
            use_arrow = False
    if use_arrow:
        import pyarrow as pa
        import pyarrow.csv as pacsv
        # dictionary types AT READ TIME: the parser builds the dictionaries block by block on its threads (3e6 rows x 6 columns here:
        # 1.9 s against 4.3 s for read + dictionary_encode, 8.6 s for pandas' parser with exact float conversion)
        types = {c: (pa.dictionary(pa.int32(), pa.string()) if c in dummy_columns else pa.float64()) for c in cols}
        # Missing values as the reference's reader sees them (spark.read.csv with an explicit schema, logistic_dlsa.py:218-237, then
        # dropna()): a NUMERIC field that does not parse is null whatever its spelling ("NA", "NULL", "nan", ...), but in a STRING
        # (factor) column only the empty field is null -- "NA" or "None" there is a level like any other, and its rows stay.  Arrow's
        # null spellings are global, so string columns are read with strings_can_be_null = False and their empty fields dropped below.
        tab = pacsv.read_csv(path, convert_options=pacsv.ConvertOptions(include_columns=cols, column_types=types,
                                                                          null_values=list(_NA_VALUES), strings_can_be_null=False))
        pdf = tab.drop_null().unify_dictionaries().to_pandas()   # dictionary columns -> pandas category, doubles stay columnar
        # a NaN that reached a double column through another spelling ("NAN", "+nan") is a value to Arrow, a missing one to dropna()
        ok = None
        for c in cols:
            if c not in dummy_columns:
                m = ~np.isnan(pdf[c].to_numpy(dtype=np.float64))
            else:
                m = (pdf[c].astype(object) != "").to_numpy()
            ok = m if ok is None else (ok & m)
        if ok is not None and not ok.all():
            pdf = pdf[ok].reset_index(drop=True)
            for c in dummy_columns:
                if c in pdf.columns and hasattr(pdf[c], "cat"):
                    pdf[c] = pdf[c].cat.remove_unused_categories()
    else:
        dtypes = {c: "str" for c in dummy_columns}
        na = {c: ([""] if c in dummy_columns else list(_NA_VALUES)) for c in cols}          # (per column: see the Arrow branch)
        pdf = pd.read_csv(path, usecols=cols, dtype=dtypes, nrows=nrows, engine="c", na_values=na, keep_default_na=False, float_precision="round_trip")
        pdf = pdf.dropna().reset_index(drop=True)
        for c in usecols_x:
            if c not in dummy_columns:
                pdf[c] = pdf[c].astype(np.float64)
    if binarize:
        pdf[Y_name] = (pdf[Y_name].astype(np.float64) > 0).astype(np.float64)          # F.when(Y > 0, 1).otherwise(0)
    return pdf[cols]


# what pandas.read_csv treats as missing by default (keep_default_na) + "NA" -- the spellings of a missing NUMERIC field, passed to
# both readers (a factor column's only missing value is the empty field) so that the rows that reach
# the fit, hence n, K = ceil(n / 1e6) and the dummy counts, do not depend on which reader is installed
_NA_VALUES = ("", "#N/A", "#N/A N/A", "#NA", "-1.#IND", "-1.#QNAN", "-NaN", "-nan", "1.#IND", "1.#QNAN", "<NA>", "N/A", "NA", "NULL",
              "NaN", "None", "n/a", "nan", "null")


def data_info_from_frame(pdf, numeric_cols):
    """The three rows of Spark's describe() that models.py:99-101 reads: count, mean, stddev (sample std, ddof = 1)."""
    # (numpy on the columns' own memory: the frame was dropna()'d, and pandas' skipna reductions on a copied sub-frame cost as much
    #  as parsing the file)
    out = {}
    for c in numeric_cols:
        a = pdf[c].to_numpy(dtype=np.float64)
        out[c] = [str(len(a)), float(a.mean()) if len(a) else float("nan"), float(a.std(ddof=1)) if len(a) > 1 else float("nan")]
    return pd.DataFrame(out)


def shard_from_frame(pdf, Y_name, dummy_info, dummy_factors_baseline, data_info, fit_intercept, sample_size_per_partition=1000000,
                     world=1, rank=0, device="cuda"):
    """Frame -> device-resident raw shard of this rank.  partition_num = ceil(n / sample_size_per_partition)
    (logistic_dlsa.py:231-232), partition_id = row % partition_num (:235-237); rank r owns the partitions k % world == r,
    each stored as a contiguous row range.  Returns dict(num, codes, y, part_offsets, spec, partition_num, sample_size,
    partitions)."""
    import torch
    n = len(pdf)
    K = max(1, int(math.ceil(n / float(sample_size_per_partition))))
    spec = DesignSpec.from_reference(list(pdf.columns), Y_name, fit_intercept, dummy_info, dummy_factors_baseline, data_info)
    # the frame goes to the device as pandas holds it (DesignSpec.numeric_to_device: per column / one strided window), the rows of this
    # rank's partitions are then grouped THERE (an HBM gather instead of a host fancy-index over every column)
    _, codes, unknown = spec.encode(pdf, dummy_info, numeric=False)
    num_d = spec.numeric_to_device(pdf, device)
    codes_d = torch.from_numpy(codes).to(device)
    y_d = torch.from_numpy(np.ascontiguousarray(pdf[Y_name].to_numpy(dtype=np.float64))).to(device)
    mine = [k for k in range(K) if k % world == rank]
    # partition_id = row % K: the rows of partition k are k, k + K, k + 2K, ... (O(n) in all, not one pass over the ids per partition)
    order = torch.cat([torch.arange(k, n, K, dtype=torch.int64, device=device) for k in mine]) if mine else torch.zeros(0, dtype=torch.int64, device=device)
    offs = np.concatenate([[0], np.cumsum([len(range(k, n, K)) for k in mine])]).astype(np.int64)
    if num_d is None:
        num_d = torch.zeros((n, 0), dtype=torch.float64, device=device)
    t = lambda a: a.index_select(0, order).contiguous()
    return {"num": t(num_d), "codes": t(codes_d), "y": t(y_d), "part_offsets": offs, "spec": spec,
            "partition_num": K, "sample_size": n, "partitions": mine, "unknown_levels": bool(unknown)}
