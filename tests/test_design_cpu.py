"""CPU: the design-matrix column plan (dlsa_amd.design.DesignSpec, host logic) and the oracle's
design_matrix restatement, pinned to the reference's own dummy / standardise path (fixture F4)."""
import numpy as np
import pandas as pd
import pytest

from oracle import dlsa_oracle as orc
from f4_fixture import load_f4

TOL_MLE = 1e-10


def rel_inf(a, b):
    return np.max(np.abs(np.asarray(a, float) - np.asarray(b, float))) / max(1e-300, np.max(np.abs(b)))


@pytest.fixture(scope="module")
def f4():
    return load_f4()


def test_column_plan_matches_reference(f4):
    from dlsa_amd.design import DesignSpec
    z, df, dummy_info, baseline, data_info = f4
    spec = DesignSpec.from_reference(list(df.columns), "label", True, dummy_info, baseline, data_info)
    assert ["par_id", "coef", "Sig_invMcoef"] + spec.names == list(z["columns"])
    assert spec.numeric_cols == ["dep", "dist"] and spec.factors == ["carrier", "dow"]
    assert spec.levels["carrier"][0] == "000_OTHERS" and "ZZ" not in spec.levels["carrier"]
    assert spec.kind.tolist() == [0, 1, 1, 2, 2, 2, 2, 2, 2]


def test_encode_folds_dropped_levels_and_flags_unknown(f4):
    from dlsa_amd.design import DesignSpec
    z, df, dummy_info, baseline, data_info = f4
    spec = DesignSpec.from_reference(list(df.columns), "label", True, dummy_info, baseline, data_info)
    num, codes, unknown = spec.encode(df, dummy_info)
    assert not unknown and num.shape == (len(df), 2) and codes.shape == (len(df), 2) and codes.dtype == np.int32
    folded = df["carrier"].isin(["ZZ", "YY"]).to_numpy()
    assert (codes[folded, 0] == 0).all() and (codes[~folded, 0] > 0).all()
    df2 = df.copy()
    df2.loc[3, "carrier"] = "QQ"                        # neither selected nor dropped
    assert spec.encode(df2, dummy_info)[2]


def test_oracle_design_and_fit_match_reference_dummy_path(f4):
    """reference logistic_model on the dummy path == oracle design_matrix + oracle fit (exact-MLE tier)."""
    from dlsa_amd.design import DesignSpec
    z, df, dummy_info, baseline, data_info = f4
    spec = DesignSpec.from_reference(list(df.columns), "label", True, dummy_info, baseline, data_info)
    num, codes, _ = spec.encode(df, dummy_info)
    X, seen = orc.design_matrix(num, codes, spec.kind, spec.src, spec.level, spec.shift, spec.scale)
    assert seen.all()
    # independent pandas construction of the same matrix
    Xp = np.column_stack([np.ones(len(df)), (df["dep"] - z["data_info_mean"][1]) / z["data_info_std"][1],
                          (df["dist"] - z["data_info_mean"][0]) / z["data_info_std"][0]]
                         + [df["carrier"] == c for c in ("AA", "BB", "CC", "DD")] + [df["dow"] == d for d in ("2", "3")])
    assert np.array_equal(X, Xp.astype(float))
    coef, smc, sig = orc.logistic_model_block(X[:, 1:], df["label"].to_numpy(), True)
    assert rel_inf(coef, z["coef_mle"]) < TOL_MLE
    assert rel_inf(sig, z["Sig_inv_mle"]) < TOL_MLE and rel_inf(smc, z["Sig_invMcoef_mle"]) < TOL_MLE
    assert rel_inf(coef, z["coef_shipped"]) < 2e-2          # the reference as shipped stops early
    ll = orc.logistic_loglik(X, df["label"].to_numpy(), np.column_stack([coef, 0.5 * coef, 0 * coef]))
    assert rel_inf(ll, z["eval_loglik"]) < TOL_MLE


def test_missing_level_is_detected_by_seen_flags(f4):
    from dlsa_amd.design import DesignSpec
    z, df, dummy_info, baseline, data_info = f4
    spec = DesignSpec.from_reference(list(df.columns), "label", True, dummy_info, baseline, data_info)
    sub = df[df["carrier"] != "CC"].reset_index(drop=True)
    num, codes, unknown = spec.encode(sub, dummy_info)
    _, seen = orc.design_matrix(num, codes, spec.kind, spec.src, spec.level, spec.shift, spec.scale)
    missing = [spec.names[j] for j in spec.dummy_cols if not seen[j]]
    assert missing == ["carrier_CC"] and not unknown
    assert list(z["zero_columns"]) == ["par_id", "coef", "Sig_invMcoef"] + spec.names and float(z["zero_absmax"]) == 0.0


def test_plain_frame_plan_keeps_frame_order():
    from dlsa_amd.design import DesignSpec
    spec = DesignSpec.from_reference(["partition_id", "label", "x2", "x0", "x1"], "label", False)
    assert spec.names == ["x2", "x0", "x1"] and spec.kind.tolist() == [1, 1, 1] and spec.src.tolist() == [0, 1, 2]
    assert spec.shift.tolist() == [0, 0, 0] and spec.scale.tolist() == [1, 1, 1]


def test_encode_level_codes_match_the_string_chain_for_every_column_type():
    """DesignSpec.encode turns a factor into level codes with one hash pass (pd.factorize, or the codes of a categorical column)
    and a lookup of the few distinct values; the result must be what the reference's per-row chain gives
    (astype(str) -> fold dropped levels into 000_OTHERS -> position among the selected levels, models.py:56-66), for strings,
    integers, floats with NaN, object columns holding None, categoricals (with and without NaN) and booleans."""
    import numpy as np
    import pandas as pd
    from dlsa_amd.design import DesignSpec, OTHERS
    rng = np.random.default_rng(0)
    n = 2000
    cases = {
        "str": pd.Series(rng.choice(["a", "b", "c", "zz"], n)),
        "int": pd.Series(rng.integers(0, 7, n)),
        "float": pd.Series(rng.choice([1.0, 2.5, 3.0, np.nan], n)),
        "objmix": pd.Series(rng.choice(np.array(["x", 1, 2.0, None], dtype=object), n)),
        "cat": pd.Series(pd.Categorical(rng.choice(["u", "v", "w"], n))),
        "catnan": pd.Series(pd.Categorical(rng.choice(np.array(["u", "v", None], dtype=object), n))),
        "bool": pd.Series(rng.random(n) < 0.5),
    }
    for name, col in cases.items():
        us = sorted(set(col.astype(str)))
        sel = us[: max(1, len(us) - 1)]
        drop = us[len(sel):]
        dummy_info = {"factor_selected": {"f": sel}, "factor_dropped": {"f": drop},
                      "factor_selected_names": {"f": ["f_" + v for v in ([OTHERS] if drop else []) + sel]}}
        df = pd.DataFrame({"partition_id": 0, "label": 1, "f": col, "x": rng.random(n)})
        spec = DesignSpec.from_reference(list(df.columns), "label", False, dummy_info, [], [])
        _, codes, unknown = spec.encode(df, dummy_info)
        chain = col.astype(str)
        chain = chain.where(~chain.isin(set(drop)), OTHERS)
        ref = pd.Categorical(chain, categories=spec.levels["f"]).codes.astype(np.int32)
        assert np.array_equal(codes[:, 0], ref), name
        assert not unknown and codes.dtype == np.int32 and codes.flags.c_contiguous
    # a value that is neither selected nor dropped: code -1 and the flag
    df = pd.DataFrame({"partition_id": 0, "label": 1, "f": ["a", "b", "q"], "x": [0.1, 0.2, 0.3]})
    info = {"factor_selected": {"f": ["a", "b"]}, "factor_dropped": {"f": []}, "factor_selected_names": {"f": ["f_a", "f_b"]}}
    spec = DesignSpec.from_reference(list(df.columns), "label", False, info, [], [])
    _, codes, unknown = spec.encode(df, info)
    assert codes[:, 0].tolist() == [0, 1, -1] and unknown
    assert spec.missing_levels(codes) == [] and spec.missing_levels(codes[:1]) == ["f_b"]


def test_numeric_columns_reach_the_device_tensor_unchanged_for_every_frame_layout():
    """DesignSpec.numeric_to_device (run here with device="cpu": the upload logic is plain torch): frames built column by column,
    from one row-major array, with scattered / mixed-dtype columns or a single feature must all give the [n, q] row-major fp64 matrix
    of the numeric columns in spec order."""
    import numpy as np
    import pandas as pd
    import torch
    from dlsa_amd import engine
    from dlsa_amd.design import DesignSpec
    rng = np.random.default_rng(9)
    for n, p in ((70000, 3), (3000, 30), (66000, 1), (50, 4)):            # above and below the 2^16-element fast paths
        X = rng.random((n, p)) - 0.5
        y = (rng.random(n) < 0.5).astype(np.int64)
        names = ["x%d" % i for i in range(p)]
        frames = {
            "row_major": pd.DataFrame(np.column_stack([np.zeros(n), y.astype(np.float64), X]), columns=["partition_id", "label"] + names),
            "columnar": pd.DataFrame({"partition_id": 0, "label": y, **{c: np.ascontiguousarray(X[:, i]) for i, c in enumerate(names)}}),
        }
        order = list(rng.permutation(names))
        frames["scattered"] = frames["columnar"][["partition_id"] + order[: p // 2] + ["label"] + order[p // 2:]]
        mixed = frames["columnar"].copy()
        mixed[names[0]] = mixed[names[0]].astype(np.float32)
        frames["mixed"] = mixed
        for layout, df in frames.items():
            spec = DesignSpec.from_reference(list(df.columns), "label", False, [], [], [])
            t = spec.numeric_to_device(df, "cpu")
            ref = df[spec.numeric_cols].to_numpy(dtype=np.float64)
            assert tuple(t.shape) == (n, p) and t.dtype == torch.float64, (layout, n, p)
            assert t.stride(0) == p and (p == 1 or t.stride(1) == 1), (layout, t.stride())
            assert np.array_equal(t.numpy(), ref), (layout, n, p)
    # engine.rows_to_device: column-major arrays and single-column views
    a = np.asfortranarray(rng.random((100, 7)))
    for arr in (a, a[:, 2:3], np.ascontiguousarray(a)[:, 4:5], a[:0]):
        t = engine.rows_to_device(arr, "cpu")
        assert np.array_equal(t.numpy(), arr) and (t.numel() == 0 or (t.stride(0) == arr.shape[1] and (arr.shape[1] == 1 or t.stride(1) == 1)))
