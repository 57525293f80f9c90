"""GPU: the host layer's argument guards, the shipped library's immunity to the timing knobs, the one-shot divisor,
and bench.py's own multi-rank launch (two gloo ranks sharing the test box's one GPU)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rel_inf(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return float(np.max(np.abs(a - b)) / max(1e-300, np.max(np.abs(b))))


@pytest.fixture(scope="module")
def eng():
    assert torch.cuda.is_available()
    from dlsa_amd import engine
    return engine


@pytest.fixture(scope="module")
def orc():
    from oracle import dlsa_oracle
    return dlsa_oracle


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def test_fp64_entry_points_refuse_other_dtypes_and_strides(eng, orc):
    """The C ABI reads raw pointers: an fp32 X or integer labels must raise, not be reinterpreted."""
    X, y = orc.synth_logistic(21, 0, 2000, 6)
    Xd, yd = dev(X), dev(y)
    beta = torch.zeros(6, dtype=torch.float64, device="cuda")
    with pytest.raises(TypeError):
        eng.irls_fit(Xd.float(), yd, [0, 2000])
    with pytest.raises(TypeError):
        eng.irls_fit(Xd, yd.long(), [0, 2000])
    with pytest.raises(TypeError):
        eng.logit_pass(Xd, yd.float(), beta)
    with pytest.raises(TypeError):
        eng.logit_pass(Xd, yd, beta.float())
    with pytest.raises(ValueError):
        eng.logit_pass(Xd, torch.stack([yd, yd], 1)[:, 0], beta)          # strided y
    with pytest.raises(ValueError):
        eng.logit_pass(Xd.t().contiguous().t(), yd, beta)                 # column-major X
    with pytest.raises(TypeError):
        eng.loglik(Xd, yd, torch.zeros((6, 2), dtype=torch.float32, device="cuda"))
    S = torch.eye(6, dtype=torch.float64, device="cuda")
    with pytest.raises(TypeError):
        eng.spd_solve(S.float(), beta)
    with pytest.raises(TypeError):
        eng.lars_path(S, beta.float(), False, 100)
    with pytest.raises(TypeError):
        eng.gram(Xd.to(torch.float16))
    with pytest.raises(TypeError):
        eng.sum_blocks(torch.zeros((1, 6), device="cuda"), torch.zeros((1, 6), device="cuda"),
                       torch.zeros((1, 6, 6), device="cuda"))


def test_integer_and_fp32_labels_are_cast_by_the_fit(orc):
    import dlsa_amd
    X, y = orc.synth_logistic(22, 0, 6000, 8)
    Xd = dev(X)
    ref = dlsa_amd.fit_logistic_partitions(Xd, dev(y), partition_num=2)
    for lab in (dev(y).long(), dev(y).float(), dev(y).bool()):
        got = dlsa_amd.fit_logistic_partitions(Xd, lab, partition_num=2)
        assert got.status == [0, 0]
        assert torch.equal(got.coef, ref.coef) and torch.equal(got.Sig_inv, ref.Sig_inv)
    with pytest.raises(TypeError):
        dlsa_amd.fit_logistic_partitions(Xd.float(), dev(y), partition_num=2)


def test_lars_max_steps_zero_means_default(eng, orc):
    """lsa.py:93-94 default max_steps = 8 m; the C ABI reads <= 0 as that default and the buffers match it."""
    rng = np.random.default_rng(5)
    A = rng.standard_normal((60, 9))
    S, b = A.T @ A, rng.standard_normal(9)
    full = eng.lars_path(dev(S), dev(b), False, 60)
    for ms in (0, -3, None):
        r = eng.lars_path(dev(S), dev(b), False, 60, max_steps=ms)
        assert r["beta"].shape == full["beta"].shape and torch.equal(r["beta"], full["beta"])
    two = eng.lars_path(dev(S), dev(b), False, 60, max_steps=2)
    assert two["beta"].shape[0] == 3 and torch.equal(two["beta"], full["beta"][:3])


def test_timing_knobs_cannot_change_results_of_the_shipped_library(eng, orc, monkeypatch):
    """DLSA_GRAM_DBG bits 1 / 16 / 128 and DLSA_OH_DBG gave wrong results for timing experiments; they exist only in
    -DDLSA_DEBUG_KNOBS builds (`make knobs`), so the environment must not be able to change a result here."""
    outs = []
    X500, _ = orc.synth_logistic(23, 0, 6000, 500, orc.SYNTH_GAUSSIAN)
    X100, _ = orc.synth_logistic(24, 0, 9000, 100, orc.SYNTH_GAUSSIAN)
    w500, w100 = np.random.default_rng(1).uniform(0.05, 0.25, 6000), np.random.default_rng(2).uniform(0.05, 0.25, 9000)
    Xw = torch.randn((4000, 1024), dtype=torch.float32, device="cuda")
    for val in (None, "1", "16", "128", "145"):
        if val is None:
            monkeypatch.delenv("DLSA_GRAM_DBG", raising=False)
            monkeypatch.delenv("DLSA_OH_DBG", raising=False)
        else:
            monkeypatch.setenv("DLSA_GRAM_DBG", val)
            monkeypatch.setenv("DLSA_OH_DBG", "3")
        outs.append((eng.gram(dev(X500), dev(w500)).cpu(), eng.gram(dev(X100), dev(w100)).cpu(), eng.gram(Xw).cpu()))
    # ... nor can the options struct that replaced the environment for the valid-result switches: the wrong-result bits are masked out
    with eng.kernel_options(gram_variant=1 | 16 | 128):
        outs.append((eng.gram(dev(X500), dev(w500)).cpu(), eng.gram(dev(X100), dev(w100)).cpu(), eng.gram(Xw).cpu()))
    for o in outs[1:]:
        for a, b in zip(o, outs[0]):
            assert torch.equal(a, b)
    assert rel_inf(outs[0][0].numpy(), orc.gram(X500, w500)) < 1e-12
    assert rel_inf(outs[0][1].numpy(), orc.gram(X100, w100)) < 1e-12


def test_onehot_knob_cannot_change_results(eng, orc, monkeypatch):
    from test_gpu_onehot import _plan, _random_design
    rng = np.random.default_rng(77)
    p, num, codes, desc, nl, level_col = _random_design(rng, 4000, 3, (12, 5))
    plan = _plan(None, p, desc, nl, level_col)
    X, _ = orc.design_matrix(num, codes, *desc)
    w = rng.uniform(0.05, 0.25, 4000)
    Ho = orc.gram(X, w)
    for val in (None, "1", "2", "3"):
        if val is None:
            monkeypatch.delenv("DLSA_OH_DBG", raising=False)
        else:
            monkeypatch.setenv("DLSA_OH_DBG", val)
        H = eng.onehot_gram(plan, dev(num), dev(codes), dev(w)).cpu().numpy()
        assert rel_inf(H, Ho) < 1e-12, val


def test_oneshot_divisor_is_global_when_given(orc):
    """dlsa.py:51-52 divides by the partition count of the whole job: an explicit num_partitions is used as is."""
    import dlsa_amd
    X, y = orc.synth_logistic(25, 0, 8000, 5)
    mb = dlsa_amd.fit_logistic_partitions(dev(X), dev(y), partition_num=4)
    a = dlsa_amd.dlsa_mapred(mb)
    b = dlsa_amd.dlsa_mapred(mb, num_partitions=8)
    assert np.allclose(a["beta_byONESHOT"].to_numpy(), mb.coef.mean(0).cpu().numpy(), rtol=1e-14, atol=0)
    assert np.allclose(b["beta_byONESHOT"].to_numpy() * 2.0, a["beta_byONESHOT"].to_numpy(), rtol=1e-14, atol=0)
    assert np.array_equal(a["beta_byOLS"].to_numpy(), b["beta_byOLS"].to_numpy())


def test_two_streams_get_their_own_workspace(eng, orc):
    X, _ = orc.synth_logistic(26, 0, 50000, 64, orc.SYNTH_GAUSSIAN)
    Xd = dev(X)
    ref = eng.gram(Xd).clone()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    outs = []
    for _ in range(4):
        with torch.cuda.stream(s1):
            outs.append(eng.gram(Xd))
        with torch.cuda.stream(s2):
            outs.append(eng.gram(Xd))
    torch.cuda.synchronize()
    assert len({k[2] for k in eng._ws_cache}) >= 3          # default stream + the two side streams
    for o in outs:
        assert torch.equal(o, ref)


def _run_bench(extra_env, *argv):
    env = dict(os.environ)
    env.update(extra_env)
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    # (--watchdog-seconds: a rank that hangs prints its Python stacks and exits; the message lands in the assertion below.  Round 6 saw
    # ONE run of the two-rank line in ~120 stop making progress inside a pytest parent -- not reproduced since, lab notes r06 section 5 --
    # so a run that ended on the watchdog or the timeout is repeated once, with a warning that carries the first attempt's output)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--watchdog-seconds", "150"] + list(argv)
    for attempt in (0, 1):
        try:
            pr = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=400)
            hung = pr.returncode != 0 and "Timeout (" in pr.stderr
            err = pr.stderr
        except subprocess.TimeoutExpired as e:
            hung, err, pr = True, str(e.stderr)[-6000:], None
        if not hung or attempt == 1:
            break
        import warnings
        warnings.warn("bench.py %s stopped making progress and was repeated; first attempt's stderr tail:\n%s" % (" ".join(argv), err[-3000:]))
    assert pr is not None and pr.returncode == 0, err[-6000:]
    lines = [ln for ln in pr.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, pr.stdout
    return json.loads(lines[0])


def test_bench_gpus_2_launches_two_ranks_itself():
    """`python bench.py --gpus 2` with no torch.distributed environment must start the two ranks itself (gloo here:
    two RCCL ranks cannot share the test box's single GPU) and report n_gpus = 2 with the reduce timed."""
    out = _run_bench({"DLSA_BENCH_BACKEND": "gloo"}, "--gpus", "2", "--rows-per-gpu", "2000000", "--steps", "3",
                     "--warmup", "1")
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["warmup"] == 1
    assert out["rccl_ranks"] == 0 and out["allreduce"]["backend"] == "gloo" and out["allreduce"]["ranks"] == 2
    assert out["allreduce"]["payload_bytes"] == (500 * 500 + 2 * 500) * 8
    assert out["allreduce"]["ms_isolated"] > 0 and out["allreduce"]["ms_in_step"] > 0
    assert out["config"]["rows_per_gpu"] == 2000000 and out["scaling"] == "weak"
    assert abs(out["value"] - 2 * 2000000 * 3 / (out["ms_per_step"] * 3e-3)) / out["value"] < 1e-6
    assert out["cpu_baseline"] is None


def test_bench_single_gpu_line_keeps_its_contract():
    out = _run_bench({}, "--rows-per-gpu", "2000000", "--steps", "3", "--warmup", "1", "--no-cpu-baseline")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in out
    assert out["n_gpus"] == 1 and out["allreduce"] is None and out["rccl_ranks"] == 0
    r = out["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert r["traffic"] is None or r["traffic"] > r["algorithmic_bytes_per_launch"] * 0.5
    # round 6: the slow-start guards and the fit legs are part of the default line
    assert out["warmup_extra_steps"] >= 1 and out["first_step_ms"] > 0 and out["median_step_ms"] > 0
    assert abs(out["first_over_median"] - out["first_step_ms"] / out["median_step_ms"]) < 1e-12 and out["first_over_median"] < 1.5
    e2e = out["extra"]["end_to_end_fit"]
    assert e2e["status_ok"] and e2e["partitions_per_rank"] == 25 and e2e["fit_path"] == 0 and e2e["map_s"] > 0
    assert e2e["k1"]["status_ok"] and e2e["k1"]["partitions_per_rank"] == 1 and e2e["k1"]["passes_per_partition"]["max"] >= 1
    ls = e2e["lock_step"]
    assert ls["status_ok"] and ls["fit_path"] == 2 and ls["partitions"] == 1000 and 0 < ls["fit_s"] < 1.0


def test_bench_refuses_world_size_mismatch():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    pr = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"], env=env, stdout=subprocess.PIPE,
                        stderr=subprocess.PIPE, text=True, timeout=300)
    assert pr.returncode != 0 and "WORLD_SIZE" in pr.stderr


def test_workspace_cache_is_bounded_per_stream(eng):
    """the per-stream scratch cache was grow-only with no eviction.  Six short-lived streams touching the Gram must
    leave at most engine._WS_MAX_STREAMS buffers behind, results staying right on every stream."""
    X = torch.randn((20000, 64), dtype=torch.float64, device="cuda")
    ref = X.T @ X
    eng.release_workspace()
    for _ in range(6):
        st = torch.cuda.Stream()
        st.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(st):
            H = eng.gram(X)
        st.synchronize()
        assert float((H - ref).abs().max()) < 1e-11 * float(ref.abs().max())
        assert len(eng._ws_cache) <= eng._WS_MAX_STREAMS
    eng.release_workspaces()
    assert len(eng._ws_cache) == 0
