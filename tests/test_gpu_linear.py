"""GPU: the linear-regression map step (SURVEY N3 / BASELINE config 5) without copies of the shard and as a STREAMING step for
shards that do not fit HBM: dlsa_xtv_stats_* (X'y, column sums, y'y, sum y in one read, fp64 sums, accumulating),
dlsa_gram_f32_acc64 (fp32 MFMA passes, fp64 slab sums), dlsa_synth_response_* (y = X beta* + sigma N(0,1) per row),
fit_linear_partitions (strided partitions, implicit intercept) and fit_linear_streaming.  The reference ships no linear map
(README.md:6 claims the method): the oracle is numpy's fp64 lstsq on the same rows -- the WLS combine of OLS blocks is
the global OLS estimate exactly."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


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


@pytest.mark.parametrize("dtype,n,p", [("f64", 5000, 37), ("f64", 1, 8), ("f64", 33333, 500), ("f64", 4097, 1025),
                                      ("f32", 5000, 37), ("f32", 77777, 2000), ("f32", 3, 256), ("f32", 20001, 999)])
def test_xtv_stats_matches_numpy_and_accumulates(eng, dtype, n, p):
    rng = np.random.default_rng(n + p)
    npdt, tdt = (np.float64, torch.float64) if dtype == "f64" else (np.float32, torch.float32)
    X = rng.standard_normal((n, p)).astype(npdt)
    v = rng.standard_normal(n).astype(npdt)
    X64, v64 = X.astype(np.float64), v.astype(np.float64)
    g, cs, st = eng.xtv_stats(dev(X), dev(v), want_colsum=True)
    tol = 1e-12
    assert rel_inf(g.cpu().numpy(), X64.T @ v64) < tol * max(1.0, np.sqrt(n))
    assert rel_inf(cs.cpu().numpy(), X64.sum(0)) < tol * max(1.0, np.sqrt(n))
    assert rel_inf(st.cpu().numpy(), [v64 @ v64, v64.sum()]) < 1e-11
    # a second chunk added on top (ragged cut, a view with a row pitch), without the column sums
    cut = n // 3
    g2, _, st2 = eng.xtv_stats(dev(X)[:cut], dev(v)[:cut].contiguous())
    eng.xtv_stats(dev(X)[cut:], dev(v)[cut:].contiguous(), g=g2, stats=st2, accumulate=True)
    assert rel_inf(g2.cpu().numpy(), X64.T @ v64) < tol * max(1.0, np.sqrt(n))
    assert rel_inf(st2.cpu().numpy(), [v64 @ v64, v64.sum()]) < 1e-11
    # same call twice: same bits (fixed summation order)
    g3, cs3, st3 = eng.xtv_stats(dev(X), dev(v), want_colsum=True)
    assert torch.equal(g, g3) and torch.equal(cs, cs3) and torch.equal(st, st3)


@pytest.mark.parametrize("n,p", [(40000, 1024), (30011, 2000), (9000, 300), (5000, 64)])
def test_gram_f32_acc64_sums_fp32_passes_in_fp64(eng, n, p):
    gen = torch.Generator(device="cuda"); gen.manual_seed(p)
    X = torch.randn((n, p), dtype=torch.float32, device="cuda", generator=gen)
    ref = X.double().T @ X.double()
    H = eng.gram_acc64(X)
    assert H.dtype == torch.float64 and torch.equal(H, H.T)
    d = ref.diagonal().sqrt()
    assert float(((H - ref).abs() / (d[:, None] * d[None, :])).max()) < 2e-6           # fp32 products / sums inside a slab
    # chunks accumulate in fp64: two halves added = one pass to fp32-slab accuracy, and into a sub-block view of a bigger matrix
    big = torch.zeros((p + 1, p + 1), dtype=torch.float64, device="cuda")
    eng.gram_acc64(X[: n // 2], out=big[1:, 1:])
    eng.gram_acc64(X[n // 2:], out=big[1:, 1:], accumulate=True)
    assert float(((big[1:, 1:] - ref).abs() / (d[:, None] * d[None, :])).max()) < 2e-6
    assert float(big[0].abs().max()) == 0.0 and float(big[:, 0].abs().max()) == 0.0


@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_synth_response_matches_oracle(eng, orc, dtype):
    n, p, seed = 3000, 21, 77
    tdt = torch.float64 if dtype == "f64" else torch.float32
    X, _ = eng.synth(seed, 100, n, p, kind=eng.SYNTH_UNIFORM, labels=False, dtype=tdt)
    y = eng.synth_response(seed, 100, X, sigma=0.5)
    Xo, yo = orc.synth_linear(seed, 100, n, p, orc.SYNTH_UNIFORM, sigma=0.5)
    assert rel_inf(y.cpu().numpy(), yo) < (1e-12 if dtype == "f64" else 1e-6)
    z = orc.synth_response_normals(seed, 0, 200000)
    assert abs(z.mean()) < 0.01 and abs(z.std() - 1.0) < 0.01                      # the noise is standard normal


@pytest.mark.parametrize("icpt", [False, True])
def test_linear_partitions_no_copy_strided_and_implicit_intercept(eng, orc, icpt):
    import dlsa_amd
    n, p, K = 30000, 33, 4
    X, y = orc.synth_linear(5, 0, n, p, orc.SYNTH_UNIFORM)
    y = y + (0.7 if icpt else 0.0)
    Xd, yd = dev(X), dev(y)
    torch.cuda.reset_peak_memory_stats()
    base = torch.cuda.memory_allocated()
    mb = dlsa_amd.fit_linear_partitions(Xd, yd, partition_num=K, fit_intercept=icpt)
    assert torch.cuda.max_memory_allocated() - base < 0.5 * X.nbytes + 64e6     # no gathered copy, no [1 | X] copy
    assert mb.status == [0] * K
    A = np.hstack([np.ones((n, 1)), X]) if icpt else X
    parts = orc.partition_rows(n, K)
    for k in range(K):
        Ak, yk = A[parts[k]], y[parts[k]]
        assert rel_inf(mb.Sig_inv[k].cpu().numpy(), Ak.T @ Ak) < 1e-12
        assert rel_inf(mb.Sig_invMcoef[k].cpu().numpy(), Ak.T @ yk) < 1e-11
        ck = np.linalg.lstsq(Ak, yk, rcond=None)[0]
        assert rel_inf(mb.coef[k].cpu().numpy(), ck) < 1e-9
        assert abs(mb.loglik[k] - float(np.sum((yk - Ak @ ck) ** 2))) < 1e-8 * float(yk @ yk)
    out = dlsa_amd.dlsa_mapred(mb)
    assert rel_inf(out["beta_byOLS"].to_numpy(), np.linalg.lstsq(A, y, rcond=None)[0]) < 1e-9


@pytest.mark.parametrize("dtype,icpt", [("f64", True), ("f64", False), ("f32", True)])
def test_linear_streaming_equals_resident_fit_and_global_ols(eng, orc, dtype, icpt):
    """The streaming map step (rows generated on the device chunk by chunk, never resident together) gives the blocks of
    the resident fit on the same rows, and their WLS combine is the global OLS estimate."""
    import dlsa_amd
    n, p, K, seed = 50000, 40, 3, 20260105
    tdt = torch.float64 if dtype == "f64" else torch.float32
    chunks = []
    mb = dlsa_amd.fit_linear_streaming(n, p, partition_num=K, chunk_rows=7000, seed=seed, kind="uniform", fit_intercept=icpt,
                                       dtype=tdt, on_chunk=lambda k, r, m: chunks.append((k, r, m)), overlap=(dtype == "f64"))
    assert mb.status == [0] * K and sum(m for _, _, m in chunks) == n and max(m for _, _, m in chunks) <= 7000
    assert len(chunks) == sum(-(-(int(n * (k + 1) / K) - int(n * k / K)) // 7000) for k in range(K))
    X, y = orc.synth_linear(seed, 0, n, p, orc.SYNTH_UNIFORM)
    if dtype == "f32":
        X, y = X.astype(np.float32).astype(np.float64), y.astype(np.float32).astype(np.float64)
    A = np.hstack([np.ones((n, 1)), X]) if icpt else X
    tol = 1e-11 if dtype == "f64" else 2e-6
    for k in range(K):
        lo, hi = int(n * k / K), int(n * (k + 1) / K)
        assert rel_inf(mb.Sig_inv[k].cpu().numpy(), A[lo:hi].T @ A[lo:hi]) < tol
        assert rel_inf(mb.Sig_invMcoef[k].cpu().numpy(), A[lo:hi].T @ y[lo:hi]) < tol * 10
    out = dlsa_amd.dlsa_mapred(mb)
    ols = np.linalg.lstsq(A, y, rcond=None)[0]
    assert rel_inf(out["beta_byOLS"].to_numpy(), ols) < (1e-9 if dtype == "f64" else 2e-5)
    truth = np.concatenate([[0.0] if icpt else [], orc.true_beta(p)])
    assert float(np.max(np.abs(out["beta_byOLS"].to_numpy() - truth))) < 0.1


def test_linear_streaming_config5_width_bounded_memory(eng):
    """Config 5's width (p = 2000 fp32) streamed in 2^18-row chunks: 1.5e6 rows (12 GB of rows) pass through a 2 GB chunk
    buffer; the peak stays under 4 GB and the estimate recovers the generating coefficients."""
    import dlsa_amd
    from conftest import _free_device_cache
    _free_device_cache()
    n, p = 1_500_000, 2000
    torch.cuda.reset_peak_memory_stats()
    base = torch.cuda.memory_allocated()
    mb = dlsa_amd.fit_linear_streaming(n, p, partition_num=2, chunk_rows=1 << 18, fit_intercept=True)
    peak = torch.cuda.max_memory_allocated() - base
    assert mb.status == [0, 0]
    assert peak < 4e9, "peak %.2f GB" % (peak / 1e9)
    assert eng.gram_last_kernel()[0].startswith("gram_wide_f32_kernel")
    out = dlsa_amd.dlsa_mapred(mb)
    truth = np.concatenate([[0.0], np.ones(800), np.zeros(1200)])
    assert float(np.max(np.abs(out["beta_byOLS"].to_numpy() - truth))) < 0.03


@pytest.mark.parametrize("n,p,ones_col,row0", [(3000, 40, False, 0), (1111, 2000, False, 12345678901), (700, 37, False, 5), (900, 22, True, 64)])
def test_synth_linear32_matches_oracle_restatement(eng, orc, n, p, ones_col, row0):
    """the fp32-native linear stream (dlsa_synth_linear_f32): same Philox words as the oracle's numpy restatement, the transcendental
    steps by the device's fp32 instructions -- agreement to a few fp32 ulps of the value scale, not bit for bit"""
    seed = 424242
    X, y = eng.synth_linear32(seed, row0, n, p, sigma=0.7, ones_col=ones_col)
    Xo, yo = orc.synth_linear32(seed, row0, n, p, sigma=0.7)
    Xg = X.cpu().numpy()
    if ones_col:
        assert np.all(Xg[:, 0] == 1.0)
        Xg = Xg[:, 1:]
    assert Xg.shape == Xo.shape and np.max(np.abs(Xg - Xo)) < 3e-6
    # the response: the device sums its own fp32 features in fp32, lane by lane
    yref = Xg.astype(np.float64) @ orc.true_beta(p) + (yo.astype(np.float64) - Xo.astype(np.float64) @ orc.true_beta(p))
    assert np.max(np.abs(y.cpu().numpy() - yref)) < 2e-5 * max(1.0, np.sqrt(p))
    # a row is a function of (seed, row index): two half calls give the same bits as one call
    h = n // 2
    Xa, ya = eng.synth_linear32(seed, row0, h, p, sigma=0.7, ones_col=ones_col)
    Xb, yb = eng.synth_linear32(seed, row0 + h, n - h, p, sigma=0.7, ones_col=ones_col)
    assert torch.equal(torch.cat([Xa, Xb]), X) and torch.equal(torch.cat([ya, yb]), y)


def test_synth_linear32_moments_and_explicit_beta(eng):
    n, p = 400000, 16
    beta = torch.linspace(-1.0, 1.0, p, dtype=torch.float32, device="cuda")
    X, y = eng.synth_linear32(9, 0, n, p, sigma=2.0, beta_true=beta)
    Xd = X.double()
    assert float(Xd.mean().abs()) < 2e-3 and abs(float(Xd.var()) - 1.0 / 12.0) < 2e-3
    C = (Xd.T @ Xd / n - torch.eye(p, dtype=torch.float64, device="cuda") / 12.0).abs().max()
    assert float(C) < 2e-3                                                           # columns (and the two Box-Muller outputs) uncorrelated
    resid = y.double() - Xd @ beta.double()
    assert abs(float(resid.mean())) < 0.02 and abs(float(resid.std()) - 2.0) < 0.02


def test_linear_streaming_config5_stated_size(eng):
    """BASELINE config 5 at its STATED per-GPU size: 6.25e7 x 2000 fp32 = 500 GB of rows generated on the device (fp32-native stream)
    and streamed through one chunk buffer.  Properties: bounded memory, the generating coefficients recovered, and the same
    stream summed with a different chunk size gives the same blocks to fp64 round-off of the fp32 partial Grams."""
    import dlsa_amd
    from conftest import _free_device_cache
    _free_device_cache()
    n, p, K = 62_500_000, 2000, 8
    torch.cuda.reset_peak_memory_stats()
    base = torch.cuda.memory_allocated()
    mb = dlsa_amd.fit_linear_streaming(n, p, partition_num=K, chunk_rows=1 << 22, fit_intercept=True, kind="gaussian32")
    peak = torch.cuda.max_memory_allocated() - base
    assert mb.status == [0] * K
    assert peak < 40e9, "peak %.2f GB" % (peak / 1e9)
    assert eng.gram_last_kernel()[0].startswith("gram_wide_f32_kernel")
    out = dlsa_amd.dlsa_mapred(mb)
    truth = np.concatenate([[0.0], np.ones(800), np.zeros(1200)])
    assert float(np.max(np.abs(out["beta_byOLS"].to_numpy() - truth))) < 0.004
    # one partition of the same stream (rows 0 .. n/K) with another chunking: same rows, other fp32 slab partials
    n1 = n // K
    a = dlsa_amd.fit_linear_streaming(n1, p, partition_num=1, chunk_rows=1 << 22, fit_intercept=True, kind="gaussian32")
    b = dlsa_amd.fit_linear_streaming(n1, p, partition_num=1, chunk_rows=3_000_000, fit_intercept=True, kind="gaussian32")
    Sa, Sb = a.Sig_inv[0].cpu().numpy(), b.Sig_inv[0].cpu().numpy()
    assert rel_inf(Sa, Sb) < 5e-6 and rel_inf(Sa, mb.Sig_inv[0].cpu().numpy()) < 5e-6
    assert rel_inf(a.coef[0].cpu().numpy(), b.coef[0].cpu().numpy()) < 1e-3         # coefficients of size <= 1 with sd ~ 1.2e-3 each


def test_linear_edge_cases_empty_rows_and_partitions(eng):
    import dlsa_amd
    X = torch.randn((0, 12), dtype=torch.float64, device="cuda")
    v = torch.randn(0, dtype=torch.float64, device="cuda")
    g, cs, st = eng.xtv_stats(X, v, want_colsum=True)
    assert float(g.abs().max()) == 0.0 and float(cs.abs().max()) == 0.0 and float(st.abs().max()) == 0.0
    g[:] = 3.0; st[:] = 5.0
    eng.xtv_stats(X, v, g=g, stats=st, accumulate=True)                 # adding nothing changes nothing
    assert float((g - 3.0).abs().max()) == 0.0 and float((st - 5.0).abs().max()) == 0.0
    # fewer rows than partitions: the empty partitions report status 4 (the reference's zero block), the others fit
    mb = dlsa_amd.fit_linear_streaming(40, 3, partition_num=64, chunk_rows=16, kind="uniform", dtype=torch.float64)
    assert mb.status.count(4) == 24 and all(s in (0, 2, 4) for s in mb.status)
    mb2 = dlsa_amd.fit_linear_partitions(torch.randn((30, 3), dtype=torch.float64, device="cuda"),
                                         torch.randn(30, dtype=torch.float64, device="cuda"), part_offsets=[0, 0, 30])
    assert mb2.status == [4, 0] and float(mb2.Sig_inv[0].abs().max()) == 0.0


@pytest.mark.parametrize("dtype,icpt,pinned", [("f64", True, True), ("f64", False, False), ("f32", True, True)])
def test_linear_chunks_from_host_equal_resident_fit_and_global_ols(eng, orc, dtype, icpt, pinned):
    """Rows that live in HOST memory, handed over as ragged chunks in a shuffled partition order (numpy arrays or pinned
    tensors; the copy of chunk i + 1 overlaps the kernels of chunk i): the blocks of the resident fit, the global OLS."""
    import dlsa_amd
    n, p, K, seed = 41000, 36, 3, 20260107
    X, y = orc.synth_linear(seed, 0, n, p, orc.SYNTH_UNIFORM)
    y = y + (0.3 if icpt else 0.0)
    ndt = np.float64 if dtype == "f64" else np.float32
    Xh, yh = X.astype(ndt), y.astype(ndt)
    rng = np.random.default_rng(3)
    cuts = np.sort(rng.choice(np.arange(1, n), size=11, replace=False))
    pieces = [(int(a), int(b)) for a, b in zip(np.r_[0, cuts], np.r_[cuts, n])]
    pieces = [pieces[i] for i in rng.permutation(len(pieces))]
    part_of = lambda a: (a * K) // n                                 # partition of a chunk = of its first row (any rule works)

    def produce():
        for a, b in pieces:
            if pinned:
                yield part_of(a), torch.from_numpy(Xh[a:b].copy()).pin_memory(), torch.from_numpy(yh[a:b].copy()).pin_memory()
            else:
                yield part_of(a), Xh[a:b], yh[a:b]
        yield 0, np.zeros((0, p), ndt), np.zeros((0,), ndt)          # an empty chunk is skipped

    mb = dlsa_amd.fit_linear_chunks(produce(), p, partition_num=K, fit_intercept=icpt)
    assert mb.status == [0] * K and mb.sample_size == n
    X64, y64 = Xh.astype(np.float64), yh.astype(np.float64)
    A = np.hstack([np.ones((n, 1)), X64]) if icpt else X64
    tol = 1e-11 if dtype == "f64" else 2e-6
    for k in range(K):
        rows = np.concatenate([np.arange(a, b) for a, b in pieces if part_of(a) == k])
        assert rel_inf(mb.Sig_inv[k].cpu().numpy(), A[rows].T @ A[rows]) < tol
        assert rel_inf(mb.Sig_invMcoef[k].cpu().numpy(), A[rows].T @ y64[rows]) < tol * 10
    out = dlsa_amd.dlsa_mapred(mb)
    assert rel_inf(out["beta_byOLS"].to_numpy(), np.linalg.lstsq(A, y64, rcond=None)[0]) < (1e-9 if dtype == "f64" else 2e-5)


def test_linear_chunks_device_chunks_and_errors(eng, orc):
    import dlsa_amd
    n, p = 9000, 12
    X, y = orc.synth_linear(11, 0, n, p, orc.SYNTH_UNIFORM)
    Xd, yd = dev(X), dev(y)
    mb = dlsa_amd.fit_linear_chunks([(0, Xd[:5000], yd[:5000]), (1, Xd[5000:], yd[5000:]), (0, X[:0], y[:0])], p, partition_num=3)
    assert mb.status == [0, 0, 4]                                    # partition 2 saw no rows: the reference's zero block
    ref = dlsa_amd.fit_linear_partitions(Xd, yd, part_offsets=[0, 5000, n, n])
    assert torch.equal(mb.Sig_inv, ref.Sig_inv) and torch.equal(mb.Sig_invMcoef, ref.Sig_invMcoef)
    with pytest.raises(ValueError, match="partition index"):
        dlsa_amd.fit_linear_chunks([(3, X, y)], p, partition_num=3)
    with pytest.raises(ValueError, match="need \\[m, 12\\]"):
        dlsa_amd.fit_linear_chunks([(0, X[:, :5], y)], p, partition_num=1)
