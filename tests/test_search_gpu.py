"""GPU parity: the HIP flat-search path (through the C-ABI) against the CPU oracle.

Bar: bit-exact ids AND bit-exact scores (integer/index work + a fixed fp32 summation order).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _unit(rng, n, d):
    x = rng.standard_normal((n, d)).astype(np.float32)
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    return x


def _check(oracle, X, Q, k, metric, dtype, mode=None, id_base=0):
    from rag_amd.index import FlatIndex, MODE_EXACT
    idx = FlatIndex(X.shape[1], metric=metric, dtype=dtype)
    if id_base:
        idx.set_id_base(id_base)
    if mode is not None:
        idx.set_mode(mode)
    if len(X):
        idx.add(X)
    D, I = idx.search(Q, k)
    m = {"ip": oracle.METRIC_IP, "l2": oracle.METRIC_L2}[metric]
    t = {"f32": oracle.DTYPE_F32, "bf16": oracle.DTYPE_BF16}[dtype]
    Do, Io = oracle.search(X.reshape(-1, Q.shape[1]), Q, k, metric=m, dtype=t, id_base=id_base)
    np.testing.assert_array_equal(I, Io)
    np.testing.assert_array_equal(D.view(np.uint32), Do.view(np.uint32))
    return idx


@pytest.mark.parametrize("metric", ["ip", "l2"])
@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("n,d,nq,k", [(1000, 384, 7, 5), (5000, 64, 33, 10), (300, 100, 3, 4), (70000, 384, 20, 10),
                                       # > 64 queries: one select workgroup per query over the whole segment
                                       # (rank words in registers up to 32768 rows, recomputed per pass beyond)
                                       (6000, 64, 70, 10), (40000, 64, 70, 10)])
def test_exact_path_matches_oracle(oracle_mod, metric, dtype, n, d, nq, k):
    from rag_amd.index import MODE_EXACT
    rng = np.random.default_rng(n + d)
    _check(oracle_mod, _unit(rng, n, d), _unit(rng, nq, d), k, metric, dtype, mode=MODE_EXACT)


@pytest.mark.parametrize("metric,dtype", [("ip", "f32"), ("l2", "bf16")])
def test_exact_path_on_the_fp32_matrix_pipe(oracle_mod, metric, dtype):
    """More than 16 queries: scan_dense runs on v_mfma_f32_32x32x2_f32 (search_exact.hip), whose accumulation is the
    k-ordered fmaf chain itself.  300 queries (five 64-query tiles, the last partial) over 70,001 rows (a partial
    row tile), rows with NaN, +-0, denormal-scale and huge values, duplicates: bit-exact ids and score bits."""
    from rag_amd.index import MODE_EXACT
    rng = np.random.default_rng(77)
    n, d, nq, k = 70_001, 384, 300, 10
    X = _unit(rng, n, d)
    X[11] = 0.0
    X[12] = -0.0
    X[13] *= 1e-38                                   # products and sums in the denormal range
    X[14] *= 1e18
    X[15, 7] = np.nan
    X[5000:5040] = X[3]
    Q = _unit(rng, nq, d)
    Q[0] = X[3]
    Q[1] *= 1e-7
    Q[2] = -X[14] * 1e-18
    _check(oracle_mod, X, Q, k, metric, dtype, mode=MODE_EXACT)
    # 17 .. 32 queries take the one-block variant
    _check(oracle_mod, X[:9000], Q[:29], 33, metric, dtype, mode=MODE_EXACT)


_VALU_CHILD = r"""
import sys, numpy as np
sys.path.insert(0, %r)
import oracle
from rag_amd.index import FlatIndex, MODE_EXACT
rng = np.random.default_rng(6)
X = rng.standard_normal((30000, 200)).astype(np.float32); Q = rng.standard_normal((70, 200)).astype(np.float32)
for dtype, odt in (("f32", oracle.DTYPE_F32), ("bf16", oracle.DTYPE_BF16)):
    idx = FlatIndex(200, metric="l2", dtype=dtype); idx.set_mode(MODE_EXACT); idx.add(X)
    D, I = idx.search(Q, 10)
    Do, Io = oracle.search(X, Q, 10, metric=oracle.METRIC_L2, dtype=odt)
    assert np.array_equal(I, Io) and np.array_equal(D.view(np.uint32), Do.view(np.uint32))
print("OK")
"""


def test_valu_scan_stays_covered_in_a_child_process():
    """MI_EXACT_MFMA=0 keeps the fp32 VALU scan for more than 16 queries (the kernel the MFMA one replaced; still the
    path for 5 .. 16 queries): read once per process, so a child process."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = dict(os.environ)
    e["MI_EXACT_MFMA"] = "0"
    out = subprocess.run([sys.executable, "-c", _VALU_CHILD % root], env=e, capture_output=True, text=True, timeout=170)
    assert out.returncode == 0 and "OK" in out.stdout, (out.stdout[-500:], out.stderr[-2000:])


@pytest.mark.parametrize("metric", ["ip", "l2"])
@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("n,d,nq,k", [(2000, 384, 1, 5), (20000, 384, 1, 5), (50000, 100, 3, 100), (200000, 384, 2, 5),
                                       (9000, 384, 4, 1000), (4097, 64, 1, 7)])
def test_few_queries_take_the_latency_kernels(oracle_mod, metric, dtype, n, d, nq, k):
    """nq <= 4: one-wave dense scan, select in pieces of >= 4096 rows + one merge (search_exact.hip); AUTO mode
    routes such searches to the exact path up to 2^20 rows."""
    rng = np.random.default_rng(n + d + nq)
    X = _unit(rng, n, d)
    X[n // 2] = X[3]                                 # a tie across pieces
    idx = _check(oracle_mod, X, np.concatenate([X[3:4], _unit(rng, nq - 1, d)]) if nq > 1 else X[3:4].copy(), k, metric, dtype)
    assert idx.last_scan_time()["kind"] == "exact_f32"


def test_k_larger_than_n_and_empty(oracle_mod):
    rng = np.random.default_rng(1)
    X, Q = _unit(rng, 3, 32), _unit(rng, 2, 32)
    for metric in ("ip", "l2"):
        _check(oracle_mod, X, Q, 8, metric, "f32")
        _check(oracle_mod, X[:0], Q, 4, metric, "f32")


def test_ties_and_duplicates_lower_id_first(oracle_mod):
    rng = np.random.default_rng(2)
    X = _unit(rng, 2000, 128)
    X[100:140] = X[7]          # 41 identical rows
    X[1500] = X[7]
    Q = np.concatenate([X[7:8], _unit(rng, 4, 128)])
    for metric in ("ip", "l2"):
        for dtype in ("f32", "bf16"):
            _check(oracle_mod, X, Q, 50, metric, dtype)


def test_id_base_offsets_labels(oracle_mod):
    rng = np.random.default_rng(3)
    _check(oracle_mod, _unit(rng, 500, 64), _unit(rng, 5, 64), 6, "ip", "f32", id_base=10**10)


@pytest.mark.parametrize("n,d,nq,k", [(40000, 384, 300, 10), (100000, 128, 64, 5), (50000, 384, 5, 10),
                                       (66000, 768, 40, 10), (40000, 384, 50, 100), (40000, 384, 17, 20)])
def test_mfma_path_matches_oracle(oracle_mod, n, d, nq, k):
    from rag_amd.index import MODE_FAST
    rng = np.random.default_rng(n + d + k)
    idx = _check(oracle_mod, _unit(rng, n, d), _unit(rng, nq, d), k, "ip", "bf16", mode=MODE_FAST)
    st = idx.stats()
    assert st["fast"] + st["exact"] == nq
    assert st["fast"] >= nq * 0.9, st          # random data: the certificate should nearly always hold


@pytest.mark.parametrize("metric", ["ip", "l2"])
@pytest.mark.parametrize("dtype", ["bf16", "f32"])
@pytest.mark.parametrize("n,d,nq,k", [(40000, 384, 70, 10), (50000, 200, 9, 5), (45000, 768, 33, 100), (33000, 100, 3, 4)])
def test_mfma_path_all_metrics_and_dtypes(oracle_mod, metric, dtype, n, d, nq, k):
    """L2 runs on the MFMA path with the accumulators started at -xx/2; an F32 index filters on
    its bf16 shadow and re-scores from the fp32 rows.  Rows are NOT unit norm here."""
    from rag_amd.index import MODE_FAST
    rng = np.random.default_rng(n + d + k + len(metric))
    X = _unit(rng, n, d) * rng.uniform(0.5, 1.5, size=(n, 1)).astype(np.float32)
    Q = _unit(rng, nq, d) * rng.uniform(0.5, 2.0, size=(nq, 1)).astype(np.float32)
    idx = _check(oracle_mod, X, Q, k, metric, dtype, mode=MODE_FAST)
    st = idx.stats()
    assert st["fast"] + st["exact"] == nq
    # a bf16 index certifies nearly always; the rounded shadow of an F32 index widens the bound
    # (2^-8 |q||x|), which at this small n / few slices sends more queries to the exact path
    # measured in round 2: every case here certifies all of its queries, fp32 indexes included
    assert st["fast"] >= nq * (0.8 if dtype == "bf16" else 0.7), st
    print(f"certified {st['fast']}/{nq} ({metric}, {dtype}, n={n})")


@pytest.mark.parametrize("metric,dtype", [("ip", "bf16"), ("l2", "bf16"), ("l2", "f32")])
def test_mfma_many_slices_grouped_finish(oracle_mod, metric, dtype):
    """Few queries over many rows: hundreds of row slices, several finish groups + merge."""
    from rag_amd.index import FlatIndex, MODE_FAST
    n, d, nq, k = 1_500_000, 128, 5, 10
    idx = FlatIndex(d, metric=metric, dtype=dtype)
    idx.set_mode(MODE_FAST)
    idx.add_synthetic(n, seed=11)
    X = oracle_mod.synth_rows(11, 0, n, d)
    Q = oracle_mod.synth_rows(12, 0, nq, d)
    D, I = idx.search(Q, k)
    m = {"ip": oracle_mod.METRIC_IP, "l2": oracle_mod.METRIC_L2}[metric]
    t = {"f32": oracle_mod.DTYPE_F32, "bf16": oracle_mod.DTYPE_BF16}[dtype]
    Do, Io = oracle_mod.search(X, Q, k, metric=m, dtype=t)
    np.testing.assert_array_equal(I, Io)
    np.testing.assert_array_equal(D.view(np.uint32), Do.view(np.uint32))
    assert idx.stats()["fast"] == nq


def test_mfma_nan_row_and_add_after_search(oracle_mod):
    """A NaN row never enters a candidate list; rows added after the first fast search of an F32
    index must reach its bf16 shadow."""
    from rag_amd.index import FlatIndex, MODE_FAST
    rng = np.random.default_rng(13)
    X = _unit(rng, 40000, 384)
    X[123, 5] = np.nan
    Q = _unit(rng, 6, 384)
    _check(oracle_mod, X, Q, 10, "ip", "bf16", mode=MODE_FAST)
    idx = FlatIndex(384, metric="l2", dtype="f32")
    idx.set_mode(MODE_FAST)
    idx.add(X[:35000])
    idx.search(Q, 10)
    X2 = np.concatenate([X[:35000], Q * 0.999 + 0.001 * X[35000:35006], X[35006:]]).astype(np.float32)
    idx.add(X2[35000:])
    D, I = idx.search(Q, 10)
    Do, Io = oracle_mod.search(X2, Q, 10, metric=oracle_mod.METRIC_L2, dtype=oracle_mod.DTYPE_F32)
    np.testing.assert_array_equal(I, Io)
    np.testing.assert_array_equal(D.view(np.uint32), Do.view(np.uint32))
    assert set(I[:, 0]) == set(range(35000, 35006))


def test_mfma_certificate_falls_back_on_duplicates(oracle_mod):
    """More identical rows than the candidate lists hold: the certificate cannot hold for that
    query, so it must be re-run on the exact path and still be bit-exact."""
    from rag_amd.index import MODE_FAST
    rng = np.random.default_rng(5)
    X = _unit(rng, 40000, 384)
    X[1000:1400] = X[3]
    Q = np.concatenate([X[3:4], _unit(rng, 9, 384)])
    idx = _check(oracle_mod, X, Q, 10, "ip", "bf16", mode=MODE_FAST)
    assert idx.stats()["exact"] >= 1


def test_mfma_clustered_rows(oracle_mod):
    """Scores that rise along the scan (a topic cluster at the end of the corpus) exercise the
    in-kernel list compaction."""
    from rag_amd.index import MODE_FAST
    rng = np.random.default_rng(6)
    n, d = 60000, 384
    X = _unit(rng, n, d)
    Q = _unit(rng, 12, d)
    ramp = np.linspace(0.0, 0.6, n).astype(np.float32)[:, None]
    X = X + ramp * Q[0][None, :]
    X /= np.linalg.norm(X, axis=1, keepdims=True)
    _check(oracle_mod, X.astype(np.float32), Q, 10, "ip", "bf16", mode=MODE_FAST)


def test_synthetic_rows_match_oracle_generator(oracle_mod):
    from rag_amd.index import FlatIndex
    for dtype in ("f32", "bf16"):
        idx = FlatIndex(384, metric="ip", dtype=dtype)
        idx.set_id_base(1000)
        idx.add_synthetic(300, seed=7)
        got = idx.reconstruct_n(0, 300)
        want = oracle_mod.synth_rows(7, 1000, 300, 384)
        if dtype == "bf16":
            want = oracle_mod.bf16_to_f32(oracle_mod.round_bf16(want))
        np.testing.assert_array_equal(got.view(np.uint32), want.view(np.uint32))


def test_merge_matches_oracle(oracle_mod):
    import torch
    from rag_amd.index import topk_merge_torch
    rng = np.random.default_rng(8)
    nq, m, k = 37, 80, 10
    s = rng.standard_normal((nq, m)).astype(np.float32)
    s[:, 5] = s[:, 6]                                   # ties
    ids = rng.permutation(10**6)[: nq * m].reshape(nq, m).astype(np.int64)
    ids[:, 70:] = -1                                    # padding
    for metric, om in (("ip", oracle_mod.METRIC_IP), ("l2", oracle_mod.METRIC_L2)):
        ss = np.abs(s) if metric == "l2" else s
        D, I = topk_merge_torch(torch.from_numpy(ss).cuda(), torch.from_numpy(ids).cuda(), k, metric)
        Do, Io = oracle_mod.topk_merge(ss, ids, k, om)
        np.testing.assert_array_equal(I.cpu().numpy(), Io)
        np.testing.assert_array_equal(D.cpu().numpy().view(np.uint32), Do.view(np.uint32))


def test_concurrent_searches_from_many_threads(oracle_mod):
    """The reference enters the retriever from LangChain RunnableParallel workers and Gradio handler
    threads (advanced_rag.py:302-305, 545-555): ctypes releases the GIL, the library serialises per
    handle.  Eight threads hammer two indexes; every answer must equal the oracle's."""
    import threading
    from rag_amd.index import FlatIndex
    rng = np.random.default_rng(21)
    Xa, Xb = _unit(rng, 40000, 384), _unit(rng, 3000, 128)
    ia = FlatIndex(384, metric="ip", dtype="bf16"); ia.add(Xa)
    ib = FlatIndex(128, metric="l2", dtype="f32"); ib.add(Xb)
    Qa, Qb = _unit(rng, 16, 384), _unit(rng, 16, 128)
    Da, Ia = oracle_mod.search(Xa, Qa, 10, metric=oracle_mod.METRIC_IP, dtype=oracle_mod.DTYPE_BF16)
    Db, Ib = oracle_mod.search(Xb, Qb, 5, metric=oracle_mod.METRIC_L2, dtype=oracle_mod.DTYPE_F32)
    errors = []

    def work(t):
        try:
            for it in range(20):
                j = (t + it) % 16
                if (t + it) % 2:
                    D, I = ia.search(Qa[j:j + 1], 10)
                    ok = np.array_equal(I[0], Ia[j]) and np.array_equal(D[0].view(np.uint32), Da[j].view(np.uint32))
                else:
                    D, I = ib.search(Qb[j:j + 1], 5)
                    ok = np.array_equal(I[0], Ib[j]) and np.array_equal(D[0].view(np.uint32), Db[j].view(np.uint32))
                if not ok:
                    errors.append((t, it))
        except Exception as e:          # noqa: BLE001
            errors.append((t, repr(e)))

    th = [threading.Thread(target=work, args=(t,)) for t in range(8)]
    [x.start() for x in th]
    [x.join() for x in th]
    assert not errors, errors[:3]


def test_answer_is_invariant_under_row_permutation_and_batching(oracle_mod):
    """Properties at a size the oracle does not need to see: (1) permuting the corpus permutes the
    ids and leaves the scores bit-identical (no ties here); (2) a query's answer does not depend
    on which batch it travelled in (Q = 1 vs a 300-query batch take different scan variants)."""
    from rag_amd.index import FlatIndex
    n, d, k = 200_000, 384, 10
    idx = FlatIndex(d, metric="ip", dtype="bf16")
    idx.add_synthetic(n, seed=31)
    X = idx.reconstruct_n(0, n)
    Q = oracle_mod.synth_rows(32, 0, 300, d)
    D, I = idx.search(Q, k)
    D1, I1 = idx.search(Q[7:8], k)
    assert np.array_equal(I1[0], I[7]) and np.array_equal(D1[0].view(np.uint32), D[7].view(np.uint32))
    perm = np.random.default_rng(33).permutation(n)
    idx2 = FlatIndex(d, metric="ip", dtype="bf16")
    idx2.add(X[perm])
    D2, I2 = idx2.search(Q, k)
    assert np.array_equal(D2.view(np.uint32), D.view(np.uint32))
    assert np.array_equal(perm[I2], I)
    # sortedness and self-consistency of the scores with the stored rows (canonical fp32 chain)
    assert np.all(D[:, :-1] >= D[:, 1:])
    for qi in (0, 150, 299):
        for j in (0, k - 1):
            want = np.float32(oracle_mod.score_pair(Q[qi], X[I[qi, j]], metric=oracle_mod.METRIC_IP, dtype=oracle_mod.DTYPE_BF16))
            assert want.view(np.uint32) == D[qi, j].view(np.uint32)


@pytest.mark.parametrize("metric,dtype,n,d,nq,k", [
    ("ip", "bf16", 70000, 384, 1500, 10),      # 6 query tiles per slice: the 16x16x32 variant of the bench line
    ("l2", "bf16", 70000, 384, 900, 10),       # the same with the accumulators started at -xx / 2
    ("ip", "f32", 70000, 384, 800, 10),        # fp32 index: bf16 shadow + fp32 re-score
    ("ip", "bf16", 90000, 128, 1000, 5),       # one k-block per row
    ("ip", "bf16", 50000, 768, 520, 10),       # six k-blocks, four waves per workgroup (128-query tiles)
    ("ip", "bf16", 60000, 384, 800, 100),      # K' = 128 lists
    ("ip", "bf16", 50000, 768, 520, 100),      # BASELINE config 5's instance: 768-d, K' = 128, eight waves x 16 queries
    ("l2", "bf16", 50000, 768, 520, 100),      # the same with the accumulators started at -xx / 2
    ("ip", "f32", 50000, 768, 300, 100),       # fp32 768-d index: bf16 shadow + fp32 re-score
])
def test_mfma_bound_variant_matches_oracle(oracle_mod, metric, dtype, n, d, nq, k):
    """More than two query tiles per slice: the plan takes the MFMA-bound kernel variant (16x16x32 MFMA, mid-unit
    barrier, progress words) — the one the bench line runs.  Rows and queries are not unit norm."""
    from rag_amd.index import MODE_FAST
    rng = np.random.default_rng(n + d + nq)
    X = _unit(rng, n, d) * rng.uniform(0.7, 1.3, size=(n, 1)).astype(np.float32)
    Q = _unit(rng, nq, d) * rng.uniform(0.5, 2.0, size=(nq, 1)).astype(np.float32)
    idx = _check(oracle_mod, X, Q, k, metric, dtype, mode=MODE_FAST)
    st = idx.stats()
    assert st["fast"] + st["exact"] == nq
    if dtype == "bf16":
        assert st["fast"] >= nq * 0.8, st


_PLAN_CHILD = r"""
import sys, numpy as np
sys.path.insert(0, %r)
import oracle
from rag_amd.index import FlatIndex, MODE_FAST
rng = np.random.default_rng(5)
def unit(n, d):
    x = rng.standard_normal((n, d)).astype(np.float32)
    return x / np.linalg.norm(x, axis=1, keepdims=True)
X, Q = unit(80000, 384), unit(700, 384)
idx = FlatIndex(384, metric="ip", dtype="bf16"); idx.set_mode(MODE_FAST); idx.add(X)
D, I = idx.search(Q, 10)
Do, Io = oracle.search(X, Q, 10, metric=oracle.METRIC_IP, dtype=oracle.DTYPE_BF16)
assert np.array_equal(I, Io) and np.array_equal(D.view(np.uint32), Do.view(np.uint32))
print("OK", idx.stats())
"""


@pytest.mark.parametrize("env", [{"MI_MFMA_M16": "0", "MI_MFMA_PD": "2"}, {"MI_MFMA_M16": "0", "MI_MFMA_PD": "3"},
                                 {"MI_MFMA_NS": "8"}, {"MI_MFMA_NS": "64", "MI_MFMA_M16": "1", "MI_MFMA_PD": "2"}])
def test_plan_overrides_do_not_change_the_answer(env):
    """The scan's plan switches (kernel variant, prefetch distance, number of slices) pick a different kernel or
    a different split of the rows; the answer stays bit-exact.  Read once per process: child processes."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = dict(os.environ)
    e.update(env)
    out = subprocess.run([sys.executable, "-c", _PLAN_CHILD % root], env=e, capture_output=True, text=True, timeout=170)
    assert out.returncode == 0 and "OK" in out.stdout, (env, out.stdout[-500:], out.stderr[-2000:])


def _exact_cross_check(idx, Q, D, I, k, planted_n):
    """at sizes the oracle cannot follow, 64 of the queries (the planted ones included) are answered
    a second time by MODE_EXACT -- search_exact.hip, an independent kernel family (fp32 matrix pipe / VALU fmaf chains +
    radix select) that is held to the oracle's bits at every smaller size -- and must agree with the MFMA path's answer
    in ids and score bits.  A true neighbour the MFMA filter dropped would show here."""
    from rag_amd.index import MODE_EXACT, MODE_FAST
    nq = Q.shape[0]
    sel = sorted(set(range(planted_n)) | set(np.linspace(planted_n, nq - 1, 64 - planted_n).astype(int).tolist()))
    before = idx.stats()
    idx.set_mode(MODE_EXACT)
    try:
        De, Ie = idx.search(np.ascontiguousarray(Q[sel]), k)
    finally:
        idx.set_mode(MODE_FAST)
    after = idx.stats()
    assert idx.last_scan_time()["kind"] == "exact_f32" and after["exact"] - before["exact"] == len(sel), (before, after)
    np.testing.assert_array_equal(Ie, I[sel])
    np.testing.assert_array_equal(De.view(np.uint32), D[sel].view(np.uint32))
    return len(sel)


def test_config5_size_properties(oracle_mod):
    """BASELINE config 5 at its full single-GPU size (50M x 768 bf16 = 76.8 GB, top-100, 10k queries), where the oracle
    cannot follow: best-first order, reported scores equal to the canonical chain of rows regenerated on the CPU,
    planted rows first, a query's answer independent of the batch (and so of the kernel variant) it travelled in,
    no exact-path fallback."""
    import torch
    from rag_amd.index import FlatIndex, MODE_FAST
    free, _ = torch.cuda.mem_get_info()
    if free < 100e9:
        pytest.skip("needs ~90 GB of HBM")
    n, d, nq, k = 50_000_000, 768, 10_000, 100
    idx = FlatIndex(d, metric="ip", dtype="bf16")
    idx.set_mode(MODE_FAST)
    idx.add_synthetic(n, seed=0)
    Q = oracle_mod.synth_rows(1, 0, nq, d)
    planted = [0, 123_456, 25_000_000, n - 1]
    for j, r in enumerate(planted):
        Q[j] = oracle_mod.synth_rows(0, r, 1, d)[0]
    D, I = idx.search(Q, k)
    st = idx.stats()
    assert st["fast"] == nq and st["exact"] == 0, st
    assert np.all(D[:, :-1] >= D[:, 1:]) and I.min() >= 0 and I.max() < n
    assert all(len(set(row)) == k for row in I[::997])
    for j, r in enumerate(planted):
        assert I[j, 0] == r, (j, r, I[j, :3])
    for qi in (0, 3, 17, 5000, nq - 1):
        for j in (0, 49, k - 1):
            x = oracle_mod.synth_rows(0, int(I[qi, j]), 1, d)[0]
            want = np.float32(oracle_mod.score_pair(Q[qi], x, metric=oracle_mod.METRIC_IP, dtype=oracle_mod.DTYPE_BF16))
            assert want.view(np.uint32) == D[qi, j].view(np.uint32), (qi, j)
    sub = [1, 4, 99, 2500, 7777, 9000, nq - 1]
    D7, I7 = idx.search(Q[sub], k)
    assert np.array_equal(I7, I[sub]) and np.array_equal(D7.view(np.uint32), D[sub].view(np.uint32))
    scan = idx.last_scan_time()
    ne = _exact_cross_check(idx, Q, D, I, k, len(planted))
    print(f"config 5: scan {scan}; {ne} queries bit-equal on the exact path")


def test_bench_size_properties(oracle_mod):
    """BASELINE config 4 at its full single-GPU size (100M x 384 bf16, 10k queries, top-10), where the oracle cannot
    follow: (1) best-first order; (2) every reported score is the canonical fp32 chain of the stored row (a sample,
    rows regenerated on the CPU from the counter-based generator); (3) planted queries -- copies of corpus rows --
    come back with their own row first; (4) a query's answer is the same whether it travels in the 10k batch (MFMA-
    bound kernel variant) or in a batch of 7 (HBM-bound variant); (5) no query needed the exact fallback."""
    import torch
    from rag_amd.index import FlatIndex, MODE_FAST
    free, _ = torch.cuda.mem_get_info()
    if free < 100e9:
        pytest.skip("needs ~90 GB of HBM")
    n, d, nq, k = 100_000_000, 384, 10_000, 10
    idx = FlatIndex(d, metric="ip", dtype="bf16")
    idx.set_mode(MODE_FAST)
    idx.add_synthetic(n, seed=0)
    Q = oracle_mod.synth_rows(1, 0, nq, d)
    planted = [0, 123_456, 50_000_000, n - 1]
    for j, r in enumerate(planted):
        Q[j] = oracle_mod.synth_rows(0, r, 1, d)[0]
    D, I = idx.search(Q, k)
    st = idx.stats()
    assert st["fast"] == nq and st["exact"] == 0, st
    assert np.all(D[:, :-1] >= D[:, 1:]) and I.min() >= 0 and I.max() < n
    for j, r in enumerate(planted):
        assert I[j, 0] == r, (j, r, I[j])
    for qi in (0, 3, 17, 5000, nq - 1):
        for j in (0, 4, k - 1):
            x = oracle_mod.synth_rows(0, int(I[qi, j]), 1, d)[0]
            want = np.float32(oracle_mod.score_pair(Q[qi], x, metric=oracle_mod.METRIC_IP, dtype=oracle_mod.DTYPE_BF16))
            assert want.view(np.uint32) == D[qi, j].view(np.uint32), (qi, j)
    sub = [1, 4, 99, 2500, 7777, 9000, nq - 1]
    D7, I7 = idx.search(Q[sub], k)
    assert np.array_equal(I7, I[sub]) and np.array_equal(D7.view(np.uint32), D[sub].view(np.uint32))
    ne = _exact_cross_check(idx, Q, D, I, k, len(planted))
    print(f"config 4: {ne} queries bit-equal on the exact path")


@pytest.mark.parametrize("metric,dtype", [("ip", "bf16"), ("l2", "bf16"), ("ip", "f32"), ("l2", "f32")])
def test_threshold_seed_keeps_the_answer(oracle_mod, metric, dtype, monkeypatch):
    """The seed pass (more than 128 queries, k <= 15) starts every slice at the 16th largest of 64 class maxima of a row
    sample.  Adversarial for it: the sample (the index's first rows) holds near-copies of some queries -- so their seed
    is as tight as it can get -- zero rows, a block of identical rows and, in a second pass, a NaN row; the true neighbours of other queries
    lie only behind the sample.  Seeded and unseeded (MI_MFMA_SEED=0, read per call) answers equal the oracle's bits."""
    from rag_amd.index import FlatIndex, MODE_FAST
    n, d, nq, k = 300_000, 384, 300, 10
    rng = np.random.default_rng(77)
    X = _unit(rng, n, d)
    Q = _unit(rng, nq, d)
    for j in range(20):                                   # twenty near-copies of query 0 inside the sample, six of query 1
        X[50 + 37 * j] = Q[0] + 0.01 * (j + 1) * X[50 + 37 * j]
    for j in range(6):
        X[900 + 11 * j] = Q[1] * (1.0 - 0.001 * j)
    for j in range(12):                                   # query 2's neighbours lie far behind the sample
        X[200_000 + 1000 * j] = Q[2] * (1.0 - 0.002 * j)
    X[300:310] = 0.0
    X[5000:5040] = X[5000]                                # identical rows: ties across the class maxima
    m = {"ip": oracle_mod.METRIC_IP, "l2": oracle_mod.METRIC_L2}[metric]
    t = {"f32": oracle_mod.DTYPE_F32, "bf16": oracle_mod.DTYPE_BF16}[dtype]
    for with_nan in (False, True):
        if with_nan:
            X[123] = np.nan                               # (a NaN norm: nothing is certified, every query re-runs exactly)
        Do, Io = oracle_mod.search(X, Q, k, metric=m, dtype=t)
        idx = FlatIndex(d, metric=metric, dtype=dtype)
        idx.set_mode(MODE_FAST)
        idx.add(X)
        for seed in ("1", "0"):
            monkeypatch.setenv("MI_MFMA_SEED", seed)
            before = idx.stats()
            D, I = idx.search(Q, k)
            after = idx.stats()
            np.testing.assert_array_equal(I, Io)
            np.testing.assert_array_equal(D.view(np.uint32), Do.view(np.uint32))
            if not with_nan:
                assert after["fast"] - before["fast"] >= 0.9 * nq, (seed, before, after)
