"""CPU: the oracle (oracle/oracle_search.c) against the committed golden vectors and an
independent float64 NumPy brute force; plus the domain properties the GPU tests rely on."""
import os

import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

GOLD = os.path.join(os.path.dirname(__file__), "golden", "search_small.npz")


def _unit(rng, n, d):
    x = rng.standard_normal((n, d)).astype(np.float32)
    return x / np.linalg.norm(x, axis=1, keepdims=True)


@pytest.mark.parametrize("metric", ["ip", "l2"])
@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("k", [1, 4, 5, 10, 100])
def test_oracle_matches_golden_bit_exact(oracle_mod, metric, dtype, k):
    g = np.load(GOLD)
    m = {"ip": oracle_mod.METRIC_IP, "l2": oracle_mod.METRIC_L2}[metric]
    t = {"f32": oracle_mod.DTYPE_F32, "bf16": oracle_mod.DTYPE_BF16}[dtype]
    D, I = oracle_mod.search(g["X"], g["Q"], k, metric=m, dtype=t)
    np.testing.assert_array_equal(I, g[f"ids_{metric}_{dtype}_k{k}"])
    np.testing.assert_array_equal(D.view(np.uint32), g[f"scores_{metric}_{dtype}_k{k}"].view(np.uint32))


def test_golden_has_the_planted_ties():
    g = np.load(GOLD)
    ids = g["ids_ip_f32_k10"][0]
    # rows 3, 40..46, 90 are identical to query 0: nine exact ties, returned lowest id first
    assert list(ids[:9]) == [3, 40, 41, 42, 43, 44, 45, 46, 90]


def test_oracle_vs_float64_bruteforce(oracle_mod):
    rng = np.random.default_rng(0)
    X, Q = _unit(rng, 30000, 384), _unit(rng, 25, 384)
    D, I = oracle_mod.search(X, Q, 10)
    ref = Q.astype(np.float64) @ X.T.astype(np.float64)
    order = np.argsort(-ref, axis=1, kind="stable")[:, :10]
    # random data: no near-ties at fp32 resolution, so the index sets agree with float64
    np.testing.assert_array_equal(I, order)
    assert np.abs(np.take_along_axis(ref, I, 1) - D).max() < 1e-6
    D2, I2 = oracle_mod.search(X, Q, 10, metric=oracle_mod.METRIC_L2)
    np.testing.assert_array_equal(I2, I)          # unit vectors: L2 order == IP order
    assert np.all(np.diff(D2, axis=1) >= 0) and np.all(np.diff(D, axis=1) <= 0)


def test_thread_count_does_not_change_the_answer(oracle_mod):
    rng = np.random.default_rng(1)
    X, Q = _unit(rng, 20000, 96), _unit(rng, 40, 96)
    a = oracle_mod.search(X, Q, 7, nthreads=1)
    b = oracle_mod.search(X, Q, 7, nthreads=4)
    np.testing.assert_array_equal(a[1], b[1])
    np.testing.assert_array_equal(a[0].view(np.uint32), b[0].view(np.uint32))


def test_padding_and_empty(oracle_mod):
    rng = np.random.default_rng(2)
    X, Q = _unit(rng, 3, 16), _unit(rng, 2, 16)
    D, I = oracle_mod.search(X, Q, 5)
    assert np.all(I[:, 3:] == -1) and np.all(np.isneginf(D[:, 3:]))
    D, I = oracle_mod.search(X, Q, 5, metric=oracle_mod.METRIC_L2)
    assert np.all(I[:, 3:] == -1) and np.all(np.isposinf(D[:, 3:]))
    D, I = oracle_mod.search(X[:0], Q, 2)
    assert np.all(I == -1)


def test_nan_ranks_last_and_negative_zero_is_canonical(oracle_mod):
    X = np.zeros((4, 8), dtype=np.float32)
    X[1, 0] = np.nan
    X[2, 0] = -1.0
    X[3, 0] = 1.0
    Q = np.zeros((1, 8), dtype=np.float32)
    Q[0, 0] = 1.0
    D, I = oracle_mod.search(X, Q, 4)
    assert list(I[0]) == [3, 0, 2, 1]
    assert D[0, 1] == 0.0 and not np.signbit(D[0, 1]) and np.isneginf(D[0, 3])


@settings(max_examples=25, deadline=None)
@given(st.integers(1, 300), st.integers(1, 40), st.integers(1, 12), st.integers(0, 2**31 - 1),
       st.sampled_from(["ip", "l2"]), st.sampled_from(["f32", "bf16"]))
def test_properties(n, d, k, seed, metric, dtype):
    import oracle
    rng = np.random.default_rng(seed)
    X = rng.integers(-3, 4, size=(n, d)).astype(np.float32)     # small integers: many exact ties
    Q = rng.integers(-3, 4, size=(3, d)).astype(np.float32)
    m = {"ip": oracle.METRIC_IP, "l2": oracle.METRIC_L2}[metric]
    t = {"f32": oracle.DTYPE_F32, "bf16": oracle.DTYPE_BF16}[dtype]
    D, I = oracle.search(X, Q, k, metric=m, dtype=t)
    # integer data: fp32 arithmetic is exact, so NumPy int math is an independent reference
    Xi, Qi = X.astype(np.int64), Q.astype(np.int64)
    dot = Qi @ Xi.T
    sc = dot if metric == "ip" else ((Qi * Qi).sum(1)[:, None] + (Xi * Xi).sum(1)[None, :] - 2 * dot)
    for qi in range(3):
        key = [(-sc[qi, i], i) if metric == "ip" else (sc[qi, i], i) for i in range(n)]
        key.sort()
        want = [i for _, i in key[:k]] + [-1] * max(0, k - n)
        assert list(I[qi]) == want
        for j in range(min(k, n)):
            assert D[qi, j] == float(sc[qi, want[j]])
    # shards + merge == whole (the multi-GPU path's invariant)
    cut = n // 2
    parts = []
    for lo, hi in ((0, cut), (cut, n)):
        parts.append(oracle.search(X[lo:hi], Q, k, metric=m, dtype=t, id_base=lo))
    Dm, Im = oracle.topk_merge(np.concatenate([p[0] for p in parts], 1), np.concatenate([p[1] for p in parts], 1), k, m)
    np.testing.assert_array_equal(Im, I)
    np.testing.assert_array_equal(Dm.view(np.uint32), D.view(np.uint32))


def test_synth_generator_is_unit_norm_and_deterministic(oracle_mod):
    a = oracle_mod.synth_rows(0, 10, 50, 384)
    b = oracle_mod.synth_rows(0, 30, 10, 384)
    np.testing.assert_array_equal(a[20:30], b)                   # keyed by the global row number
    assert np.allclose(np.linalg.norm(a, axis=1), 1.0, atol=1e-6)
    c = oracle_mod.synth_rows(1, 10, 50, 384)
    assert np.abs((a * c).sum(1)).max() < 0.3                    # other seed: unrelated rows
