"""GPU: parity hardening of the flat search: the committed golden fixture run through the HIP
path directly, adversarial inputs for the MFMA scan's certificate (mf_err_bound, search_mfma.hip), and BASELINE
config 2's shape.  Bar: bit-exact ids and score bits against the oracle / the fixture."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "search_small.npz")


def _unit(rng, n, d):
    x = rng.standard_normal((n, d)).astype(np.float32)
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    return x


def _run(oracle, X, Q, k, metric, dtype, mode):
    from rag_amd.index import FlatIndex
    idx = FlatIndex(X.shape[1], metric=metric, dtype=dtype)
    idx.set_mode(mode)
    idx.add(X)
    D, I = idx.search(Q, k)
    m = {"ip": oracle.METRIC_IP, "l2": oracle.METRIC_L2}[metric]
    t = {"f32": oracle.DTYPE_F32, "bf16": oracle.DTYPE_BF16}[dtype]
    Do, Io = oracle.search(X, Q, k, metric=m, dtype=t)
    np.testing.assert_array_equal(I, Io)
    np.testing.assert_array_equal(D.view(np.uint32), Do.view(np.uint32))
    return idx, D, I


def test_golden_fixture_through_the_hip_path():
    """tests/golden/search_small.npz (exact-rational fmaf chains, planted ties / duplicates / a zero row; written by
    make_golden.py independently of oracle/ and rag_amd/) against the HIP exact path, every (metric, dtype, k)."""
    from rag_amd.index import FlatIndex
    g = np.load(GOLD)
    X, Q = g["X"], g["Q"]
    for metric in ("ip", "l2"):
        for dtype in ("f32", "bf16"):
            idx = FlatIndex(X.shape[1], metric=metric, dtype=dtype)
            idx.add(X)
            for k in (1, 4, 5, 10, 100):
                D, I = idx.search(Q, k)
                tag = f"{metric}_{dtype}_k{k}"
                np.testing.assert_array_equal(I, g[f"ids_{tag}"], err_msg=tag)
                np.testing.assert_array_equal(D.view(np.uint32), g[f"scores_{tag}"].view(np.uint32), err_msg=tag)


@pytest.mark.parametrize("dtype", ["bf16", "f32"])
def test_one_ulp_ladder_under_heavy_cancellation(oracle_mod, dtype):
    """41 rows whose canonical scores for query 0 are consecutive fp32 numbers (1 ulp apart), on top of two terms of
    +-64 that cancel exactly in the canonical order but not in the MFMA's: the scan's scores of those rows are noise
    at the 1e-5 level, so the filter cannot order them and the certificate must not claim it did.  k = 10 cuts the
    ladder in the middle.  The answer stays bit-exact; query 0 is answered by the exact re-run."""
    from rag_amd.index import MODE_FAST
    rng = np.random.default_rng(17)
    n, d, k = 40_000, 384, 10
    X = _unit(rng, n, d) * 0.25
    q0 = np.zeros(d, dtype=np.float32)
    q0[0] = q0[1] = 8.0
    q0[2] = 1.0
    q0[d - 1] = 1.0
    base = np.zeros(d, dtype=np.float32)
    base[0], base[1], base[2] = 8.0, -8.0, 0.75             # 64 - 64 + 0.75: partial sum in [0.5, 1)
    rows = rng.permutation(np.arange(20_000, 20_500))[:41]
    for j, r in enumerate(rows):
        X[r] = base
        X[r, d - 1] = j * 2.0 ** -24                          # bf16-representable; adds j ulps exactly
    Q = np.concatenate([q0[None, :], _unit(rng, 40, d)])
    want = np.array([np.float32(oracle_mod.score_pair(q0, X[r], metric=oracle_mod.METRIC_IP,
                                                      dtype={"f32": oracle_mod.DTYPE_F32, "bf16": oracle_mod.DTYPE_BF16}[dtype]))
                     for r in rows])
    assert np.all(np.diff(want.view(np.uint32).astype(np.int64)) == 1)      # the ladder is what it claims to be
    idx, D, I = _run(oracle_mod, X, Q, k, "ip", dtype, MODE_FAST)
    assert set(I[0]) == set(rows[-k:]) and list(D[0].view(np.uint32)) == sorted(D[0].view(np.uint32), reverse=True)
    st = idx.stats()
    assert st["exact"] >= 1 and st["fast"] + st["exact"] == len(Q), st
    # (the bound is built on the LARGEST row norm of the index, 11.3 here against 0.25 for the rest, so it also sends
    # part of the ordinary queries to the exact path: correct, only slower)
    print(f"ladder, {dtype}: certified {st['fast']}/{len(Q)}")


@pytest.mark.parametrize("metric", ["ip", "l2"])
@pytest.mark.parametrize("scale_x,scale_q", [(1e-20, 1e-20), (1e-30, 1.0), (1e18, 1e18), (1e18, 1e-30)])
def test_denormal_and_huge_scales(oracle_mod, metric, scale_x, scale_q):
    """Rows and queries scaled so the products are fp32 denormals (1e-40) or near the top of the range (1e36): the
    canonical chain keeps denormals, the MFMA may flush them -- the error bound must not shrink to zero with the
    norms -- and nothing overflows into a wrong order.  Both index dtypes."""
    from rag_amd.index import MODE_FAST
    rng = np.random.default_rng(23)
    n, d, k = 40_000, 128, 10
    X = (_unit(rng, n, d).astype(np.float64) * scale_x).astype(np.float32)
    Q = (_unit(rng, 9, d).astype(np.float64) * scale_q).astype(np.float32)
    for dtype in ("bf16", "f32"):
        idx, D, I = _run(oracle_mod, X, Q, k, metric, dtype, MODE_FAST)
        st = idx.stats()
        assert st["fast"] + st["exact"] == 9, st
        print(f"scale {scale_x:g} x {scale_q:g} {metric} {dtype}: certified {st['fast']}/9")


def test_signed_zero_rows(oracle_mod):
    """Rows of +0, -0 and mixed zeros all score +0.0 canonically (the chain starts from +0.0f) and tie by id; with every
    other row scoring negative they are the top of the list."""
    from rag_amd.index import MODE_FAST
    rng = np.random.default_rng(29)
    n, d, k = 40_000, 128, 12
    q = np.abs(_unit(rng, 1, d))
    X = -np.abs(_unit(rng, n, d))                            # every dot is negative
    X[100] = 0.0
    X[39_000] = -0.0
    X[20_000] = np.where(np.arange(d) % 2 == 0, 0.0, -0.0)
    X[5] = 0.0
    Q = np.concatenate([q, -q, _unit(rng, 3, d)])
    for metric in ("ip", "l2"):
        for dtype in ("bf16", "f32"):
            idx, D, I = _run(oracle_mod, X.astype(np.float32), Q, k, metric, dtype, MODE_FAST)
            if metric == "ip":
                assert list(I[0, :4]) == [5, 100, 20_000, 39_000] and np.all(D[0, :4].view(np.uint32) == 0)


def test_config2_shape_f32_index_auto_mode(oracle_mod):
    """BASELINE config 2: 1M x 384 fp32 corpus, exact top-10, AUTO mode (200 of the 1k queries: the oracle is a scalar
    chain).  The MFMA scan runs on the bf16 shadow, scores come from the fp32 rows; >= 90 % of the queries must be
    certified -- a silent fall-back to the exact scan would keep the answer right and lose the speed."""
    from rag_amd.index import FlatIndex
    n, d, nq, k = 1_000_000, 384, 200, 10
    idx = FlatIndex(d, metric="ip", dtype="f32")
    idx.add_synthetic(n, seed=41)
    X = oracle_mod.synth_rows(41, 0, n, d)
    Q = oracle_mod.synth_rows(42, 0, nq, d)
    D, I = idx.search(Q, k)
    Do, Io = oracle_mod.search(X, Q, k, metric=oracle_mod.METRIC_IP, dtype=oracle_mod.DTYPE_F32)
    np.testing.assert_array_equal(I, Io)
    np.testing.assert_array_equal(D.view(np.uint32), Do.view(np.uint32))
    st = idx.stats()
    print(f"config-2 shape: certified {st['fast']}/{nq}, exact re-runs {st['exact']}")
    assert idx.last_scan_time()["kind"] == "mfma_bf16" and st["fast"] >= 0.9 * nq, st
