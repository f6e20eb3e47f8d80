"""GPU: the sharded search composed from the HIP engine and the HIP merge on ONE device (SURVEY.md §4 item 4,
§8e): G FlatIndex row slices with set_id_base, each writing its packed block as a rank would, the blocks laid
side by side as the all-gather leaves them, merged in place -- equal to the single-index answer bit for bit.
(The collective itself is covered by the world-2 gloo test; RCCL only moves the blocks.)"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _unit(rng, n, d):
    x = rng.standard_normal((n, d)).astype(np.float32)
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    return x


def _sharded_answer(X, Q, k, metric, dtype, G, mode=None):
    import torch
    from rag_amd.distributed import packed_layout, shard_range
    from rag_amd.index import FlatIndex, topk_merge_packed_torch, topk_merge_torch
    n, d = X.shape
    nq = Q.shape[0]
    off, block = packed_layout(nq, k)
    q = torch.from_numpy(Q).cuda()
    gathered = torch.empty((G, block), dtype=torch.uint8, device="cuda")
    shards = []
    for r in range(G):
        lo, hi = shard_range(n, G, r)
        idx = FlatIndex(d, metric=metric, dtype=dtype)
        idx.set_id_base(lo)
        if mode is not None:
            idx.set_mode(mode)
        if hi > lo:
            idx.add(X[lo:hi])
        idx.search_packed(q, k, gathered[r])
        shards.append(idx)
    D, I = topk_merge_packed_torch(gathered, nq, G, k, k, metric)
    # the same lists as one row per query, lists side by side: the other merge entry point
    Ds = torch.cat([gathered[r, :nq * k * 4].view(torch.float32).view(nq, k) for r in range(G)], dim=1)
    Is = torch.cat([gathered[r, off:off + nq * k * 8].view(torch.int64).view(nq, k) for r in range(G)], dim=1)
    D2, I2 = topk_merge_torch(Ds, Is, k, metric)
    torch.cuda.synchronize()
    assert torch.equal(I, I2) and torch.equal(D.view(torch.int32), D2.view(torch.int32))
    return D.cpu().numpy(), I.cpu().numpy(), shards


@pytest.mark.parametrize("G", [2, 8])
@pytest.mark.parametrize("metric,dtype", [("ip", "f32"), ("l2", "bf16")])
def test_row_slices_merge_to_the_single_index_answer(oracle_mod, G, metric, dtype):
    """Small shards (exact path): duplicate rows straddle every shard boundary (ties must resolve to the lower GLOBAL
    id across shards) and k exceeds the rows of a shard (lists padded with id -1)."""
    from rag_amd.distributed import shard_range
    rng = np.random.default_rng(G)
    n, d, k = 403, 64, 70                                  # 8 shards of 51 rows < k
    X = _unit(rng, n, d)
    for r in range(1, G):
        lo, _ = shard_range(n, G, r)
        X[lo - 1] = X[lo] = X[7]                          # the same row on both sides of the boundary
    Q = np.concatenate([X[7:8], _unit(rng, 6, d)])
    D, I, _ = _sharded_answer(X, Q, k, metric, dtype, G)
    m = {"ip": oracle_mod.METRIC_IP, "l2": oracle_mod.METRIC_L2}[metric]
    t = {"f32": oracle_mod.DTYPE_F32, "bf16": oracle_mod.DTYPE_BF16}[dtype]
    Do, Io = oracle_mod.search(X, Q, k, metric=m, dtype=t)
    np.testing.assert_array_equal(I, Io)
    np.testing.assert_array_equal(D.view(np.uint32), Do.view(np.uint32))
    assert list(I[0, :3]) == sorted(I[0, :3])             # the planted duplicates: ascending global ids


@pytest.mark.parametrize("G", [2, 8])
def test_mfma_shards_merge_to_the_single_index_answer(oracle_mod, G):
    """Shards large enough for the MFMA scan + certificate (>= 32768 rows each), bf16, IP, 300 queries, and the whole
    corpus in ONE index as the reference answer (HIP) next to the oracle's."""
    from rag_amd.index import FlatIndex, MODE_FAST
    n, d, nq, k = G * 33_000 + 17, 128, 300, 10
    X = oracle_mod.synth_rows(61, 0, n, d)
    X[33_000 - 1] = X[33_000] = X[5]
    Q = oracle_mod.synth_rows(62, 0, nq, d)
    Q[0] = X[5]
    D, I, shards = _sharded_answer(X, Q, k, "ip", "bf16", G, mode=MODE_FAST)
    assert all(s.stats()["fast"] + s.stats()["exact"] == nq for s in shards)
    Do, Io = oracle_mod.search(X, Q, k, metric=oracle_mod.METRIC_IP, dtype=oracle_mod.DTYPE_BF16)
    np.testing.assert_array_equal(I, Io)
    np.testing.assert_array_equal(D.view(np.uint32), Do.view(np.uint32))
    one = FlatIndex(d, metric="ip", dtype="bf16")
    one.add(X)
    D1, I1 = one.search(Q, k)
    np.testing.assert_array_equal(I, I1)
    np.testing.assert_array_equal(D.view(np.uint32), D1.view(np.uint32))


@pytest.mark.parametrize("nlists,k_in,k", [(8, 1000, 1000), (8, 2048, 2048), (3, 1500, 700), (40, 300, 10)])
def test_merge_beyond_4096_entries_per_query(oracle_mod, nlists, k_in, k):
    """k = 1000 on 8 ranks is 8000 entries per query: merged in passes of 4096-entry chunks (mi_topk_merge_dev and
    the packed form), equal to the oracle's merge; ties, NaN-free scores, padding ids."""
    import torch
    from rag_amd.distributed import packed_layout
    from rag_amd.index import topk_merge_packed_torch, topk_merge_torch
    rng = np.random.default_rng(nlists * k)
    nq = 5
    m = nlists * k_in
    s = rng.standard_normal((nq, m)).astype(np.float32)
    s[:, 100:160] = s[:, 99:100]                          # a run of ties
    ids = np.stack([rng.permutation(10**7)[:m] for _ in range(nq)]).astype(np.int64)
    ids[:, -37:] = -1                                     # padding at the end of the last list
    # each list best-first, as a rank delivers it
    for li in range(nlists):
        blk = slice(li * k_in, (li + 1) * k_in)
        key = np.where(ids[:, blk] >= 0, s[:, blk], -np.inf)
        o = np.argsort(-key, axis=1, kind="stable")
        s[:, blk] = np.take_along_axis(s[:, blk], o, axis=1)
        ids[:, blk] = np.take_along_axis(ids[:, blk], o, axis=1)
    Do, Io = oracle_mod.topk_merge(s, ids, k, oracle_mod.METRIC_IP)
    D, I = topk_merge_torch(torch.from_numpy(s).cuda(), torch.from_numpy(ids).cuda(), k, "ip")
    np.testing.assert_array_equal(I.cpu().numpy(), Io)
    np.testing.assert_array_equal(D.cpu().numpy().view(np.uint32), Do.view(np.uint32))
    off, block = packed_layout(nq, k_in)
    packed = np.zeros((nlists, block), dtype=np.uint8)
    for li in range(nlists):
        packed[li, :nq * k_in * 4] = np.ascontiguousarray(s[:, li * k_in:(li + 1) * k_in]).view(np.uint8).reshape(-1)
        packed[li, off:off + nq * k_in * 8] = np.ascontiguousarray(ids[:, li * k_in:(li + 1) * k_in]).view(np.uint8).reshape(-1)
    Dp, Ip = topk_merge_packed_torch(torch.from_numpy(packed).cuda(), nq, nlists, k_in, k, "ip")
    np.testing.assert_array_equal(Ip.cpu().numpy(), Io)
    np.testing.assert_array_equal(Dp.cpu().numpy().view(np.uint32), Do.view(np.uint32))


@pytest.mark.parametrize("dtype", ["bf16", "f32"])
def test_shard_files_round_trip(tmp_path, oracle_mod, dtype):
    """Per-shard persistence: rows leave in the storage type (bf16 bit patterns, not widened
    to fp32), a shard file loads back with add_raw, and the reloaded index answers bit for bit like the original."""
    from rag_amd.distributed import ShardedFlatIndex
    n, d, k = 40_000, 384, 10
    sh = ShardedFlatIndex(d, metric="ip", dtype=dtype, device=0)
    sh.add_synthetic_global(n, seed=71)
    X = oracle_mod.synth_rows(71, 0, n, d)
    raw = sh.local.reconstruct_raw(100, 50)
    if dtype == "bf16":
        assert raw.dtype == np.uint16
        np.testing.assert_array_equal(raw, oracle_mod.round_bf16(X[100:150]))
    else:
        np.testing.assert_array_equal(raw.view(np.uint32), X[100:150].view(np.uint32))
    Q = oracle_mod.synth_rows(72, 0, 33, d)
    D, I = sh.search(Q, k)
    sh.save_local(str(tmp_path / "s"))
    sh2 = ShardedFlatIndex.load_local(str(tmp_path / "s"), device=0)
    assert sh2.local.ntotal == n
    D2, I2 = sh2.search(Q, k)
    np.testing.assert_array_equal(I2, I)
    np.testing.assert_array_equal(D2.view(np.uint32), D.view(np.uint32))
    m = oracle_mod.DTYPE_BF16 if dtype == "bf16" else oracle_mod.DTYPE_F32
    Do, Io = oracle_mod.search(X, Q, k, metric=oracle_mod.METRIC_IP, dtype=m)
    np.testing.assert_array_equal(I, Io)
