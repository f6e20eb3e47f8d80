"""CPU, world_size 2, gloo: the row-shard + all-gather + merge path of
rag_amd.distributed.ShardedFlatIndex equals the single-index answer."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, metric, dtype, n, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist
    import oracle
    from rag_amd.distributed import ShardedFlatIndex, shard_range
    from tests._fakes import OracleIndex
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(42)                      # same corpus/queries on every rank
        X = rng.integers(-4, 5, size=(n, 24)).astype(np.float32)   # integer data: plenty of exact ties
        Q = rng.integers(-4, 5, size=(9, 24)).astype(np.float32)
        m = {"ip": oracle.METRIC_IP, "l2": oracle.METRIC_L2}[metric]

        def merge(Dm, Im, k, metric_name):
            D, I = oracle.topk_merge(Dm.numpy(), Im.numpy(), k, m)
            return torch.from_numpy(D), torch.from_numpy(I)

        sh = ShardedFlatIndex(24, metric=metric, dtype=dtype, local_index=OracleIndex(24, metric, dtype), merge_fn=merge)
        sh.add_global(X)
        lo, hi = shard_range(n, world, rank)
        assert sh.local.ntotal == hi - lo and sh.local.id_base == lo
        for k in (1, 5, 40):
            D, I = sh.search(Q, k)
            np.save(os.path.join(out_dir, f"D_{rank}_{k}.npy"), D)
            np.save(os.path.join(out_dir, f"I_{rank}_{k}.npy"), I)
        sh.save_local(os.path.join(out_dir, "store"))            # one shard file per rank
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("metric,dtype,n", [("ip", "f32", 101), ("l2", "bf16", 64), ("ip", "bf16", 3)])
def test_two_rank_sharded_search_equals_single_index(tmp_path, oracle_mod, metric, dtype, n):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(2, port, metric, dtype, n, str(tmp_path)), nprocs=2, join=True)
    rng = np.random.default_rng(42)
    X = rng.integers(-4, 5, size=(n, 24)).astype(np.float32)
    Q = rng.integers(-4, 5, size=(9, 24)).astype(np.float32)
    m = {"ip": oracle_mod.METRIC_IP, "l2": oracle_mod.METRIC_L2}[metric]
    t = {"f32": oracle_mod.DTYPE_F32, "bf16": oracle_mod.DTYPE_BF16}[dtype]
    for k in (1, 5, 40):
        Dw, Iw = oracle_mod.search(X, Q, k, metric=m, dtype=t)
        for rank in (0, 1):                                   # every rank holds the identical, exact answer
            D = np.load(tmp_path / f"D_{rank}_{k}.npy")
            I = np.load(tmp_path / f"I_{rank}_{k}.npy")
            np.testing.assert_array_equal(I, Iw)
            np.testing.assert_array_equal(D.view(np.uint32), Dw.view(np.uint32))
    # the per-rank shard files, loaded back on ONE rank (re-sharding on load), give the same answer
    from rag_amd.distributed import ShardedFlatIndex
    from tests._fakes import OracleIndex
    import torch
    assert sorted(f for f in os.listdir(tmp_path / "store") if f.endswith(".npy")) == ["index.shard000.rows.npy", "index.shard001.rows.npy"]
    one = ShardedFlatIndex.load_local(str(tmp_path / "store"), local_index=OracleIndex(24, metric, dtype),
                                      merge_fn=lambda Dm, Im, k, mn: tuple(torch.from_numpy(a) for a in oracle_mod.topk_merge(Dm.numpy(), Im.numpy(), k, m)))
    assert one.local.ntotal == n and one.n_total == n
    D1, I1 = one.search(Q, 5)
    Dw, Iw = oracle_mod.search(X, Q, 5, metric=m, dtype=t)
    np.testing.assert_array_equal(I1, Iw)
    np.testing.assert_array_equal(D1.view(np.uint32), Dw.view(np.uint32))


def _embed_worker(rank, world, port, n, gather, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    import oracle
    from rag_amd.distributed import embed_sharded, shard_range
    from tests._fakes import OracleEncoder
    from tests.test_distributed_cpu import _embed_inputs
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cfg, blob, flat, offsets = _embed_inputs(oracle, n)
        enc = OracleEncoder(cfg, blob)
        E = embed_sharded(enc, flat, offsets, gather=gather, max_len=12)
        lo, hi = shard_range(n, world, rank)
        assert E.shape == ((n if gather else hi - lo), cfg.hidden_size) and enc.calls == (1 if hi > lo else 0)
        np.save(os.path.join(out_dir, f"E_{rank}.npy"), E)
    finally:
        dist.destroy_process_group()


def _embed_inputs(oracle, n):
    from rag_amd._capi import BertConfig
    from rag_amd.encoder_weights import random_blob
    vals = (50, 32, 1, 2, 64, 16, 2, 1e-12, 1)
    blob = random_blob(BertConfig(*vals), seed=2, init_range=0.2)
    rng = np.random.default_rng(9)
    lens = rng.integers(0, 15, size=n)                     # empty sequences and ones beyond max_len = 12 included
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    flat = rng.integers(0, 50, size=int(offsets[-1])).astype(np.int32)
    return oracle.BertConfig(*vals), blob, flat, offsets


@pytest.mark.parametrize("n,gather", [(11, True), (11, False), (1, True)])
def test_two_rank_sharded_embedding_equals_single_rank(tmp_path, oracle_mod, n, gather):
    """embed_sharded: contiguous chunk ranges per rank, no collective on the path, optional all-gather of the
    [n][hidden] result (BASELINE config 3's data-parallel split), against one rank encoding everything."""
    import torch.multiprocessing as mp
    from rag_amd.distributed import shard_range
    from tests._fakes import OracleEncoder
    mp.spawn(_embed_worker, args=(2, _free_port(), n, gather, str(tmp_path)), nprocs=2, join=True)
    cfg, blob, flat, offsets = _embed_inputs(oracle_mod, n)
    want = OracleEncoder(cfg, blob).encode_ragged(flat, offsets, max_len=12)
    for rank in (0, 1):
        E = np.load(tmp_path / f"E_{rank}.npy")
        lo, hi = shard_range(n, 2, rank)
        np.testing.assert_array_equal(E, want if gather else want[lo:hi])


def test_shard_ranges_partition_the_rows():
    from rag_amd.distributed import shard_range
    for n in (0, 1, 7, 100, 100_000_001):
        for world in (1, 2, 4, 8):
            r = [shard_range(n, world, i) for i in range(world)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(r, r[1:]))


def test_world_of_one_host_path_and_dtype_spelling(tmp_path, oracle_mod):
    """No process group, a host engine and no merge_fn: the local lists are the answer; dtype='BF16' (any case) saves bf16
    bit patterns, not a numeric cast of them, and loads back to the same answer."""
    from rag_amd.distributed import ShardedFlatIndex
    from tests._fakes import OracleIndex
    rng = np.random.default_rng(5)
    X = rng.standard_normal((300, 24)).astype(np.float32)
    Q = rng.standard_normal((4, 24)).astype(np.float32)
    sh = ShardedFlatIndex(24, metric="ip", dtype="BF16", local_index=OracleIndex(24, "ip", "bf16"))
    assert sh.dtype == "bf16"
    sh.add_global(X)
    D, I = sh.search(Q, 7)
    Do, Io = oracle_mod.search(X, Q, 7, metric=oracle_mod.METRIC_IP, dtype=oracle_mod.DTYPE_BF16)
    assert np.array_equal(I, Io) and np.array_equal(D.view(np.uint32), Do.view(np.uint32))
    sh.save_local(str(tmp_path / "s"))
    rows = np.load(str(tmp_path / "s" / "index.shard000.rows.npy"))
    assert rows.dtype == np.uint16 and np.array_equal(rows, oracle_mod.round_bf16(X))
    back = ShardedFlatIndex.load_local(str(tmp_path / "s"), local_index=OracleIndex(24, "ip", "bf16"))
    D2, I2 = back.search(Q, 7)
    assert np.array_equal(I2, Io) and np.array_equal(D2.view(np.uint32), Do.view(np.uint32))
