"""GPU: ONE store over several GPUs from ONE process (mi_multi_index_*): advanced_rag.py is a single
Python process (:336-337) that builds one store (:295) and searches it from any thread (:296, :409).  A one-GPU box lists
device 0 several times -- one shard per entry, peer copies become device copies -- so everything but the xGMI transport
runs here: block-round-robin placement, per-shard host threads, local-to-global id rewrite, packed blocks gathered on the
first device, the HIP merge.  Bar: ids and score bits of the single-index answer and of the oracle.
UNMEASURED ON HARDWARE with more than one physical GPU (the boxes here have one)."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _unit(rng, n, d):
    x = rng.standard_normal((n, d)).astype(np.float32)
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    return x


def _oracle(oracle, X, Q, k, metric, dtype):
    m = {"ip": oracle.METRIC_IP, "l2": oracle.METRIC_L2}[metric]
    t = {"f32": oracle.DTYPE_F32, "bf16": oracle.DTYPE_BF16}[dtype]
    return oracle.search(X, Q, k, metric=m, dtype=t)


@pytest.mark.parametrize("G", [1, 3, 8])
@pytest.mark.parametrize("metric,dtype", [("ip", "f32"), ("l2", "bf16")])
def test_small_blocks_ties_across_shards_and_appends(oracle_mod, G, metric, dtype):
    """Blocks of 50 rows over G shards: the same row planted on both sides of every block boundary (ties must resolve to the
    lower GLOBAL id although the copies live on different shards), k larger than a shard, rows added in three uneven calls
    (an append lands in the middle of a block), host and device entry points, rows read back in global order."""
    import torch
    from rag_amd.index import MultiFlatIndex
    rng = np.random.default_rng(10 + G)
    n, d, k = 1_237, 64, 200
    X = _unit(rng, n, d)
    for b in range(50, n - 1, 50):
        X[b - 1] = X[b] = X[7]
    Q = np.concatenate([X[7:8], _unit(rng, 6, d)])
    idx = MultiFlatIndex(d, metric=metric, dtype=dtype, devices=[0] * G, block_rows=50)
    for lo, hi in ((0, 333), (333, 334), (334, n)):
        idx.add(X[lo:hi])
    assert idx.ntotal == n and sum(idx.shard_ntotal(s) for s in range(G)) == n
    if G == 8:
        assert idx.shard_ntotal(0) == 187 and idx.shard_ntotal(7) == 150      # 25 blocks: shard 0 holds blocks 0, 8, 16 and the 37 rows of 24
    D, I = idx.search(Q, k)
    Do, Io = _oracle(oracle_mod, X, Q, k, metric, dtype)
    np.testing.assert_array_equal(I, Io)
    np.testing.assert_array_equal(D.view(np.uint32), Do.view(np.uint32))
    assert list(I[0, :5]) == sorted(I[0, :5])
    Dt, It = idx.search_torch(torch.from_numpy(Q).cuda(), k)
    assert np.array_equal(It.cpu().numpy(), Io) and np.array_equal(Dt.cpu().numpy().view(np.uint32), Do.view(np.uint32))
    rows = idx.reconstruct_n(40, 300)
    want = oracle_mod.bf16_to_f32(oracle_mod.round_bf16(X[40:340])) if dtype == "bf16" else X[40:340]
    np.testing.assert_array_equal(rows, want)
    idx.reset()
    assert idx.ntotal == 0
    D0, I0 = idx.search(Q[:2], 3)
    assert np.all(I0 == -1)


def test_mfma_shards_equal_the_single_index(oracle_mod):
    """Shards large enough for the MFMA scan + certificate (default 16,384-row blocks, 3 shards of ~45k rows), bf16 / IP,
    300 queries: multi == one FlatIndex == oracle, and every (query, shard) search was answered (certified or re-run)."""
    from rag_amd.index import FlatIndex, MODE_FAST, MultiFlatIndex
    n, d, nq, k = 135_000, 128, 300, 10
    X = oracle_mod.synth_rows(71, 0, n, d)
    X[16_383] = X[16_384] = X[5]                            # a tie across the first block boundary (shard 0 | shard 1)
    Q = oracle_mod.synth_rows(72, 0, nq, d)
    Q[0] = X[5]
    multi = MultiFlatIndex(d, metric="ip", dtype="bf16", devices=[0, 0, 0])
    multi.set_mode(MODE_FAST)
    multi.add(X)
    D, I = multi.search(Q, k)
    one = FlatIndex(d, metric="ip", dtype="bf16")
    one.add(X)
    D1, I1 = one.search(Q, k)
    np.testing.assert_array_equal(I, I1)
    np.testing.assert_array_equal(D.view(np.uint32), D1.view(np.uint32))
    Do, Io = _oracle(oracle_mod, X, Q, k, "ip", "bf16")
    np.testing.assert_array_equal(I, Io)
    np.testing.assert_array_equal(D.view(np.uint32), Do.view(np.uint32))
    st = multi.stats()
    assert st["queries"] == nq and st["fast"] + st["exact"] == 3 * nq and st["fast"] > 0, st


def test_synthetic_rows_land_where_the_generator_says(oracle_mod):
    """mi_multi_index_add_synthetic places generator rows [0, n) block by block: reading them back in global order gives
    the oracle generator's rows, whatever the shard count (the bench's corpus on several GPUs)."""
    from rag_amd.index import MultiFlatIndex
    n, d = 70_000, 384
    idx = MultiFlatIndex(d, metric="ip", dtype="bf16", devices=[0, 0, 0, 0])
    idx.add_synthetic(40_001, seed=3)
    idx.add_synthetic(n - 40_001, seed=3)
    got = idx.reconstruct_n(0, n)
    want = oracle_mod.bf16_to_f32(oracle_mod.round_bf16(oracle_mod.synth_rows(3, 0, n, d)))
    np.testing.assert_array_equal(got, want)


def test_store_over_devices_is_the_drop_in(tmp_path):
    """FAISS.from_documents(..., devices=[...]) (advanced_rag.py:295-296): same documents, scores and order as the
    one-GPU store; save_local / load_local round trip onto another shard count; searched from 8 threads at once."""
    from rag_amd import FAISS, Document
    from tests._fakes import HashEmbeddings
    emb = HashEmbeddings(96)
    docs = [Document(page_content=f"chunk {i} of the corpus about topic {i % 17}", metadata={"i": i}) for i in range(900)]
    one = FAISS.from_documents(docs, emb)
    multi = FAISS.from_documents(docs, emb, devices=[0, 0, 0])
    assert type(multi.index).__name__ == "MultiFlatIndex" and multi.index.ntotal == 900
    queries = [f"topic {j}" for j in range(12)]
    want = [[(d.page_content, s) for d, s in one.similarity_search_with_score(q, k=5)] for q in queries]
    got = [[(d.page_content, s) for d, s in multi.similarity_search_with_score(q, k=5)] for q in queries]
    assert got == want
    multi.save_local(str(tmp_path / "s"))
    back = FAISS.load_local(str(tmp_path / "s"), emb, devices=[0, 0])
    assert [[(d.page_content, s) for d, s in back.similarity_search_with_score(q, k=5)] for q in queries] == want
    errors = []

    def work(t):
        try:
            r = multi.as_retriever(search_kwargs={"k": 5})
            for it in range(10):
                q = queries[(t + it) % len(queries)]
                if [d.page_content for d in r.invoke(q)] != [c for c, _ in want[(t + it) % len(queries)]]:
                    errors.append((t, it))
        except Exception as e:          # noqa: BLE001
            errors.append((t, repr(e)))
    th = [threading.Thread(target=work, args=(t,)) for t in range(8)]
    [x.start() for x in th]
    [x.join() for x in th]
    assert not errors, errors[:3]


def test_bad_arguments_are_refused():
    from rag_amd import MI355Error
    from rag_amd.index import MultiFlatIndex
    with pytest.raises(MI355Error):
        MultiFlatIndex(64, devices=[0, 99])
    idx = MultiFlatIndex(64, devices=[0, 0])
    idx.add(np.zeros((3, 64), dtype=np.float32))
    from rag_amd import _capi
    with pytest.raises(MI355Error):                          # the placement is fixed once rows exist
        _capi.check(idx._lib.mi_multi_index_set_block_rows(idx._h, 10))
    with pytest.raises(MI355Error):
        idx.reconstruct_n(2, 5)
    with pytest.raises(MI355Error):
        idx.search(np.zeros((1, 64), dtype=np.float32), 5000)
