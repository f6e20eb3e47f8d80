"""GPU: bulk embedding through the boundary the application calls (advanced_rag.py:295, FAISS.from_documents ->
embed_documents): ragged token-budget batching inside the library (mi_encoder_encode_ragged), the kernel family a
bulk call takes, and thread safety of one encoder shared by many host threads.  Bar: 1 - cosine <= 1e-3 against
the CPU oracle on the same token ids (BASELINE.json north_star)."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TOL = 1e-3


def _cos_err(a, b):
    return 1.0 - (a * b).sum(1) / np.maximum(np.linalg.norm(a, axis=1) * np.linalg.norm(b, axis=1), 1e-30)


def _oracle_rows(oracle, ocfg, blob, flat, offsets, rows, max_len=0):
    out = []
    for i in rows:
        ids = flat[offsets[i]:offsets[i + 1]]
        if max_len:
            ids = ids[:max_len]
        out.append(oracle.encode(ocfg, blob, ids[None, :].astype(np.int32), np.ones((1, ids.size), dtype=np.int32))[0])
    return np.stack(out)


def test_ragged_entry_point_matches_oracle_in_input_order(oracle_mod):
    """Lengths 0 .. beyond max_len in shuffled order: results come back in input order, an empty sequence is the
    zero vector, long ones are truncated; small budget via max_batch so several passes and both staging sets run."""
    from rag_amd._capi import BertConfig
    from rag_amd.embeddings import Encoder
    from rag_amd.encoder_weights import random_blob
    vals = (500, 128, 2, 4, 256, 64, 2, 1e-12, 1)
    cfg, ocfg = BertConfig(*vals), oracle_mod.BertConfig(*vals)
    blob = random_blob(cfg, seed=4, init_range=0.1)
    rng = np.random.default_rng(3)
    n = 57
    lens = rng.integers(1, 80, size=n)
    lens[5] = 0
    lens[20] = 64
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    flat = rng.integers(0, 500, size=int(offsets[-1])).astype(np.int32)
    enc = Encoder(cfg, blob)
    for max_batch in (0, 7):
        E = enc.encode_ragged(flat, offsets, max_len=60, max_batch=max_batch)
        assert E.shape == (n, 128) and np.all(E[5] == 0)
        rows = [i for i in range(n) if i != 5]
        Eo = _oracle_rows(oracle_mod, ocfg, blob, flat, offsets, rows, max_len=60)
        assert np.all(_cos_err(E[rows], Eo) < TOL)
    st = enc.stats()
    assert st["sequences"] == 2 * n and st["latency_passes"] >= 1 + 9
    # the padded [B][L] entry point gives the same vectors (padding changes nothing)
    L = 60
    ids = np.zeros((n, L), dtype=np.int32)
    mask = np.zeros((n, L), dtype=np.int32)
    for i in range(n):
        m = min(int(lens[i]), L)
        ids[i, :m] = flat[offsets[i]:offsets[i] + m]
        mask[i, :m] = 1
    Ep = enc.encode_ids(ids, mask)
    assert np.all(_cos_err(E[rows], Ep[rows]) < 1e-5)
    from rag_amd import MI355Error
    bad = flat.copy()
    bad[3] = 500
    with pytest.raises(MI355Error):
        enc.encode_ragged(bad, offsets)


def test_from_documents_of_2000_chunks_runs_the_bulk_kernels(tmp_path, oracle_mod):
    """FAISS.from_documents on 2,000 chunks (advanced_rag.py:295): the embed call is cut by the library's token
    budget, so every pass has >= 12,288 tokens and runs the fused bulk kernels (mi_encoder_stats), and the stored
    rows equal the per-chunk oracle embeddings within 1e-3."""
    from rag_amd import FAISS, Document, HuggingFaceEmbeddings
    from rag_amd._capi import BertConfig
    from rag_amd.encoder_weights import load_local_dir
    from tests.test_retrievers import _write_model_dir
    words = ["the", "cat", "sat", "on", "mat", "dogs", "chase", "red", "was", "flat", "index", "row"]
    _write_model_dir(str(tmp_path / "m"), hidden=384, layers=2, heads=12, inter=1536)
    emb = HuggingFaceEmbeddings(model_name=str(tmp_path / "m"), model_kwargs={"device": 0}, max_seq_length=64)
    rng = np.random.default_rng(1)
    texts = [" ".join(rng.choice(words, size=int(rng.integers(8, 61)))) for _ in range(2000)]
    vs = FAISS.from_documents([Document(page_content=t, metadata={"i": i}) for i, t in enumerate(texts)], emb)
    st = emb.encoder.stats()
    assert st["sequences"] == 2000 and st["bulk_passes"] >= 1 and st["latency_passes"] == 0, st
    assert st["tokens"] >= 12288 * st["bulk_passes"]
    E = vs.index.reconstruct_n(0, 2000)
    toks = emb._tokenize(texts)
    flat, offsets = emb._to_csr(toks)
    _, blob = load_local_dir(str(tmp_path / "m"))
    ocfg = oracle_mod.BertConfig(*[getattr(emb.encoder.cfg, f) for f, _ in BertConfig._fields_])
    rows = list(range(0, 2000, 8))
    Eo = _oracle_rows(oracle_mod, ocfg, blob, flat, offsets, rows)
    err = _cos_err(E[rows], Eo)
    assert np.all(err < TOL), err.max()
    # a caller that sets batch_size gets its cap (sentence-transformers' semantics): 32 rows per pass -> latency kernels
    emb32 = HuggingFaceEmbeddings(model_name=str(tmp_path / "m"), model_kwargs={"device": 0}, encode_kwargs={"batch_size": 32},
                                  max_seq_length=64)
    E32 = emb32.embed_documents_array(texts[:100])
    assert emb32.encoder.stats()["latency_passes"] == 4 and np.all(_cos_err(E32, E[:100]) < 1e-4)


def test_one_encoder_shared_by_eight_threads(oracle_mod):
    """LangChain's RunnableParallel workers and Gradio handlers call embed_query / embed_documents on ONE
    embeddings object (advanced_rag.py:302-305, 545-555).  Eight threads encode different batches through the
    host-pointer and the ragged entry points at once; every result equals the single-threaded one bit for bit."""
    from rag_amd._capi import BertConfig
    from rag_amd.embeddings import Encoder
    from rag_amd.encoder_weights import random_blob
    cfg = BertConfig(1000, 384, 2, 12, 1536, 512, 2, 1e-12, 1)
    enc = Encoder(cfg, random_blob(cfg, seed=8, init_range=0.05))
    rng = np.random.default_rng(5)
    jobs = []
    for t in range(8):
        B, L = int(rng.integers(1, 9)), int(rng.integers(4, 90))
        ids = rng.integers(0, 1000, size=(B, L)).astype(np.int32)
        mask = np.ones((B, L), dtype=np.int32)
        mask[0, L // 2:] = 0
        lens = rng.integers(1, 70, size=B + 3)
        off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        flat = rng.integers(0, 1000, size=int(off[-1])).astype(np.int32)
        jobs.append((ids, mask, flat, off))
    want = [(enc.encode_ids(i, m), enc.encode_ragged(f, o)) for i, m, f, o in jobs]
    errors = []

    def work(t):
        try:
            ids, mask, flat, off = jobs[t]
            for it in range(12):
                a = enc.encode_ids(ids, mask)
                b = enc.encode_ragged(flat, off)
                if not (np.array_equal(a, want[t][0]) and np.array_equal(b, want[t][1])):
                    errors.append((t, it))
        except Exception as e:          # noqa: BLE001
            errors.append((t, repr(e)))

    th = [threading.Thread(target=work, args=(t,)) for t in range(8)]
    [x.start() for x in th]
    [x.join() for x in th]
    assert not errors, errors[:3]


_PADDED_CHILD = r"""
import sys
sys.path.insert(0, %r)
import numpy as np
import oracle
from rag_amd._capi import BertConfig
from rag_amd.embeddings import Encoder
from rag_amd.encoder_weights import random_blob
vals = (500, 384, 2, 12, 1536, 256, 2, 1e-12, 1)
cfg, ocfg = BertConfig(*vals), oracle.BertConfig(*vals)
blob = random_blob(cfg, seed=9, init_range=0.05)
rng = np.random.default_rng(2)
lens = rng.integers(0, 70, size=40)
off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
flat = rng.integers(0, 500, size=int(off[-1])).astype(np.int32)
enc = Encoder(cfg, blob)
E = enc.encode_ragged(flat, off)
worst = 0.0
for i in range(40):
    if lens[i] == 0:
        assert np.all(E[i] == 0)
        continue
    ids = flat[off[i]:off[i + 1]][None, :]
    Eo = oracle.encode(ocfg, blob, ids, np.ones_like(ids))[0]
    worst = max(worst, 1.0 - float((E[i] * Eo).sum() / (np.linalg.norm(E[i]) * np.linalg.norm(Eo))))
print("TOKENS", enc.stats()["tokens"], "REAL", int(np.maximum(lens, 1).sum()), "WORST", worst)
"""


@pytest.mark.parametrize("varlen", ["0", "1"])
def test_padded_and_packed_passes_in_child_processes(varlen):
    """MI_ENC_VARLEN (read once per process): "0" pads a pass to its longest sequence, the default packs its tokens back to
    back.  Both against the oracle, empty sequences included; the packed form computes exactly the real tokens."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = dict(os.environ)
    e["MI_ENC_VARLEN"] = varlen
    out = subprocess.run([sys.executable, "-c", _PADDED_CHILD % root], env=e, capture_output=True, text=True, timeout=170)
    assert out.returncode == 0, out.stderr[-2000:]
    f = out.stdout.strip().split()
    tokens, real, worst = int(f[1]), int(f[3]), float(f[5])
    assert worst < TOL, worst
    assert (tokens == real) if varlen == "1" else (tokens > real)


def test_passes_on_two_streams_do_not_share_the_workspace_at_once(oracle_mod):
    """every pass of an encoder uses ONE activation workspace.  encode_ids_torch enqueues on PyTorch's
    stream and returns with its kernels still running; a host-pointer call right behind it runs on the ctx's own
    (non-blocking) stream.  The second pass must wait for the first (the library chains them through the event it
    records behind every pass); both results against the oracle, several rounds, sizes that re-grow the workspace."""
    import torch
    from rag_amd._capi import BertConfig
    from rag_amd.embeddings import Encoder
    from rag_amd.encoder_weights import random_blob
    vals = (1000, 384, 2, 12, 1536, 512, 2, 1e-12, 1)
    cfg, ocfg = BertConfig(*vals), oracle_mod.BertConfig(*vals)
    blob = random_blob(cfg, seed=12, init_range=0.05)
    enc = Encoder(cfg, blob)
    rng = np.random.default_rng(8)
    big_ids = rng.integers(0, 1000, size=(192, 128)).astype(np.int32)       # 24,576 tokens: the bulk kernels, ~1 ms on the GPU
    big_mask = np.ones_like(big_ids)
    small_ids = rng.integers(0, 1000, size=(3, 40)).astype(np.int32)
    small_mask = np.ones_like(small_ids)
    rows = [0, 95, 191]
    want_big = oracle_mod.encode(ocfg, blob, big_ids[rows], big_mask[rows])
    want_small = oracle_mod.encode(ocfg, blob, small_ids, small_mask)
    ids_t, mask_t = torch.from_numpy(big_ids).cuda(), torch.from_numpy(big_mask).cuda()
    torch.cuda.synchronize()
    for rnd in range(6):
        if rnd == 3:                                          # a larger pass in between re-grows the workspace
            enc.encode_ids(np.tile(big_ids, (2, 1)), np.tile(big_mask, (2, 1)))
        Eb = enc.encode_ids_torch(ids_t, mask_t)              # no sync: still running when the next call starts
        Es = enc.encode_ids(small_ids, small_mask)            # ctx stream
        Eb2 = enc.encode_ids_torch(ids_t, mask_t)             # and back onto PyTorch's stream
        torch.cuda.synchronize()
        assert np.all(_cos_err(Es, want_small) < TOL), rnd
        assert np.all(_cos_err(Eb.cpu().numpy()[rows], want_big) < TOL), rnd
        assert torch.equal(Eb, Eb2), rnd


def test_slab_pipeline_equals_the_one_shot_call(tmp_path):
    """embed_documents_array on more texts than one slab: the worker thread tokenises slab i + 1 while slab i is on the GPU;
    same rows in the same order as the one-shot call (a slab boundary changes which sequences share a pass and
    so which kernel family / GEMM tiling the pass takes -- bf16-level differences, far inside the 1e-3 bar -- never
    which text a row belongs to)."""
    from rag_amd import HuggingFaceEmbeddings
    from tests.test_retrievers import _write_model_dir
    words = ["the", "cat", "sat", "on", "mat", "dogs", "chase", "red", "was", "flat", "index", "row"]
    _write_model_dir(str(tmp_path / "m"), hidden=384, layers=2, heads=12, inter=1536)
    rng = np.random.default_rng(4)
    texts = [" ".join(rng.choice(words, size=int(rng.integers(1, 61)))) for _ in range(1500)]
    texts[17] = "line one\nline two"
    one = HuggingFaceEmbeddings(model_name=str(tmp_path / "m"), max_seq_length=64)
    E1 = one.embed_documents_array(texts)
    assert one.last_timing["slabs"] == 1
    slabs = HuggingFaceEmbeddings(model_name=str(tmp_path / "m"), max_seq_length=64, slab_texts=256)
    E2 = slabs.embed_documents_array(texts)
    assert slabs.last_timing["slabs"] == 6 and slabs.last_timing["tokens"] == one.last_timing["tokens"]
    assert E2.shape == E1.shape and np.all(_cos_err(E1, E2) < 5e-5), _cos_err(E1, E2).max()
    # several encoders in this process (one per listed device; device 0 twice on a one-GPU box): same rows, same order
    multi = HuggingFaceEmbeddings(model_name=str(tmp_path / "m"), max_seq_length=64, devices=[0, 0, 0], slab_texts=700)
    E3 = multi.embed_documents_array(texts)
    assert E3.shape == E1.shape and np.all(_cos_err(E1, E3) < 5e-5), _cos_err(E1, E3).max()
    assert sum(e.stats()["sequences"] for e in multi.encoders) == 1500 and all(e.stats()["sequences"] > 0 for e in multi.encoders)


def test_two_lanes_equal_one_lane_and_stay_ordered(oracle_mod, monkeypatch):
    """A call of several passes alternates them between the encoder's two lanes (own stream, workspace and staging each).
    Same bits as the one-lane run (MI_ENC_LANES=1, read per call) for the device-pointer and the host-pointer entry
    points, rows against the oracle, and the call stays one piece of work on the caller's stream: a consumer enqueued on
    PyTorch's stream right behind it -- no synchronisation -- reads finished rows of BOTH lanes."""
    import torch
    from rag_amd._capi import BertConfig
    from rag_amd.embeddings import Encoder
    from rag_amd.encoder_weights import random_blob
    vals = (1000, 384, 2, 12, 1536, 512, 2, 1e-12, 1)
    cfg, ocfg = BertConfig(*vals), oracle_mod.BertConfig(*vals)
    blob = random_blob(cfg, seed=21, init_range=0.05)
    enc = Encoder(cfg, blob)
    monkeypatch.setenv("MI_ENC_PASS_TOKENS", "16384")       # read once per process: only effective if nothing encoded before
    rng = np.random.default_rng(5)
    B, L = 5 * 512 + 100, 128                               # several passes whatever the pass size, a ragged last one
    ids = rng.integers(0, 1000, size=(B, L)).astype(np.int32)
    mask = np.ones_like(ids)
    mask[7, 50:] = 0
    rows = [0, 7, 511, 512, 1300, B - 1]
    want = oracle_mod.encode(ocfg, blob, ids[rows], mask[rows])
    ids_t, mask_t = torch.from_numpy(ids).cuda(), torch.from_numpy(mask).cuda()
    torch.cuda.synchronize()
    passes0 = sum(enc.stats()[k] for k in ("bulk_passes", "latency_passes"))
    got, sums = {}, {}
    for lanes in ("2", "1"):
        monkeypatch.setenv("MI_ENC_LANES", lanes)
        E = enc.encode_ids_torch(ids_t, mask_t)
        sums[lanes] = E.sum(dim=1)                            # consumer on PyTorch's stream, no sync in between
        torch.cuda.synchronize()
        got[lanes] = E.cpu().numpy()
        assert np.allclose(sums[lanes].cpu().numpy(), got[lanes].sum(axis=1), rtol=0, atol=1e-4)
        assert np.all(_cos_err(got[lanes][rows], want) < TOL)
        Eh = enc.encode_ids(ids[:1200], mask[:1200])        # host pointers: the padded entry point, lanes inside
        assert np.array_equal(Eh, got[lanes][:1200])
    assert np.array_equal(got["2"], got["1"])
    assert sum(enc.stats()[k] for k in ("bulk_passes", "latency_passes")) - passes0 >= 8       # it really was several passes per call
    lens = rng.integers(1, L + 1, size=3000)
    flat = np.concatenate([ids[i % B, :n] for i, n in enumerate(lens)]).astype(np.int32)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    R = {}
    for lanes in ("2", "1"):
        monkeypatch.setenv("MI_ENC_LANES", lanes)
        R[lanes] = enc.encode_ragged(flat, off, max_len=L)
    assert np.array_equal(R["2"], R["1"])


def test_fused_qkv_attention_equals_the_two_kernels_and_the_oracle(oracle_mod, monkeypatch):
    """qkv_attn_kernel (encoder_qkv_attn.hip: QKV projection + attention in one kernel, bulk passes of sequences up to 256
    tokens) against the two kernels it replaces (MI_QKV_FUSED=0, read per call) and against the oracle: sequence lengths on
    every edge of its tiling (1, 31..33, 63..65, 127..129, 255, 256), a batch that is no multiple of 8 (the XCD-aware grid is
    rounded up), packed sequences and padded ones with a key mask -- masked keys in the middle, and a fully masked row, which
    must still come out uniform over its padded length as the reference's finfo.min arithmetic gives it.  Longer sequences
    (257 ..) keep taking the two kernels."""
    from rag_amd._capi import BertConfig
    from rag_amd.embeddings import Encoder
    from rag_amd.encoder_weights import random_blob
    vals = (1000, 384, 2, 12, 1536, 512, 2, 1e-12, 1)
    cfg, ocfg = BertConfig(*vals), oracle_mod.BertConfig(*vals)
    blob = random_blob(cfg, seed=31, init_range=0.05)
    enc = Encoder(cfg, blob)
    monkeypatch.setenv("MI_ENC_LATENCY_TOKENS", "0")        # every pass takes the bulk kernels
    monkeypatch.setenv("MI_QKV_FUSED_MIN_MEAN", "0")        # ... and the fused kernel whatever the mean length
    rng = np.random.default_rng(9)
    edge = [1, 2, 31, 32, 33, 63, 64, 65, 96, 127, 128, 129, 191, 192, 193, 255, 256]
    lens = np.array(edge + list(rng.integers(1, 257, size=70 - len(edge))))          # 70 sequences: not a multiple of 8
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    flat = rng.integers(0, 1000, size=int(off[-1])).astype(np.int32)

    def both(fn):
        out = {}
        for m in ("1", "0"):
            monkeypatch.setenv("MI_QKV_FUSED", m)
            out[m] = fn()
        return out["1"], out["0"]

    # packed sequences, one pass per length group and all in one
    for max_batch in (0, 9):
        Ef, Eu = both(lambda: enc.encode_ragged(flat, off, max_len=256, max_batch=max_batch))
        assert np.all(_cos_err(Ef, Eu) < 2e-5), float(_cos_err(Ef, Eu).max())
        rows = list(range(len(edge))) + [40, 69]
        want = _oracle_rows(oracle_mod, ocfg, blob, flat, off, rows)
        assert np.all(_cos_err(Ef[rows], want) < TOL)
    # padded with a key mask
    B, L = 21, 200
    ids = rng.integers(0, 1000, size=(B, L)).astype(np.int32)
    mask = np.ones((B, L), dtype=np.int32)
    mask[1, 150:] = 0
    mask[2, 40:90] = 0                                        # a hole in the middle
    mask[3, :] = 0
    mask[3, 0] = 1                                            # one key
    mask[4, 1:] = 0
    mask[5, :] = 0                                            # nothing to attend to, nothing to pool: the zero vector either way
    Ef, Eu = both(lambda: enc.encode_ids(ids, mask))
    live = [i for i in range(B) if i != 5]
    assert np.all(_cos_err(Ef[live], Eu[live]) < 2e-5)
    assert np.array_equal(Ef[5], Eu[5])
    want = oracle_mod.encode(ocfg, blob, ids[:6], mask[:6])
    assert np.all(_cos_err(Ef[:5], want[:5]) < TOL)
    # the hidden states of a fully masked sequence (attention uniform over its padded length) through the classifier head,
    # which reads token 0 whatever the mask says
    from rag_amd.encoder_weights import random_head
    enc.set_head(*random_head(cfg, num_labels=2, seed=3, init_range=0.1))
    types = np.zeros_like(ids)
    Lf, Lu = both(lambda: enc.classify(ids[:8], types[:8], mask[:8]))
    assert np.allclose(Lf, Lu, rtol=0, atol=2e-2), float(np.abs(Lf - Lu).max())
    # beyond 256 tokens the fused kernel steps aside: same bits with the switch either way
    lens2 = np.array([300, 257, 512, 64])
    off2 = np.concatenate([[0], np.cumsum(lens2)]).astype(np.int64)
    flat2 = rng.integers(0, 1000, size=int(off2[-1])).astype(np.int32)
    Ef, Eu = both(lambda: enc.encode_ragged(flat2, off2, max_len=512))
    assert np.array_equal(Ef, Eu)
