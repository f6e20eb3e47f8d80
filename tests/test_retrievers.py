"""CPU: the callers of the hot path: BM25 + weighted reciprocal-rank fusion
(advanced_rag.py:291-301), store persistence, local-directory model loading + tokenisation
(advanced_rag.py:104-107), top-k -> rerank batch assembly.  rank-bm25 / langchain are absent from
this image: expected values are computed by hand from their published formulas (parity unpinned
by the reference, which has no fixtures for these)."""
import json
import math
import os

import numpy as np
import pytest

from rag_amd.documents import Document
from rag_amd.retrievers import BM25Okapi, BM25Retriever, EnsembleRetriever, rerank_pairs
from rag_amd.vectorstores import EUCLIDEAN_DISTANCE, InMemoryDocstore, MI355VectorStore
from tests._fakes import HashEmbeddings, OracleIndex

CORPUS = ["the cat sat on the mat", "dogs chase cats", "the mat was red", "flat index scans every row"]


def test_bm25_scores_match_the_formula():
    toks = [t.split() for t in CORPUS]
    bm = BM25Okapi(toks)
    N, avgdl = 4, sum(len(t) for t in toks) / 4
    # "mat": in docs 0 and 2 -> idf = log(4-2+.5) - log(2+.5) = 0 ; "cat": doc 0 only
    idf_cat = math.log(4 - 1 + 0.5) - math.log(1 + 0.5)
    assert bm.idf["mat"] == pytest.approx(0.0)
    s = bm.get_scores(["cat"])
    want0 = idf_cat * (1 * 2.5) / (1 + 1.5 * (1 - 0.75 + 0.75 * 6 / avgdl))
    assert s[0] == pytest.approx(want0) and s[1] == 0 and s[2] == 0 and s[3] == 0
    # "the" is in 2 of 4 docs: idf exactly 0 (not negative, so not floored) -> contributes nothing
    s2 = bm.get_scores(["the", "cat"])
    assert s2[0] == pytest.approx(want0) and s2[2] == 0 and s2[1] == 0


def test_bm25_negative_idf_is_floored_at_epsilon_average():
    toks = [["a", "b"], ["a", "c"], ["a", "d"], ["e"]]       # "a" in 3 of 4 docs -> negative idf
    bm = BM25Okapi(toks)
    raw = {w: math.log(4 - n + 0.5) - math.log(n + 0.5) for w, n in {"a": 3, "b": 1, "c": 1, "d": 1, "e": 1}.items()}
    assert raw["a"] < 0
    assert bm.idf["a"] == pytest.approx(0.25 * sum(raw.values()) / 5)


def test_bm25_retriever_k_attribute_and_order():
    docs = [Document(page_content=t, metadata={"i": i}) for i, t in enumerate(CORPUS)]
    r = BM25Retriever.from_documents(docs)
    assert r.k == 4
    r.k = 2                                                   # advanced_rag.py:292
    out = r.invoke("cat mat")
    assert len(out) == 2 and out[0].page_content == CORPUS[0] and out[0].metadata == {"i": 0}
    assert r.get_relevant_documents("cat mat") == out


class _Fixed:
    def __init__(self, texts):
        self.docs = [Document(page_content=t) for t in texts]

    def invoke(self, q, config=None):
        return self.docs


def test_weighted_reciprocal_rank_fusion_by_hand():
    a, b = _Fixed(["x", "y", "z"]), _Fixed(["y", "w"])
    ens = EnsembleRetriever(retrievers=[a, b], weights=[0.6, 0.4])          # advanced_rag.py:102,298-301
    got = [d.page_content for d in ens.invoke("q")]
    score = {"x": 0.6 / 61, "y": 0.6 / 62 + 0.4 / 61, "z": 0.6 / 63, "w": 0.4 / 62}
    assert got == sorted(score, key=score.get, reverse=True) == ["y", "x", "z", "w"]
    assert len(got) == len(set(got))                                          # deduped by page_content
    with pytest.raises(ValueError):
        EnsembleRetriever(retrievers=[a, b], weights=[1.0])


def _store(metric="l2", dtype="f32"):
    emb = HashEmbeddings(32)
    vs = MI355VectorStore(emb, OracleIndex(32, metric=metric, dtype=dtype), InMemoryDocstore(), {},
                          distance_strategy=EUCLIDEAN_DISTANCE)
    vs.add_texts(CORPUS, metadatas=[{"i": i} for i in range(len(CORPUS))])
    return vs


def test_reference_retriever_graph_runs_end_to_end():
    """advanced_rag.py:291-305: bm25 + dense -> ensemble -> `extract_question | ensemble` -> capture_context."""
    vs = _store()
    bm25 = BM25Retriever.from_documents([Document(page_content=t) for t in CORPUS])
    bm25.k = 3
    dense = vs.as_retriever(search_kwargs={"k": 3})
    ens = EnsembleRetriever(retrievers=[bm25, dense], weights=[0.6, 0.4])
    chain = (lambda x: x["question"]) | ens
    docs = chain.invoke({"question": "cat on the mat"})
    assert docs[0].page_content == CORPUS[0] and 3 <= len(docs) <= 6
    ctx = "\n".join(str(d) for d in docs)                                     # advanced_rag.py:123
    assert ctx.count("page_content=") == len(docs)


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_save_and_load_local_round_trip(tmp_path, dtype):
    vs = _store(dtype=dtype)
    vs.save_local(str(tmp_path))
    files = sorted(os.listdir(tmp_path))
    assert files == ["index.rows.npy", "index.store.json"]
    assert json.load(open(tmp_path / "index.store.json"))["format"] == "mi355rag-store-1"
    vs2 = MI355VectorStore.load_local(str(tmp_path), HashEmbeddings(32),
                                      index_factory=lambda d, metric, dtype: OracleIndex(d, metric=metric, dtype=dtype))
    q = "the red mat"
    a = vs.similarity_search_with_score(q, k=4)
    b = vs2.similarity_search_with_score(q, k=4)
    assert [(d.page_content, d.metadata, s) for d, s in a] == [(d.page_content, d.metadata, s) for d, s in b]


def _write_model_dir(path, hidden=128, layers=1, heads=4, inter=128, vocab_words=("[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]",
                     "the", "cat", "sat", "on", "mat", "dogs", "chase", "##s", "red", "was", "flat", "index", "row"), head_labels=0):
    from safetensors.numpy import save_file
    from rag_amd._capi import BertConfig
    from rag_amd.encoder_weights import random_blob
    os.makedirs(path, exist_ok=True)
    V = len(vocab_words)
    cfgd = {"vocab_size": V, "hidden_size": hidden, "num_hidden_layers": layers, "num_attention_heads": heads,
            "intermediate_size": inter, "max_position_embeddings": 64, "type_vocab_size": 2, "layer_norm_eps": 1e-12,
            "hidden_act": "gelu"}
    json.dump(cfgd, open(os.path.join(path, "config.json"), "w"))
    open(os.path.join(path, "vocab.txt"), "w").write("\n".join(vocab_words) + "\n")
    rng = np.random.default_rng(0)
    H, I = hidden, inter
    sd = {"embeddings.word_embeddings.weight": rng.standard_normal((V, H)), "embeddings.position_embeddings.weight": rng.standard_normal((64, H)),
          "embeddings.token_type_embeddings.weight": rng.standard_normal((2, H)), "embeddings.LayerNorm.weight": np.ones(H),
          "embeddings.LayerNorm.bias": np.zeros(H)}
    for l in range(layers):
        p = f"encoder.layer.{l}."
        for nm, shp in [("attention.self.query", (H, H)), ("attention.self.key", (H, H)), ("attention.self.value", (H, H)),
                        ("attention.output.dense", (H, H)), ("intermediate.dense", (I, H)), ("output.dense", (H, I))]:
            sd[p + nm + ".weight"] = rng.standard_normal(shp) * 0.02
            sd[p + nm + ".bias"] = rng.standard_normal(shp[0]) * 0.02
        for nm in ("attention.output.LayerNorm", "output.LayerNorm"):
            sd[p + nm + ".weight"] = np.ones(H)
            sd[p + nm + ".bias"] = np.zeros(H)
    # sentence-transformers checkpoints prefix the names; the loader strips it
    out = {"bert." + k: np.ascontiguousarray(v, dtype=np.float32) for k, v in sd.items()}
    if head_labels:                                   # a BertForSequenceClassification checkpoint (cross-encoder)
        sd["pooler.dense.weight"] = rng.standard_normal((H, H)) * 0.05
        sd["pooler.dense.bias"] = rng.standard_normal(H) * 0.05
        sd["classifier.weight"] = rng.standard_normal((head_labels, H)) * 0.2
        sd["classifier.bias"] = rng.standard_normal(head_labels) * 0.05
        out["bert.pooler.dense.weight"] = sd["pooler.dense.weight"].astype(np.float32)
        out["bert.pooler.dense.bias"] = sd["pooler.dense.bias"].astype(np.float32)
        out["classifier.weight"] = sd["classifier.weight"].astype(np.float32)
        out["classifier.bias"] = sd["classifier.bias"].astype(np.float32)
    save_file(out, os.path.join(path, "model.safetensors"))
    return cfgd, sd


def test_local_dir_loader_and_tokenizer(tmp_path):
    """model_name is a LOCAL directory (config.json + model.safetensors + vocab.txt); nothing is fetched."""
    from rag_amd.embeddings import _load_tokenizer
    from rag_amd.encoder_weights import load_local_dir
    cfgd, sd = _write_model_dir(str(tmp_path / "m"))
    cfg, blob = load_local_dir(str(tmp_path / "m"))
    assert cfg.hidden_size == 128 and cfg.num_hidden_layers == 1 and cfg.normalize == 1
    V, H = cfgd["vocab_size"], 128
    np.testing.assert_array_equal(blob[:V * H].reshape(V, H), sd["embeddings.word_embeddings.weight"].astype(np.float32))
    tok = _load_tokenizer(str(tmp_path / "m"), max_len=8)
    enc = tok.encode("The cats sat on the mat and then some more words follow")
    assert enc.ids[0] == 2 and enc.ids[-1] == 3 and len(enc.ids) == 8         # [CLS] ... [SEP], truncated at 8
    assert tok.encode("cats").ids == [2, 6, 12, 3]                            # WordPiece: cat + ##s, lower-cased
    with pytest.raises(FileNotFoundError):
        load_local_dir("sentence-transformers/all-MiniLM-L6-v2")              # a hub name is not a directory


def test_rerank_pair_assembly(tmp_path):
    from rag_amd.embeddings import _load_tokenizer
    _write_model_dir(str(tmp_path / "m"))
    tok = _load_tokenizer(str(tmp_path / "m"), max_len=32)
    docs = [Document(page_content="the cat sat"), Document(page_content="dogs")]
    ids, types, mask = rerank_pairs("red mat", docs, tok, max_length=32)
    assert ids.shape == types.shape == mask.shape == (2, 8)
    assert ids[0].tolist() == [2, 13, 9, 3, 5, 6, 7, 3]                       # [CLS] q [SEP] passage [SEP]
    assert types[0].tolist() == [0, 0, 0, 0, 1, 1, 1, 1]
    assert mask[1].tolist() == [1, 1, 1, 1, 1, 1, 0, 0] and ids[1, 6:].tolist() == [0, 0]


def test_rerank_pairs_keep_the_closing_token_when_cut(tmp_path):
    """A tokenizer with no truncation configured: pairs longer than max_length are cut here and still end in [SEP]."""
    from rag_amd.embeddings import _load_tokenizer
    from rag_amd.retrievers import rerank_pairs_csr
    _write_model_dir(str(tmp_path / "m"))
    tok = _load_tokenizer(str(tmp_path / "m"), max_len=32)
    tok.no_truncation()
    long_doc = Document(page_content=" ".join(["cat", "sat", "mat", "dogs"] * 10))
    ids, types, mask = rerank_pairs("red mat", [long_doc, Document(page_content="dogs")], tok, max_length=12)
    assert ids.shape == (2, 12) and ids[0, -1] == 3 and types[0, -1] == 1 and mask[0].all()      # [SEP] of segment 1
    assert ids[0, :4].tolist() == [2, 13, 9, 3]
    f_ids, f_types, off = rerank_pairs_csr([("red mat", long_doc.page_content), ("red mat", "dogs")], tok, max_length=12)
    assert off.tolist() == [0, 12, 18] and f_ids[11] == 3 and f_types[11] == 1 and f_ids[17] == 3


def test_rank_with_a_multi_label_head_needs_a_column():
    from rag_amd.retrievers import CrossEncoderReranker

    class _Enc:                                            # a test double: logits from the passage length
        def classify(self, ids, types, mask):
            n = mask.sum(axis=1).astype(np.float32)
            return np.stack([n, -n], axis=1)

    class _Tok:
        def encode(self, q, p):
            class E:
                pass
            e = E()
            e.ids = [2] + [5] * len(q.split()) + [3] + [6] * len(p.split()) + [3]
            e.type_ids = [0] * (len(q.split()) + 2) + [1] * (len(p.split()) + 1)
            return e

    docs = [Document(page_content="a b c"), Document(page_content="a"), Document(page_content="a b c d e")]
    rr = CrossEncoderReranker(encoder=_Enc(), tokenizer=_Tok(), max_length=64)
    assert rr.predict([("q", d.page_content) for d in docs]).shape == (3, 2)
    with pytest.raises(ValueError):
        rr.rank("q", docs)
    by0 = CrossEncoderReranker(encoder=_Enc(), tokenizer=_Tok(), max_length=64, score_label=0).rank("q", docs)
    by1 = CrossEncoderReranker(encoder=_Enc(), tokenizer=_Tok(), max_length=64, score_label=1).rank("q", docs, top_n=2)
    assert [d.page_content for d in by0] == ["a b c d e", "a b c", "a"]
    assert [d.page_content for d in by1] == ["a", "a b c"]


def test_reranker_slab_pipeline_with_native_pairs(tmp_path):
    """CrossEncoderReranker.predict over several slabs: the native pair tokenizer and `tokenizers` hand the encoder the
    same flat arrays (a test double that hashes them stands in for the GPU), in the pairs' order."""
    from rag_amd.embeddings import _load_tokenizer
    from rag_amd.retrievers import CrossEncoderReranker
    _write_model_dir(str(tmp_path / "m"), hidden=32, layers=1, heads=2, inter=64, head_labels=1)
    tok = _load_tokenizer(str(tmp_path / "m"), 24)

    class _Enc:
        calls = 0

        def classify_ragged(self, flat_ids, flat_types, offsets):
            _Enc.calls += 1
            w = np.arange(1, flat_ids.size + 1, dtype=np.int64) % 7 + 1
            c = np.concatenate([[0], np.cumsum(flat_ids.astype(np.int64) * w * (1 + flat_types))])
            return (c[offsets[1:]] - c[offsets[:-1]]).astype(np.float32)[:, None] - np.diff(offsets)[:, None] * 3.0

    rng = np.random.default_rng(5)
    words = ["the", "cat", "sat", "on", "mat", "dogs", "chase", "red", "Flat!", "index-row", "zzz"]
    pairs = [(" ".join(rng.choice(words, size=int(rng.integers(0, 12)))), " ".join(rng.choice(words, size=int(rng.integers(0, 40)))))
             for _ in range(300)]
    rr = CrossEncoderReranker(encoder=_Enc(), tokenizer=tok, max_length=24)
    assert rr.native_tokenizer is not None
    rr.slab_pairs = 64
    got = rr.predict(pairs)
    assert got.shape == (300,) and _Enc.calls == 5 and rr.native_tokenizer.last_native_texts == 300 - 64 * 4
    # slab by slab through the library: the double's score depends on the position inside a slab, so keep the slabs
    nat, rr.native_tokenizer = rr.native_tokenizer, None
    assert np.array_equal(rr.predict(pairs), got)
    rr.native_tokenizer = nat
    assert rr.predict([]).shape == (0,)
