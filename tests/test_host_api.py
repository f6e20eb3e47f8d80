"""CPU: the Python host layer mirrors the reference-facing interface
(advanced_rag.py:295-296, 256/303, 122-124) — exercised over an oracle-backed index."""
import numpy as np
import pytest

from rag_amd.documents import Document
from rag_amd.vectorstores import (EUCLIDEAN_DISTANCE, MAX_INNER_PRODUCT, InMemoryDocstore, MI355VectorStore,
                                  VectorStoreRetriever)
from tests._fakes import HashEmbeddings, OracleIndex

TEXTS = ["the cat sat on the mat", "dogs chase cats", "retrieval augmented generation with flat search",
         "matrix cores accelerate dense retrieval", "a flat index scans every row", "the mat was red",
         "gradio renders the user interface"]


def _store(metric="l2", texts=TEXTS, metas=None):
    emb = HashEmbeddings(32)
    idx = OracleIndex(32, metric=metric)
    strat = MAX_INNER_PRODUCT if metric == "ip" else EUCLIDEAN_DISTANCE
    vs = MI355VectorStore(emb, idx, InMemoryDocstore(), {}, distance_strategy=strat)
    vs.add_texts(texts, metadatas=metas)
    return vs


def test_document_str_matches_langchain_form():
    d = Document(page_content="hello 'x'", metadata={"source": "u"})
    assert str(d) == "page_content=\"hello 'x'\" metadata={'source': 'u'}"
    assert "\n".join(str(x) for x in [d, d]).count("page_content=") == 2      # advanced_rag.py:123


def test_similarity_search_best_first_and_default_k():
    vs = _store()
    docs = vs.similarity_search("flat index search")
    assert len(docs) == 4                                                     # LangChain default k=4
    assert docs[0].page_content in ("a flat index scans every row", "retrieval augmented generation with flat search")
    with_scores = vs.similarity_search_with_score("flat index search", k=5)
    s = [x[1] for x in with_scores]
    assert s == sorted(s) and len(s) == 5                                     # L2: ascending distance
    vs_ip = _store("ip")
    s = [x[1] for x in vs_ip.similarity_search_with_score("flat index search", k=5)]
    assert s == sorted(s, reverse=True)


def test_k_larger_than_store_skips_minus_one_labels():
    vs = _store(texts=TEXTS[:3])
    assert len(vs.similarity_search("cat", k=10)) == 3


def test_retriever_invoke_and_pipe():
    vs = _store()
    r = vs.as_retriever(search_kwargs={"k": 5})                               # advanced_rag.py:296
    assert isinstance(r, VectorStoreRetriever)
    docs = r.invoke("cats")
    assert len(docs) == 5 and all(isinstance(d, Document) for d in docs)
    assert r.get_relevant_documents("cats") == docs
    chain = (lambda x: x["question"]) | r                                     # advanced_rag.py:256
    assert chain.invoke({"question": "cats"}) == docs
    chain2 = r | (lambda ds: "\n".join(str(d) for d in ds))                   # advanced_rag.py:122-124
    assert chain2.invoke("cats").count("page_content=") == 5
    assert r.batch(["cats", "flat"])[0] == docs


def test_filter_and_metadata_roundtrip():
    metas = [{"source": f"s{i % 2}", "i": i} for i in range(len(TEXTS))]
    vs = _store(metas=metas)
    docs = vs.similarity_search("the mat", k=3, filter={"source": "s1"}, fetch_k=7)
    assert docs and all(d.metadata["source"] == "s1" for d in docs)
    docs = vs.similarity_search("the mat", k=3, filter=lambda m: m["i"] < 2, fetch_k=7)
    assert all(d.metadata["i"] < 2 for d in docs)


def test_add_validation_errors():
    vs = _store()
    with pytest.raises(ValueError):
        vs.add_texts(["a", "b"], ids=["same", "same"])
    with pytest.raises(ValueError):
        vs.add_texts(["a", "b"], metadatas=[{}])
    first_id = vs.index_to_docstore_id[0]
    with pytest.raises(ValueError):
        vs.add_texts(["again"], ids=[first_id])


def test_from_embeddings_and_by_vector():
    emb = HashEmbeddings(32)
    pairs = [(t, emb.embed_query(t)) for t in TEXTS]
    idx = OracleIndex(32)
    vs = MI355VectorStore(emb, idx, InMemoryDocstore(), {})
    vs.add_embeddings(pairs)
    got = vs.similarity_search_by_vector(emb.embed_query("dogs chase cats"), k=1)
    assert got[0].page_content == "dogs chase cats"
    D, I = vs.search_batch(np.asarray([emb.embed_query(t) for t in TEXTS[:2]], dtype=np.float32), 2)
    assert D.shape == (2, 2) and list(I[:, 0]) == [0, 1]


def test_relevance_scores_and_threshold_retriever():
    vs = _store()
    out = vs.similarity_search_with_relevance_scores("dogs chase cats", k=3)
    assert abs(out[0][1] - 1.0) < 1e-6 and out[0][1] >= out[1][1] >= out[2][1]
    r = vs.as_retriever(search_type="similarity_score_threshold", search_kwargs={"k": 7, "score_threshold": 0.9})
    assert [d.page_content for d in r.invoke("dogs chase cats")] == ["dogs chase cats"]
    with pytest.raises(ValueError):
        vs.as_retriever(search_type="mmr")


def test_delete_all_resets_the_store():
    vs = _store()
    vs.delete_all()
    assert vs.index.ntotal == 0 and vs.similarity_search("cats") == []
