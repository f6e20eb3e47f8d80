"""GPU: the reference's retrieval slice end to end through the drop-in classes
(advanced_rag.py:104-107, 291-305): local-directory model -> tokenise -> HIP encoder ->
FAISS.from_documents -> retriever -> ensemble, checked against the oracle at each hand-off."""
import numpy as np
import pytest

from tests.test_retrievers import CORPUS, _write_model_dir

pytestmark = pytest.mark.gpu

TEXTS = CORPUS + ["the red cat was on the flat mat", "dogs sat on the red mat", "index the row"]


def test_reference_slice_end_to_end(tmp_path, oracle_mod):
    from rag_amd import FAISS, BM25Retriever, Document, EnsembleRetriever, HuggingFaceEmbeddings
    from rag_amd._capi import BertConfig
    _write_model_dir(str(tmp_path / "m"))
    emb = HuggingFaceEmbeddings(model_name=str(tmp_path / "m"), model_kwargs={"device": 0})     # advanced_rag.py:104-107
    docs = [Document(page_content=t, metadata={"i": i}) for i, t in enumerate(TEXTS)]

    # embeddings vs the oracle on the same token ids (tolerance: 1 - cos <= 1e-3)
    E = emb.embed_documents_array(TEXTS)
    toks = emb._tokenize(TEXTS)
    ocfg = oracle_mod.BertConfig(*[getattr(emb.encoder.cfg, f) for f, _ in BertConfig._fields_])
    from rag_amd.encoder_weights import load_local_dir
    _, blob = load_local_dir(str(tmp_path / "m"))
    for i, t in enumerate(toks):
        ids = np.asarray([t], dtype=np.int32)
        Eo = oracle_mod.encode(ocfg, blob, ids, np.ones_like(ids))
        cos = float((E[i] * Eo[0]).sum() / (np.linalg.norm(E[i]) * np.linalg.norm(Eo[0])))
        assert 1.0 - cos < 1e-3, (i, cos)
    assert np.allclose(np.linalg.norm(E, axis=1), 1.0, atol=1e-3)
    assert np.allclose(np.asarray(emb.embed_query(TEXTS[2])), E[2], atol=1e-6)                   # batch-of-1 == batch

    # vector store + retriever (advanced_rag.py:295-296): ids and scores bit-exact vs the oracle on E
    vs = FAISS.from_documents(docs, emb)
    q = emb.embed_query("the cat on the mat")
    got = vs.similarity_search_with_score_by_vector(q, k=5)
    Do, Io = oracle_mod.search(E, np.asarray([q], dtype=np.float32), 5, metric=oracle_mod.METRIC_L2, dtype=oracle_mod.DTYPE_F32)
    assert [d.metadata["i"] for d, _ in got] == Io[0].tolist()
    assert np.array_equal(np.asarray([s for _, s in got], dtype=np.float32).view(np.uint32), Do[0].view(np.uint32))

    # the retriever graph (advanced_rag.py:291-305)
    bm25 = BM25Retriever.from_documents(docs)
    bm25.k = 5
    dense = vs.as_retriever(search_kwargs={"k": 5})
    ens = EnsembleRetriever(retrievers=[bm25, dense], weights=[0.6, 0.4])
    out = ((lambda x: x["question"]) | ens).invoke({"question": "the cat on the mat"})
    assert 5 <= len(out) <= 10 and len({d.page_content for d in out}) == len(out)
    assert "\n".join(str(d) for d in out).count("page_content=") == len(out)

    # persistence round trip on the real index
    vs.save_local(str(tmp_path / "store"))
    vs2 = FAISS.load_local(str(tmp_path / "store"), emb)
    got2 = vs2.similarity_search_with_score_by_vector(q, k=5)
    assert [(d.page_content, s) for d, s in got] == [(d.page_content, s) for d, s in got2]

    # reset (advanced_rag.py:435-437)
    vs.delete_all()
    assert vs.similarity_search_by_vector(q, k=3) == []
