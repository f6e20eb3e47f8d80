"""The callers either side of the dense hot path (SURVEY.md §8f "next" #1), so the reference's
whole retriever graph runs against the drop-in:

    self.bm25_retriever = BM25Retriever.from_documents(self.split_data)          advanced_rag.py:291
    self.bm25_retriever.k = self.top_k                                           advanced_rag.py:292
    self.ensemble_retriever = EnsembleRetriever(retrievers=[bm25, faiss],
                                                weights=[bm25_weight, faiss_weight])   advanced_rag.py:298-301

Host-side Python on purpose: tens of documents per query, nothing data-parallel.  Algorithms
restate langchain==0.1.6 EnsembleRetriever (weighted reciprocal-rank fusion, c = 60, dedupe by
page_content) and rank-bm25==0.2.2 BM25Okapi (requirements.txt:2,15); neither package is in
this image, so parity is pinned by hand-computed cases in tests/test_retrievers.py.
"""
from __future__ import annotations

import math
import os
from typing import Any, Callable, Dict, Iterable, List, Optional, Sequence

from .documents import Document
from .vectorstores import _runnable_lambda


def default_preprocessing_func(text: str) -> List[str]:
    return text.split()


class _Runnable:
    """invoke / get_relevant_documents / `|` composition shared by the retrievers."""

    def get_relevant_documents(self, query: str, **kwargs: Any) -> List[Document]:
        raise NotImplementedError

    def invoke(self, input: str, config: Any = None, **kwargs: Any) -> List[Document]:
        return self.get_relevant_documents(input)

    async def ainvoke(self, input: str, config: Any = None, **kwargs: Any) -> List[Document]:
        return self.get_relevant_documents(input)

    __call__ = invoke

    def __or__(self, other: Any):
        return _runnable_lambda(self.invoke) | other

    def __ror__(self, other: Any):
        left = other if hasattr(other, "invoke") else _runnable_lambda(other)
        return left | _runnable_lambda(self.invoke)


class BM25Okapi:
    """rank_bm25.BM25Okapi: k1 = 1.5, b = 0.75, epsilon = 0.25."""

    def __init__(self, corpus: Sequence[Sequence[str]], k1: float = 1.5, b: float = 0.75, epsilon: float = 0.25):
        self.k1, self.b, self.epsilon = k1, b, epsilon
        self.corpus_size = len(corpus)
        self.doc_len = [len(d) for d in corpus]
        self.avgdl = (sum(self.doc_len) / self.corpus_size) if self.corpus_size else 0.0
        self.doc_freqs: List[Dict[str, int]] = []
        nd: Dict[str, int] = {}
        for doc in corpus:
            f: Dict[str, int] = {}
            for w in doc:
                f[w] = f.get(w, 0) + 1
            self.doc_freqs.append(f)
            for w in f:
                nd[w] = nd.get(w, 0) + 1
        # idf = log(N - n + 0.5) - log(n + 0.5); negative values are floored at epsilon * average idf
        self.idf: Dict[str, float] = {}
        idf_sum, negative = 0.0, []
        for w, n in nd.items():
            v = math.log(self.corpus_size - n + 0.5) - math.log(n + 0.5)
            self.idf[w] = v
            idf_sum += v
            if v < 0:
                negative.append(w)
        self.average_idf = idf_sum / len(self.idf) if self.idf else 0.0
        eps = self.epsilon * self.average_idf
        for w in negative:
            self.idf[w] = eps

    def get_scores(self, query: Sequence[str]) -> List[float]:
        scores = [0.0] * self.corpus_size
        for q in query:
            idf = self.idf.get(q) or 0.0
            for i, f in enumerate(self.doc_freqs):
                qf = f.get(q, 0)
                denom = qf + self.k1 * (1 - self.b + self.b * self.doc_len[i] / self.avgdl) if self.avgdl else 1.0
                scores[i] += idf * (qf * (self.k1 + 1) / denom)
        return scores

    def get_top_n(self, query: Sequence[str], documents: Sequence[Any], n: int = 5) -> List[Any]:
        scores = self.get_scores(query)
        # numpy.argsort(scores)[::-1][:n] in rank_bm25: descending score, ties by HIGHER index first
        order = sorted(range(len(scores)), key=lambda i: (scores[i], i), reverse=True)[:n]
        return [documents[i] for i in order]


class BM25Retriever(_Runnable):
    """langchain_community.retrievers.BM25Retriever: `.k` is a plain attribute (advanced_rag.py:292)."""

    def __init__(self, vectorizer: BM25Okapi, docs: List[Document], k: int = 4,
                 preprocess_func: Callable[[str], List[str]] = default_preprocessing_func):
        self.vectorizer = vectorizer
        self.docs = docs
        self.k = k
        self.preprocess_func = preprocess_func

    @classmethod
    def from_texts(cls, texts: Iterable[str], metadatas: Optional[Iterable[dict]] = None,
                   bm25_params: Optional[Dict[str, Any]] = None,
                   preprocess_func: Callable[[str], List[str]] = default_preprocessing_func, **kwargs: Any):
        texts = list(texts)
        metas = list(metadatas) if metadatas is not None else [{} for _ in texts]
        vec = BM25Okapi([preprocess_func(t) for t in texts], **(bm25_params or {}))
        docs = [Document(page_content=t, metadata=m) for t, m in zip(texts, metas)]
        return cls(vec, docs, preprocess_func=preprocess_func, **kwargs)

    @classmethod
    def from_documents(cls, documents: Iterable[Document], **kwargs: Any):
        documents = list(documents)
        return cls.from_texts([d.page_content for d in documents], [d.metadata for d in documents], **kwargs)

    def get_relevant_documents(self, query: str, **kwargs: Any) -> List[Document]:
        return self.vectorizer.get_top_n(self.preprocess_func(query), self.docs, n=self.k)


class EnsembleRetriever(_Runnable):
    """Weighted reciprocal-rank fusion of several retrievers' lists (langchain 0.1.6):
    score(doc) = sum_i weight_i / (rank_i(doc) + c), documents identified by page_content,
    best first.  Ties keep first-seen order (LangChain iterates a set there, i.e. its tie
    order is unspecified)."""

    def __init__(self, retrievers: List[Any], weights: Optional[List[float]] = None, c: int = 60):
        self.retrievers = list(retrievers)
        n = len(self.retrievers)
        self.weights = list(weights) if weights is not None else ([1.0 / n] * n if n else [])
        if len(self.weights) != n:
            raise ValueError("Number of rank lists must be equal to the number of weights.")
        self.c = c

    def rank_fusion(self, query: str) -> List[Document]:
        lists = []
        for r in self.retrievers:
            docs = r.invoke(query) if hasattr(r, "invoke") else r.get_relevant_documents(query)
            lists.append([d if isinstance(d, Document) else Document(page_content=str(d)) for d in docs])
        return self.weighted_reciprocal_rank(lists)

    def weighted_reciprocal_rank(self, doc_lists: List[List[Document]]) -> List[Document]:
        if len(doc_lists) != len(self.weights):
            raise ValueError("Number of rank lists must be equal to the number of weights.")
        score: Dict[str, float] = {}
        for docs, w in zip(doc_lists, self.weights):
            for rank, d in enumerate(docs, start=1):
                score[d.page_content] = score.get(d.page_content, 0.0) + w * (1.0 / (rank + self.c))
        by_content = {d.page_content: d for docs in doc_lists for d in docs}      # later lists win, as in LangChain
        order = sorted(score, key=lambda key: score[key], reverse=True)           # stable: ties keep first-seen order
        return [by_content[key] for key in order]

    def get_relevant_documents(self, query: str, **kwargs: Any) -> List[Document]:
        return self.rank_fusion(query)


def rerank_pairs(query: str, documents: Sequence[Document], tokenizer: Any, max_length: int = 512):
    """Top-k -> cross-encoder batch assembly (BASELINE config 5; README.md:27 names a reranker the
    reference code lacks): tokenises (query, passage) pairs into padded int32 arrays
    (ids, token_type_ids, mask), longest pair first is NOT applied: order follows `documents`."""
    import numpy as np
    encs = [tokenizer.encode(query, d.page_content) for d in documents]
    L = min(max((len(e.ids) for e in encs), default=1), max_length)
    n = len(encs)
    ids = np.zeros((n, L), dtype=np.int32)
    types = np.zeros((n, L), dtype=np.int32)
    mask = np.zeros((n, L), dtype=np.int32)
    for i, e in enumerate(encs):
        ei, et = _cut_pair(e.ids, e.type_ids, L)
        m = len(ei)
        ids[i, :m] = ei
        types[i, :m] = et
        mask[i, :m] = 1
    return ids, types, mask


def _cut_pair(ids, type_ids, max_length: int):
    """A pair longer than max_length that the tokenizer did not truncate itself: cut the passage but keep the closing
    token ([SEP]) and its segment id, as the tokenizer's own truncation does."""
    if len(ids) <= max_length:
        return list(ids), list(type_ids)
    if max_length < 2:
        return list(ids[:max_length]), list(type_ids[:max_length])
    return list(ids[:max_length - 1]) + [ids[-1]], list(type_ids[:max_length - 1]) + [type_ids[-1]]


def rerank_pairs_csr(pairs: Sequence[Sequence[str]], tokenizer: Any, max_length: int = 512):
    """(query, passage) pairs -> one flat int32 token array, its token types and offsets [n + 1] (no padding): the input
    of mi_encoder_classify_ragged.  Truncation to max_length is the tokenizer's (it keeps the closing [SEP]); a
    tokenizer without truncation configured is cut here, keeping the closing token too (_cut_pair)."""
    import numpy as np
    ids, types, offsets = [], [], [0]
    batch = getattr(tokenizer, "encode_batch", None)                 # (`tokenizers`: all pairs in one call, on all cores)
    encs = batch([(q, passage) for q, passage in pairs]) if batch else (tokenizer.encode(q, passage) for q, passage in pairs)
    for e in encs:
        ei, et = _cut_pair(e.ids, e.type_ids, max_length)
        ids.extend(ei)
        types.extend(et)
        offsets.append(offsets[-1] + len(ei))
    return np.asarray(ids, dtype=np.int32), np.asarray(types, dtype=np.int32), np.asarray(offsets, dtype=np.int64)


class CrossEncoderReranker:
    """Scores (query, passage) pairs with a BERT cross-encoder on the MI355X and re-orders the candidates: the
    consumer of BASELINE config 5's top-100 lists (README.md:27 and requirements.txt:24 name a reranker; the
    reference code has none, so the surface follows sentence-transformers' CrossEncoder: `predict(pairs)` and
    `rank`).  `model_name` is a LOCAL directory (config.json, model.safetensors with a
    BertForSequenceClassification head, tokenizer.json or vocab.txt); or pass `encoder` + `tokenizer` built elsewhere.
    One label: the score is the logit (`activation="sigmoid"` squashes it as CrossEncoder.predict does by default)."""

    def __init__(self, model_name: Optional[str] = None, device: int = 0, max_length: int = 512,
                 encoder: Any = None, tokenizer: Any = None, activation: Optional[str] = None,
                 score_label: Optional[int] = None):
        from .embeddings import Encoder, _load_tokenizer
        if encoder is None:
            from .encoder_weights import load_local_dir
            cfg, blob, head = load_local_dir(model_name, normalize=False, with_head=True)
            encoder = Encoder(cfg, blob, device=device)
            encoder.set_head(*head)
            max_length = min(max_length, int(cfg.max_position_embeddings))
            tokenizer = _load_tokenizer(model_name, max_length)
        self.encoder, self.tokenizer, self.max_length, self.activation = encoder, tokenizer, int(max_length), activation
        self.score_label = score_label                  # column of a multi-label head that rank() orders by
        # the stock BERT pair pipeline runs natively on all host cores (mi_wordpiece_encode_pairs), token for token with
        # `tokenizers`; anything else, and every pair the native path declines, stays with the library
        self.native_tokenizer = None
        self.slab_pairs = 16384
        if os.environ.get("MI_NATIVE_TOKENIZER", "1") != "0" and hasattr(encoder, "classify_ragged"):
            try:
                from .embeddings import NativeWordPiece
                nat = NativeWordPiece(tokenizer, self.max_length)
                self.native_tokenizer = nat if nat.pairs_supported() else None
            except Exception:
                self.native_tokenizer = None

    def _pairs_csr(self, pairs):
        if self.native_tokenizer is not None:
            return self.native_tokenizer.encode_pairs_csr(pairs)
        return rerank_pairs_csr(pairs, self.tokenizer, self.max_length)

    def predict(self, pairs: Sequence[Sequence[str]]):
        import numpy as np
        if not pairs:
            return np.zeros((0,), dtype=np.float32)
        # pairs of different lengths go down as one flat token array: the library packs its passes without padding
        # (mi_encoder_classify_ragged).  An encoder without that entry point (a test double) gets the padded batch.
        if hasattr(self.encoder, "classify_ragged"):
            # slabs: a worker thread tokenises slab i + 1 (the native call and `tokenizers` both release the GIL) while
            # the GPU scores slab i
            from concurrent.futures import ThreadPoolExecutor
            pairs = pairs if isinstance(pairs, list) else list(pairs)
            cuts = list(range(0, len(pairs), self.slab_pairs)) + [len(pairs)]
            out = []
            with ThreadPoolExecutor(max_workers=1) as ex:
                nxt = ex.submit(self._pairs_csr, pairs[cuts[0]:cuts[1]])
                for j in range(1, len(cuts)):
                    flat_ids, flat_types, offsets = nxt.result()
                    if j + 1 < len(cuts):
                        nxt = ex.submit(self._pairs_csr, pairs[cuts[j]:cuts[j + 1]])
                    out.append(self.encoder.classify_ragged(flat_ids, flat_types, offsets))
            logits = out[0] if len(out) == 1 else np.concatenate(out)
        else:
            docs = [Document(page_content=p[1]) for p in pairs]
            queries = {p[0] for p in pairs}
            if len(queries) == 1:
                ids, types, mask = rerank_pairs(pairs[0][0], docs, self.tokenizer, self.max_length)
            else:
                parts = [rerank_pairs(p[0], [d], self.tokenizer, self.max_length) for p, d in zip(pairs, docs)]
                L = max(a[0].shape[1] for a in parts)
                ids, types, mask = (np.concatenate([np.pad(a[j], ((0, 0), (0, L - a[j].shape[1]))) for a in parts]) for j in range(3))
            logits = self.encoder.classify(ids, types, mask)
        s = logits[:, 0] if logits.shape[1] == 1 else logits
        if self.activation == "sigmoid":
            s = 1.0 / (1.0 + np.exp(-s))
        return s

    def rank(self, query: str, documents: Sequence[Document], top_n: Optional[int] = None) -> List[Document]:
        """Candidates best first by cross-encoder score; ties keep the retriever's order."""
        import numpy as np
        docs = list(documents)
        if not docs:
            return []
        s = self.predict([(query, d.page_content) for d in docs])
        if s.ndim > 1:
            # a multi-label head has no single relevance score: the caller names the column (sentence-transformers'
            # CrossEncoder.rank is defined for num_labels == 1 only)
            if self.score_label is None:
                raise ValueError(f"rank() needs one score per pair; the head has {s.shape[1]} labels: pass score_label=<column>")
            s = s[:, int(self.score_label)]
        order = np.argsort(-s, kind="stable")
        return [docs[i] for i in (order if top_n is None else order[:top_n])]
