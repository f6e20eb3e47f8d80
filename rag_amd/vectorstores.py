"""Vector store + retriever behind the interface the reference calls:

    self.vector_store = FAISS.from_documents(self.split_data, self.embed_func)      advanced_rag.py:295
    self.faiss_retriever = self.vector_store.as_retriever(search_kwargs={"k": 5})    advanced_rag.py:296
    RunnableLambda(extract_question) | ensemble_retriever                            advanced_rag.py:256,303

Same names, argument meaning and error behaviour as langchain-community==0.0.19's FAISS
wrapper (requirements.txt:3): default metric is squared L2 (ascending), `k` defaults to 4,
label -1 is skipped, best first.  The index lives in HBM (rag_amd.index.FlatIndex); nothing
numeric happens in this file.
"""
from __future__ import annotations

import uuid
from typing import Any, Callable, Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np

from .documents import Document
from .index import FlatIndex, MultiFlatIndex

EUCLIDEAN_DISTANCE = "EUCLIDEAN_DISTANCE"
MAX_INNER_PRODUCT = "MAX_INNER_PRODUCT"
COSINE = "COSINE"


class InMemoryDocstore:
    """id -> Document map (langchain_community.docstore.in_memory.InMemoryDocstore)."""

    def __init__(self, _dict: Optional[Dict[str, Document]] = None):
        self._dict: Dict[str, Document] = dict(_dict) if _dict else {}

    def add(self, texts: Dict[str, Document]) -> None:
        overlapping = set(texts).intersection(self._dict)
        if overlapping:
            raise ValueError(f"Tried to add ids that already exist: {overlapping}")
        self._dict.update(texts)

    def delete(self, ids: List[str]) -> None:
        missing = set(ids).difference(self._dict)
        if missing:
            raise ValueError(f"Tried to delete ids that does not  exist: {missing}")
        for _id in ids:
            self._dict.pop(_id)

    def search(self, search: str):
        if search not in self._dict:
            return f"ID {search} not found."
        return self._dict[search]


def _runnable_lambda(fn: Callable):
    """RunnableLambda when langchain_core is installed, else a tiny pipe with the same
    `.invoke` / `|` behaviour, so `extract_question | retriever` (advanced_rag.py:256) works."""
    try:
        from langchain_core.runnables import RunnableLambda  # type: ignore
        return RunnableLambda(fn)
    except Exception:
        return _Pipe([fn])


class _Pipe:
    def __init__(self, steps: List[Callable]):
        self.steps = list(steps)

    def invoke(self, value: Any, config: Any = None, **kwargs: Any) -> Any:
        for s in self.steps:
            value = s.invoke(value) if hasattr(s, "invoke") else s(value)
        return value

    __call__ = invoke

    def __or__(self, other: Any) -> "_Pipe":
        return _Pipe(self.steps + (other.steps if isinstance(other, _Pipe) else [other]))

    def __ror__(self, other: Any) -> "_Pipe":
        return _Pipe((other.steps if isinstance(other, _Pipe) else [other]) + self.steps)


class VectorStoreRetriever:
    """`vector_store.as_retriever(search_kwargs={"k": k})` (advanced_rag.py:296): maps
    str -> List[Document]; composes with `|` like an LCEL runnable."""

    allowed_search_types = ("similarity", "similarity_score_threshold")

    def __init__(self, vectorstore: "MI355VectorStore", search_type: str = "similarity",
                 search_kwargs: Optional[dict] = None, tags: Optional[List[str]] = None):
        if search_type not in self.allowed_search_types:
            raise ValueError(f"search_type of {search_type} not allowed. Valid values are: {self.allowed_search_types}")
        self.vectorstore = vectorstore
        self.search_type = search_type
        self.search_kwargs = dict(search_kwargs or {})
        self.tags = tags or []
        if search_type == "similarity_score_threshold":
            thr = self.search_kwargs.get("score_threshold")
            if thr is None or not isinstance(thr, float):
                raise ValueError("`score_threshold` is not specified with a float value(0~1) in `search_kwargs`.")

    def get_relevant_documents(self, query: str, **kwargs: Any) -> List[Document]:
        if self.search_type == "similarity":
            return self.vectorstore.similarity_search(query, **self.search_kwargs)
        docs_and_sims = self.vectorstore.similarity_search_with_relevance_scores(query, **self.search_kwargs)
        return [doc for doc, _ in docs_and_sims]

    def invoke(self, input: str, config: Any = None, **kwargs: Any) -> List[Document]:
        return self.get_relevant_documents(input)

    async def ainvoke(self, input: str, config: Any = None, **kwargs: Any) -> List[Document]:
        return self.get_relevant_documents(input)

    def batch(self, inputs: Sequence[str], config: Any = None, **kwargs: Any) -> List[List[Document]]:
        """All queries in one corpus pass (the reference would loop)."""
        k = self.search_kwargs.get("k", 4)
        return self.vectorstore.similarity_search_batch(list(inputs), k=k)

    __call__ = invoke

    def __or__(self, other: Any):
        return _runnable_lambda(self.invoke) | other

    def __ror__(self, other: Any):
        left = other if hasattr(other, "invoke") else _runnable_lambda(other)
        return left | _runnable_lambda(self.invoke)


_LC_RETRIEVER = None


def _langchain_retriever_class():
    """A `langchain_core.retrievers.BaseRetriever` subclass over the same store, when langchain_core is importable.
    LangChain's own EnsembleRetriever (advanced_rag.py:19, :298) is a pydantic model whose `retrievers` field
    validates its members as Runnables: the plain VectorStoreRetriever above composes with `|` but would be rejected
    there, so with LangChain present `as_retriever` hands out this class instead (documents are converted to
    langchain_core's Document).  NOT TESTED in this repository: langchain_core is absent from the build image and
    the GPU box; without it, import BM25Retriever / EnsembleRetriever from rag_amd as well."""
    global _LC_RETRIEVER
    if _LC_RETRIEVER is not None:
        return _LC_RETRIEVER or None
    try:
        from langchain_core.documents import Document as LCDocument  # type: ignore
        from langchain_core.retrievers import BaseRetriever  # type: ignore
    except Exception:
        _LC_RETRIEVER = False
        return None

    class LangChainVectorStoreRetriever(BaseRetriever):  # pragma: no cover - needs langchain_core
        vectorstore: Any
        search_type: str = "similarity"
        search_kwargs: dict = {}

        class Config:
            arbitrary_types_allowed = True

        def _get_relevant_documents(self, query: str, *, run_manager: Any = None) -> List[Any]:
            inner = VectorStoreRetriever(self.vectorstore, search_type=self.search_type, search_kwargs=self.search_kwargs)
            return [LCDocument(page_content=d.page_content, metadata=d.metadata) for d in inner.get_relevant_documents(query)]

    _LC_RETRIEVER = LangChainVectorStoreRetriever
    return _LC_RETRIEVER


class MI355VectorStore:
    """Flat exact vector store with the FAISS wrapper's constructor and methods."""

    def __init__(self, embedding_function: Any, index: FlatIndex, docstore: InMemoryDocstore,
                 index_to_docstore_id: Dict[int, str], relevance_score_fn: Optional[Callable[[float], float]] = None,
                 normalize_L2: bool = False, distance_strategy: str = EUCLIDEAN_DISTANCE):
        self.embedding_function = embedding_function
        self.index = index
        self.docstore = docstore
        self.index_to_docstore_id = index_to_docstore_id
        self.distance_strategy = distance_strategy
        self.override_relevance_score_fn = relevance_score_fn
        self._normalize_L2 = normalize_L2

    # ---- embedding hooks
    @property
    def embeddings(self):
        return self.embedding_function if hasattr(self.embedding_function, "embed_query") else None

    def _embed_documents(self, texts: List[str]) -> np.ndarray:
        ef = self.embedding_function
        if hasattr(ef, "embed_documents_array"):
            return np.asarray(ef.embed_documents_array(texts), dtype=np.float32)
        if hasattr(ef, "embed_documents"):
            return np.asarray(ef.embed_documents(texts), dtype=np.float32)
        return np.asarray([ef(t) for t in texts], dtype=np.float32)

    def _embed_query(self, text: str) -> np.ndarray:
        ef = self.embedding_function
        v = ef.embed_query(text) if hasattr(ef, "embed_query") else ef(text)
        return np.asarray(v, dtype=np.float32)

    @staticmethod
    def _l2_normalize(x: np.ndarray) -> np.ndarray:
        n = np.linalg.norm(x, axis=1, keepdims=True)
        return x / np.maximum(n, 1e-10)

    # ---- build
    def __add(self, texts: Iterable[str], embeddings: np.ndarray, metadatas: Optional[List[dict]] = None,
              ids: Optional[List[str]] = None) -> List[str]:
        texts = list(texts)
        embeddings = np.asarray(embeddings, dtype=np.float32).reshape(len(texts), -1)
        if metadatas is not None and len(metadatas) != len(texts):
            raise ValueError("metadatas and texts must have the same length")
        documents = [Document(page_content=t, metadata=(metadatas[i] if metadatas else {})) for i, t in enumerate(texts)]
        ids = list(ids) if ids is not None else [str(uuid.uuid4()) for _ in texts]
        if len(ids) != len(texts):
            raise ValueError("ids and texts must have the same length")
        if len(set(ids)) != len(ids):
            raise ValueError("Duplicate ids found in the ids list.")
        if self._normalize_L2:
            embeddings = self._l2_normalize(embeddings)
        self.docstore.add({_id: doc for _id, doc in zip(ids, documents)})
        start = len(self.index_to_docstore_id)
        if len(texts):
            self.index.add(embeddings)
        self.index_to_docstore_id.update({start + j: _id for j, _id in enumerate(ids)})
        return ids

    def add_texts(self, texts: Iterable[str], metadatas: Optional[List[dict]] = None,
                  ids: Optional[List[str]] = None, **kwargs: Any) -> List[str]:
        texts = list(texts)
        return self.__add(texts, self._embed_documents(texts), metadatas=metadatas, ids=ids)

    def add_documents(self, documents: List[Document], **kwargs: Any) -> List[str]:
        return self.add_texts([d.page_content for d in documents], [d.metadata for d in documents], **kwargs)

    def add_embeddings(self, text_embeddings: Iterable[Tuple[str, List[float]]], metadatas: Optional[List[dict]] = None,
                       ids: Optional[List[str]] = None, **kwargs: Any) -> List[str]:
        pairs = list(text_embeddings)
        return self.__add([t for t, _ in pairs], np.asarray([e for _, e in pairs], dtype=np.float32),
                          metadatas=metadatas, ids=ids)

    @classmethod
    def __from(cls, texts: List[str], embeddings: np.ndarray, embedding: Any, metadatas: Optional[List[dict]] = None,
               ids: Optional[List[str]] = None, normalize_L2: bool = False,
               distance_strategy: str = EUCLIDEAN_DISTANCE, dtype: str = "f32", device: int = 0,
               devices: Optional[Sequence[int]] = None, **kwargs: Any) -> "MI355VectorStore":
        embeddings = np.asarray(embeddings, dtype=np.float32)
        if embeddings.ndim != 2 or embeddings.shape[0] == 0:
            # the reference path would fail on embeddings[0]; keep that an IndexError
            raise IndexError("list index out of range")
        metric = "ip" if distance_strategy == MAX_INNER_PRODUCT else "l2"
        # devices=[0, ..., 7]: one store over several GPUs from this one process (the reference is one process,
        # advanced_rag.py:336-337); rows round-robin by blocks, answer identical to the single-GPU store
        if devices is not None and len(devices) > 0:
            index = MultiFlatIndex(embeddings.shape[1], metric=metric, dtype=dtype, devices=devices)
        else:
            index = FlatIndex(embeddings.shape[1], metric=metric, dtype=dtype, device=device)
        store = cls(embedding, index, InMemoryDocstore(), {}, normalize_L2=normalize_L2,
                    distance_strategy=distance_strategy, **kwargs)
        store.__add(texts, embeddings, metadatas=metadatas, ids=ids)
        return store

    @classmethod
    def from_texts(cls, texts: List[str], embedding: Any, metadatas: Optional[List[dict]] = None,
                   ids: Optional[List[str]] = None, **kwargs: Any) -> "MI355VectorStore":
        texts = list(texts)
        if hasattr(embedding, "embed_documents_array"):
            emb = embedding.embed_documents_array(texts)
        else:
            emb = embedding.embed_documents(texts)
        return cls.__from(texts, np.asarray(emb, dtype=np.float32), embedding, metadatas=metadatas, ids=ids, **kwargs)

    @classmethod
    def from_documents(cls, documents: List[Document], embedding: Any, **kwargs: Any) -> "MI355VectorStore":
        """FAISS.from_documents(self.split_data, self.embed_func) — advanced_rag.py:295."""
        return cls.from_texts([d.page_content for d in documents], embedding,
                              metadatas=[d.metadata for d in documents], **kwargs)

    @classmethod
    def from_embeddings(cls, text_embeddings: Iterable[Tuple[str, List[float]]], embedding: Any,
                        metadatas: Optional[List[dict]] = None, ids: Optional[List[str]] = None,
                        **kwargs: Any) -> "MI355VectorStore":
        pairs = list(text_embeddings)
        return cls.__from([t for t, _ in pairs], np.asarray([e for _, e in pairs], dtype=np.float32), embedding,
                          metadatas=metadatas, ids=ids, **kwargs)

    # ---- search
    def search_batch(self, Q: np.ndarray, k: int) -> Tuple[np.ndarray, np.ndarray]:
        """(scores [nq][k], labels [nq][k]) for a matrix of query vectors; -1 padded."""
        Q = np.asarray(Q, dtype=np.float32)
        if Q.ndim == 1:
            Q = Q[None, :]
        if self._normalize_L2:
            Q = self._l2_normalize(Q)
        return self.index.search(Q, k)

    def _docs_from(self, scores: np.ndarray, labels: np.ndarray, k: int, filter: Any = None,
                   score_threshold: Optional[float] = None) -> List[Tuple[Document, float]]:
        out: List[Tuple[Document, float]] = []
        for s, i in zip(scores, labels):
            if i == -1:
                continue
            _id = self.index_to_docstore_id[int(i)]
            doc = self.docstore.search(_id)
            if not isinstance(doc, Document):
                raise ValueError(f"Could not find document for id {_id}, got {doc}")
            if filter is not None:
                if callable(filter):
                    if not filter(doc.metadata):
                        continue
                else:
                    flt = {key: (v if isinstance(v, list) else [v]) for key, v in filter.items()}
                    if not all(doc.metadata.get(key) in v for key, v in flt.items()):
                        continue
            out.append((doc, float(s)))
        if score_threshold is not None:
            keep = (lambda s: s >= score_threshold) if self.distance_strategy == MAX_INNER_PRODUCT \
                else (lambda s: s <= score_threshold)
            out = [(d, s) for d, s in out if keep(s)]
        return out[:k]

    def similarity_search_with_score_by_vector(self, embedding: List[float], k: int = 4, filter: Any = None,
                                               fetch_k: int = 20, **kwargs: Any) -> List[Tuple[Document, float]]:
        kk = k if filter is None else fetch_k
        scores, labels = self.search_batch(np.asarray(embedding, dtype=np.float32), kk)
        return self._docs_from(scores[0], labels[0], k, filter=filter, score_threshold=kwargs.get("score_threshold"))

    def similarity_search_with_score(self, query: str, k: int = 4, filter: Any = None, fetch_k: int = 20,
                                     **kwargs: Any) -> List[Tuple[Document, float]]:
        return self.similarity_search_with_score_by_vector(self._embed_query(query), k, filter=filter,
                                                           fetch_k=fetch_k, **kwargs)

    def similarity_search_by_vector(self, embedding: List[float], k: int = 4, filter: Any = None,
                                    fetch_k: int = 20, **kwargs: Any) -> List[Document]:
        return [d for d, _ in self.similarity_search_with_score_by_vector(embedding, k, filter=filter,
                                                                           fetch_k=fetch_k, **kwargs)]

    def similarity_search(self, query: str, k: int = 4, filter: Any = None, fetch_k: int = 20,
                          **kwargs: Any) -> List[Document]:
        return [d for d, _ in self.similarity_search_with_score(query, k, filter=filter, fetch_k=fetch_k, **kwargs)]

    def similarity_search_batch(self, queries: List[str], k: int = 4) -> List[List[Document]]:
        """Many query strings against one corpus pass."""
        if not queries:
            return []
        ef = self.embedding_function
        if hasattr(ef, "embed_documents_array"):
            Q = np.asarray(ef.embed_documents_array(list(queries)), dtype=np.float32)
        else:
            Q = np.asarray([self._embed_query(q) for q in queries], dtype=np.float32)
        scores, labels = self.search_batch(Q, k)
        return [[d for d, _ in self._docs_from(s, l, k)] for s, l in zip(scores, labels)]

    # ---- relevance scores (LangChain's [0,1] mapping)
    def _select_relevance_score_fn(self) -> Callable[[float], float]:
        if self.override_relevance_score_fn is not None:
            return self.override_relevance_score_fn
        if self.distance_strategy == MAX_INNER_PRODUCT:
            return lambda d: 1.0 - d if d > 0 else -1.0 * d
        if self.distance_strategy == EUCLIDEAN_DISTANCE:
            return lambda d: 1.0 - d / (2.0 ** 0.5)
        if self.distance_strategy == COSINE:
            return lambda d: 1.0 - d
        raise ValueError(f"Unknown distance strategy {self.distance_strategy}")

    def similarity_search_with_relevance_scores(self, query: str, k: int = 4, **kwargs: Any) -> List[Tuple[Document, float]]:
        fn = self._select_relevance_score_fn()
        thr = kwargs.pop("score_threshold", None)
        out = [(d, fn(s)) for d, s in self.similarity_search_with_score(query, k, **kwargs)]
        if thr is not None:
            out = [(d, s) for d, s in out if s >= thr]
        return out

    # ---- retriever
    def as_retriever(self, **kwargs: Any) -> VectorStoreRetriever:
        tags = (kwargs.pop("tags", None) or []) + [type(self).__name__]
        lc = _langchain_retriever_class()
        if lc is not None:                                  # LangChain present: a real BaseRetriever / Runnable
            VectorStoreRetriever(self, tags=tags, **kwargs)   # same argument validation
            return lc(vectorstore=self, tags=tags, **kwargs)
        return VectorStoreRetriever(self, tags=tags, **kwargs)

    # ---- persistence.  LangChain's save_local pickles the docstore; this
    # format executes nothing on load: rows.npy (fp32, or uint16 bf16 bit patterns) + store.json.
    def save_local(self, folder_path: str, index_name: str = "index") -> None:
        import json
        import os
        os.makedirs(folder_path, exist_ok=True)
        n = self.index.ntotal
        dtype = "bf16" if getattr(self.index, "dtype", 0) == 1 else "f32"
        # rows in the storage type (bf16 bit patterns as uint16: the stored values exactly), streamed in slabs into
        # the .npy file: neither widened to fp32 nor held on the host as a whole
        npdt = np.uint16 if dtype == "bf16" else np.float32
        path = os.path.join(folder_path, f"{index_name}.rows.npy")
        if n and hasattr(self.index, "reconstruct_raw"):
            mm = np.lib.format.open_memmap(path, mode="w+", dtype=npdt, shape=(n, self.index.d))
            slab = max(1, (256 << 20) // (self.index.d * 4))
            for r0 in range(0, n, slab):
                m = min(slab, n - r0)
                mm[r0:r0 + m] = self.index.reconstruct_raw(r0, m)
            mm.flush()
            del mm
        else:
            rows = self.index.reconstruct_n(0, n) if n else np.zeros((0, self.index.d), dtype=np.float32)
            if dtype == "bf16":                  # stored values are exactly representable: keep the high halves
                rows = (rows.view(np.uint32) >> 16).astype(np.uint16)
            np.save(path, rows, allow_pickle=False)
        meta = {"format": "mi355rag-store-1", "d": int(self.index.d), "dtype": dtype,
                "distance_strategy": self.distance_strategy, "normalize_L2": bool(self._normalize_L2),
                "ids": [self.index_to_docstore_id[i] for i in range(len(self.index_to_docstore_id))],
                "documents": {k: {"page_content": v.page_content, "metadata": v.metadata}
                              for k, v in self.docstore._dict.items()}}
        with open(os.path.join(folder_path, f"{index_name}.store.json"), "w") as f:
            json.dump(meta, f)

    @classmethod
    def load_local(cls, folder_path: str, embeddings: Any, index_name: str = "index", device: int = 0,
                   index_factory: Optional[Callable[..., Any]] = None, devices: Optional[Sequence[int]] = None,
                   **kwargs: Any) -> "MI355VectorStore":
        import json
        import os
        with open(os.path.join(folder_path, f"{index_name}.store.json")) as f:
            meta = json.load(f)
        if meta.get("format") != "mi355rag-store-1":
            raise ValueError(f"{folder_path}: not a mi355rag store")
        rows = np.load(os.path.join(folder_path, f"{index_name}.rows.npy"), allow_pickle=False, mmap_mode="r")
        metric = "ip" if meta["distance_strategy"] == MAX_INNER_PRODUCT else "l2"
        if devices:
            make = index_factory or (lambda d, metric, dtype: MultiFlatIndex(d, metric=metric, dtype=dtype, devices=devices))
        else:
            make = index_factory or (lambda d, metric, dtype: FlatIndex(d, metric=metric, dtype=dtype, device=device))
        index = make(meta["d"], metric=metric, dtype=meta["dtype"])
        slab = max(1, (256 << 20) // (int(meta["d"]) * 4))
        for r0 in range(0, len(rows), slab):
            part = np.ascontiguousarray(rows[r0:r0 + slab])
            if hasattr(index, "add_raw"):
                index.add_raw(part)
            else:
                index.add((part.astype(np.uint32) << 16).view(np.float32) if meta["dtype"] == "bf16" else part)
        docs = {k: Document(page_content=v["page_content"], metadata=v["metadata"]) for k, v in meta["documents"].items()}
        return cls(embeddings, index, InMemoryDocstore(docs), {i: _id for i, _id in enumerate(meta["ids"])},
                   normalize_L2=meta["normalize_L2"], distance_strategy=meta["distance_strategy"], **kwargs)

    # ---- maintenance
    def delete_all(self) -> None:
        """reset_app_updated drops the store (advanced_rag.py:435-437)."""
        self.index.reset()
        self.docstore = InMemoryDocstore()
        self.index_to_docstore_id = {}


# the name the reference imports (advanced_rag.py:16)
FAISS = MI355VectorStore
