"""rag_amd — MI355X-native dense-retrieval hot path of alxd/rag's advanced_rag.py.

Drop-in surface (same names the reference imports at advanced_rag.py:16-17,21):
    from rag_amd import FAISS, HuggingFaceEmbeddings, Document
Everything numeric runs in rag_amd/lib/libmi355rag.so (hand-written HIP for gfx950); there is
no CPU fallback.
"""
from ._capi import MI355Error, LIB_PATH  # noqa: F401
from .documents import Document  # noqa: F401
from .index import FlatIndex  # noqa: F401
from .vectorstores import FAISS, MI355VectorStore, VectorStoreRetriever  # noqa: F401
from .embeddings import HuggingFaceEmbeddings, MI355Embeddings  # noqa: F401
from .retrievers import BM25Retriever, CrossEncoderReranker, EnsembleRetriever  # noqa: F401

__all__ = ["Document", "FlatIndex", "FAISS", "MI355VectorStore", "VectorStoreRetriever",
           "HuggingFaceEmbeddings", "MI355Embeddings", "MI355Error", "BM25Retriever", "EnsembleRetriever", "CrossEncoderReranker"]
