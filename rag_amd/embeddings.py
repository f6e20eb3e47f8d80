"""Embeddings behind the interface the reference constructs at advanced_rag.py:104-107:

    HuggingFaceEmbeddings(model_name="sentence-transformers/all-MiniLM-L6-v2",
                          model_kwargs={"device": "cpu"})

Same constructor keywords and the same two methods LangChain calls (`embed_documents`,
`embed_query`); the encoder forward runs in libmi355rag.so on the MI355X.  `model_name`
must be a LOCAL directory (config.json, model.safetensors, tokenizer.json or vocab.txt):
nothing is ever fetched by name.
"""
from __future__ import annotations

import ctypes
import os
from typing import Any, Dict, List, Optional

import numpy as np

from . import _capi
from .encoder_weights import load_local_dir


class Encoder:
    """Thin handle over mi_encoder_*: token ids + mask -> pooled, L2-normalised embeddings."""

    def __init__(self, cfg: _capi.BertConfig, blob: np.ndarray, device: int = 0):
        self._lib = _capi.load()
        self._ctx = _capi.get_ctx(device)
        self.cfg = cfg
        self.device = int(device)
        blob = np.ascontiguousarray(blob, dtype=np.float32)
        want = int(self._lib.mi_encoder_blob_elems(ctypes.byref(cfg)))
        if blob.size != want:
            raise ValueError(f"weights blob has {blob.size} elements, config needs {want}")
        h = ctypes.c_void_p()
        _capi.check(self._lib.mi_encoder_create(self._ctx, ctypes.byref(cfg), blob.ctypes.data_as(ctypes.c_void_p),
                                                blob.size, ctypes.byref(h)))
        self._h = h

    def __del__(self):
        h = getattr(self, "_h", None)
        if h is not None and h.value:
            try:
                self._lib.mi_encoder_destroy(h)
            except Exception:
                pass
            self._h = None

    @property
    def hidden_size(self) -> int:
        return int(self.cfg.hidden_size)

    def encode_ids(self, ids: np.ndarray, mask: np.ndarray) -> np.ndarray:
        ids = np.ascontiguousarray(ids, dtype=np.int32)
        mask = np.ascontiguousarray(mask, dtype=np.int32)
        if ids.ndim != 2 or ids.shape != mask.shape:
            raise ValueError("ids and mask must be [B][L] arrays of the same shape")
        B, L = ids.shape
        out = np.empty((B, self.hidden_size), dtype=np.float32)
        _capi.check(self._lib.mi_encoder_encode(self._h, B, L, ids.ctypes.data_as(ctypes.c_void_p),
                                                mask.ctypes.data_as(ctypes.c_void_p),
                                                out.ctypes.data_as(ctypes.c_void_p)))
        return out

    def encode_ids_torch(self, ids, mask):
        """ids, mask: int32 CUDA tensors [B][L]; returns an fp32 CUDA tensor [B][hidden]."""
        import torch
        B, L = ids.shape
        out = torch.empty((B, self.hidden_size), dtype=torch.float32, device=ids.device)
        stream = _capi.stream_arg(ids.device)
        _capi.check(self._lib.mi_encoder_encode_dev(self._h, B, L, ctypes.c_void_p(ids.data_ptr()),
                                                    ctypes.c_void_p(mask.data_ptr()), ctypes.c_void_p(out.data_ptr()),
                                                    stream))
        return out

    def encode_ids_torch_noalloc(self, ids, mask, out=None):
        """As encode_ids_torch through mi_encoder_encode_dev_noalloc: no hipMalloc inside the library (small batches
        only; the caller may pass `out` so that nothing at all is allocated around the call)."""
        import torch
        B, L = ids.shape
        if out is None:
            out = torch.empty((B, self.hidden_size), dtype=torch.float32, device=ids.device)
        _capi.check(self._lib.mi_encoder_encode_dev_noalloc(self._h, B, L, ctypes.c_void_p(ids.data_ptr()),
                                                            ctypes.c_void_p(mask.data_ptr()), ctypes.c_void_p(out.data_ptr()),
                                                            _capi.stream_arg(ids.device)))
        return out

    def encode_ragged(self, flat_ids: np.ndarray, offsets: np.ndarray, max_len: int = 0, max_batch: int = 0) -> np.ndarray:
        """n ragged sequences (CSR: flat token ids + offsets [n + 1]) -> [n][hidden], input order.  Length-sorted,
        token-budget batching and pinned double-buffered staging happen inside the library (mi_encoder_encode_ragged)."""
        flat_ids = np.ascontiguousarray(flat_ids, dtype=np.int32)
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        if offsets.ndim != 1 or offsets.size < 1 or offsets[0] != 0 or offsets[-1] != flat_ids.size:
            raise ValueError("offsets must be [n + 1], start at 0 and end at len(flat_ids)")
        n = offsets.size - 1
        out = np.empty((n, self.hidden_size), dtype=np.float32)
        _capi.check(self._lib.mi_encoder_encode_ragged(self._h, n, flat_ids.ctypes.data_as(ctypes.c_void_p),
                                                       offsets.ctypes.data_as(ctypes.c_void_p), int(max_len), int(max_batch),
                                                       out.ctypes.data_as(ctypes.c_void_p)))
        return out

    def set_head(self, pooler_w, pooler_b, cls_w, cls_b) -> None:
        """Loads a BertForSequenceClassification head (pooler dense + classifier), fp32."""
        H = self.hidden_size
        pw = np.ascontiguousarray(pooler_w, dtype=np.float32)
        pb = np.ascontiguousarray(pooler_b, dtype=np.float32)
        cw = np.ascontiguousarray(cls_w, dtype=np.float32)
        cb = np.ascontiguousarray(cls_b, dtype=np.float32)
        if pw.shape != (H, H) or pb.shape != (H,) or cw.ndim != 2 or cw.shape[1] != H or cb.shape != (cw.shape[0],):
            raise ValueError("head shapes: pooler_w [H][H], pooler_b [H], cls_w [num_labels][H], cls_b [num_labels]")
        self.num_labels = int(cw.shape[0])
        p = lambda a: a.ctypes.data_as(ctypes.c_void_p)          # noqa: E731
        _capi.check(self._lib.mi_encoder_set_head(self._h, p(pw), p(pb), p(cw), p(cb), self.num_labels))

    def classify(self, ids: np.ndarray, type_ids: Optional[np.ndarray], mask: np.ndarray) -> np.ndarray:
        """Cross-encoder logits [B][num_labels] for padded (query, passage) pair batches."""
        ids = np.ascontiguousarray(ids, dtype=np.int32)
        mask = np.ascontiguousarray(mask, dtype=np.int32)
        if ids.ndim != 2 or ids.shape != mask.shape:
            raise ValueError("ids and mask must be [B][L] arrays of the same shape")
        tp = None
        if type_ids is not None:
            type_ids = np.ascontiguousarray(type_ids, dtype=np.int32)
            if type_ids.shape != ids.shape:
                raise ValueError("type_ids must have the shape of ids")
            tp = type_ids.ctypes.data_as(ctypes.c_void_p)
        B, L = ids.shape
        out = np.empty((B, getattr(self, "num_labels", 1)), dtype=np.float32)
        _capi.check(self._lib.mi_encoder_classify(self._h, B, L, ids.ctypes.data_as(ctypes.c_void_p), tp,
                                                  mask.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p)))
        return out

    def classify_ragged(self, flat_ids: np.ndarray, flat_type_ids: np.ndarray, offsets: np.ndarray, max_batch: int = 0) -> np.ndarray:
        """Cross-encoder logits [n][num_labels] for n pairs of different lengths (CSR: flat token ids and token types,
        offsets [n + 1]); passes are packed without padding inside the library (mi_encoder_classify_ragged)."""
        flat_ids = np.ascontiguousarray(flat_ids, dtype=np.int32)
        flat_type_ids = np.ascontiguousarray(flat_type_ids, dtype=np.int32)
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        if flat_ids.shape != flat_type_ids.shape or flat_ids.ndim != 1 or offsets.ndim != 1 or offsets.size < 1:
            raise ValueError("flat_ids / flat_type_ids must be 1-d arrays of one shape, offsets [n + 1]")
        if int(offsets[-1]) > flat_ids.size:
            raise ValueError("offsets reach beyond flat_ids")
        n = offsets.size - 1
        out = np.empty((n, getattr(self, "num_labels", 1)), dtype=np.float32)
        _capi.check(self._lib.mi_encoder_classify_ragged(self._h, n, flat_ids.ctypes.data_as(ctypes.c_void_p),
                                                         flat_type_ids.ctypes.data_as(ctypes.c_void_p),
                                                         offsets.ctypes.data_as(ctypes.c_void_p), int(max_batch),
                                                         out.ctypes.data_as(ctypes.c_void_p)))
        return out

    def stats(self) -> dict:
        out = (ctypes.c_int64 * 4)()
        _capi.check(self._lib.mi_encoder_stats(self._h, out))
        return {"sequences": out[0], "tokens": out[1], "bulk_passes": out[2], "latency_passes": out[3]}

    def last_time_ms(self) -> float:
        ms = ctypes.c_float()
        _capi.check(self._lib.mi_encoder_last_time(self._h, ctypes.byref(ms)))
        return ms.value


def _load_tokenizer(path: str, max_len: int):
    from tokenizers import Tokenizer
    tj = os.path.join(path, "tokenizer.json")
    if os.path.exists(tj):
        tok = Tokenizer.from_file(tj)
    else:
        vocab = os.path.join(path, "vocab.txt")
        if not os.path.exists(vocab):
            raise FileNotFoundError(f"{path}: need tokenizer.json or vocab.txt")
        from tokenizers import BertWordPieceTokenizer
        tok = BertWordPieceTokenizer(vocab, lowercase=True)._tokenizer
    tok.no_padding()
    tok.enable_truncation(max_length=max_len)
    return tok


class MI355Embeddings:
    """HuggingFaceEmbeddings-compatible: model_name (local dir), model_kwargs, encode_kwargs."""

    def __init__(self, model_name: str, cache_folder: Optional[str] = None,
                 model_kwargs: Optional[Dict[str, Any]] = None, encode_kwargs: Optional[Dict[str, Any]] = None,
                 multi_process: bool = False, max_seq_length: int = 256, **_: Any):
        self.model_name = model_name
        self.model_kwargs = dict(model_kwargs or {})
        self.encode_kwargs = dict(encode_kwargs or {})
        dev = self.model_kwargs.get("device", 0)
        # the reference passes "cpu" (advanced_rag.py:106); this build has only the GPU path
        if isinstance(dev, str):
            dev = int(dev.split(":")[1]) if ":" in dev else 0
        self.device = int(dev)
        self._batch_size_set = "batch_size" in self.encode_kwargs
        self.batch_size = int(self.encode_kwargs.get("batch_size", 32))
        normalize = bool(self.encode_kwargs.get("normalize_embeddings", True))
        cfg, blob = load_local_dir(model_name, normalize=normalize)
        self.max_seq_length = min(int(max_seq_length), int(cfg.max_position_embeddings))
        self.tokenizer = _load_tokenizer(model_name, self.max_seq_length)
        self.encoder = Encoder(cfg, blob, device=self.device)

    # ---- tokenisation; the length-sorted batching of sentence-transformers' encode loop lives in the library
    def _tokenize(self, texts: List[str]):
        return [e.ids for e in self.tokenizer.encode_batch(texts)]

    @staticmethod
    def _to_csr(toks) -> tuple:
        lens = np.fromiter((len(t) for t in toks), dtype=np.int64, count=len(toks))
        offsets = np.zeros(len(toks) + 1, dtype=np.int64)
        np.cumsum(lens, out=offsets[1:])
        flat = np.fromiter((i for t in toks for i in t), dtype=np.int32, count=int(offsets[-1]))
        return flat, offsets

    def embed_tokenized(self, flat_ids: np.ndarray, offsets: np.ndarray) -> np.ndarray:
        """Token ids in (CSR), embeddings out, input order: the step of embed_documents below the tokenizer.
        `encode_kwargs["batch_size"]` caps the rows per pass only when the caller set it; by default a pass is cut
        by the library's token budget, so a bulk call (FAISS.from_documents, advanced_rag.py:295) runs the fused
        bulk kernels."""
        return self.encoder.encode_ragged(flat_ids, offsets, max_len=self.max_seq_length,
                                          max_batch=self.batch_size if self._batch_size_set else 0)

    def embed_documents_array(self, texts: List[str]) -> np.ndarray:
        texts = [t.replace("\n", " ") for t in texts]
        if not texts:
            return np.zeros((0, self.encoder.hidden_size), dtype=np.float32)
        flat, offsets = self._to_csr(self._tokenize(texts))
        return self.embed_tokenized(flat, offsets)

    def embed_documents(self, texts: List[str]) -> List[List[float]]:
        return self.embed_documents_array(list(texts)).tolist()

    def embed_query(self, text: str) -> List[float]:
        return self.embed_documents_array([text])[0].tolist()


# the name the reference imports (advanced_rag.py:17)
HuggingFaceEmbeddings = MI355Embeddings
