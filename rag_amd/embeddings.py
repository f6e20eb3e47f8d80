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
        stream = torch.cuda.current_stream(ids.device).cuda_stream
        _capi.check(self._lib.mi_encoder_encode_dev(self._h, B, L, ctypes.c_void_p(ids.data_ptr()),
                                                    ctypes.c_void_p(mask.data_ptr()), ctypes.c_void_p(out.data_ptr()),
                                                    ctypes.c_void_p(stream)))
        return out

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
        self.batch_size = int(self.encode_kwargs.get("batch_size", 32))
        normalize = bool(self.encode_kwargs.get("normalize_embeddings", True))
        cfg, blob = load_local_dir(model_name, normalize=normalize)
        self.max_seq_length = min(int(max_seq_length), int(cfg.max_position_embeddings))
        self.tokenizer = _load_tokenizer(model_name, self.max_seq_length)
        self.encoder = Encoder(cfg, blob, device=self.device)

    # ---- tokenisation + length-sorted batching (sentence-transformers' encode loop)
    def _tokenize(self, texts: List[str]):
        return [e.ids for e in self.tokenizer.encode_batch(texts)]

    def embed_documents_array(self, texts: List[str]) -> np.ndarray:
        texts = [t.replace("\n", " ") for t in texts]
        if not texts:
            return np.zeros((0, self.encoder.hidden_size), dtype=np.float32)
        toks = self._tokenize(texts)
        order = np.argsort([-len(t) for t in toks], kind="stable")
        out = np.empty((len(texts), self.encoder.hidden_size), dtype=np.float32)
        bs = max(self.batch_size, 1)
        for s in range(0, len(order), bs):
            sel = order[s:s + bs]
            L = max(len(toks[i]) for i in sel)
            L = max(L, 1)
            ids = np.zeros((len(sel), L), dtype=np.int32)
            mask = np.zeros((len(sel), L), dtype=np.int32)
            for r, i in enumerate(sel):
                t = toks[i]
                ids[r, :len(t)] = t
                mask[r, :len(t)] = 1
            out[sel] = self.encoder.encode_ids(ids, mask)
        return out

    def embed_documents(self, texts: List[str]) -> List[List[float]]:
        return self.embed_documents_array(list(texts)).tolist()

    def embed_query(self, text: str) -> List[float]:
        return self.embed_documents_array([text])[0].tolist()


# the name the reference imports (advanced_rag.py:17)
HuggingFaceEmbeddings = MI355Embeddings
