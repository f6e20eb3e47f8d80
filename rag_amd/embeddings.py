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


def _is_standard_bert_pipeline(tok) -> Optional[bool]:
    """None unless the tokenizer is the stock BERT pipeline the native WordPiece restates (BertNormalizer with clean_text,
    BertPreTokenizer, WordPiece with [UNK] / ## / 100 characters, [CLS] $A [SEP]); else its `lowercase` flag."""
    import json
    try:
        cfg = json.loads(tok.to_str())
        nz, pt, md, pp = cfg.get("normalizer") or {}, cfg.get("pre_tokenizer") or {}, cfg.get("model") or {}, cfg.get("post_processor") or {}
        if nz.get("type") != "BertNormalizer" or not nz.get("clean_text", False) or pt.get("type") != "BertPreTokenizer":
            return None
        if md.get("type") != "WordPiece" or md.get("unk_token") != "[UNK]" or md.get("continuing_subword_prefix") != "##":
            return None
        if int(md.get("max_input_chars_per_word", 100)) != 100:
            return None
        if pp.get("type") == "BertProcessing":
            if pp.get("cls", [None])[0] != "[CLS]" or pp.get("sep", [None])[0] != "[SEP]":
                return None
        elif pp.get("type") == "TemplateProcessing":
            single = [(list(x.keys())[0], list(x.values())[0].get("id")) for x in pp.get("single", [])]
            if single != [("SpecialToken", "[CLS]"), ("Sequence", "A"), ("SpecialToken", "[SEP]")]:
                return None
        else:
            return None
        import re
        if any(not a.get("special", False) or not re.fullmatch(r"\[[A-Z]+\]", a.get("content", "")) for a in cfg.get("added_tokens") or []):
            return None                                   # added words bypass WordPiece; the native path steps aside only for "[" A-Z+ "]"
        return bool(nz.get("lowercase", True))
    except Exception:
        return None


class NativeWordPiece:
    """mi_wordpiece_* (rag_amd/csrc/wordpiece.hip): the stock BERT tokenizer pipeline in C++ on all host cores, straight
    into the CSR arrays mi_encoder_encode_ragged takes.  ASCII text needs nothing else; for the first text with other
    characters, per-code-point tables (planes 0 and 1) are read off the tokenizer's own normalizer and pre-tokenizer --
    both work character by character, so the tables ARE their behaviour, whatever Unicode version the library carries --
    and handed to mi_wordpiece_set_unicode.  Texts the native path declines (a spelled-out special token, a character
    beyond plane 1, a character of non-zero combining class that survives NFD and could be reordered by it) go through `tokenizers`,
    so the result is the reference tokenizer's for every input: tests/test_wordpiece.py.  Built only for the stock
    pipeline, and only if a probe of the vocabulary reproduces `tokenizers` token for token."""

    UNICODE_CODEPOINTS = 0x20000

    def __init__(self, tok, max_len: int, n_threads: int = 0):
        lower = _is_standard_bert_pipeline(tok)
        if lower is None:
            raise ValueError("not the stock BERT WordPiece pipeline")
        tr = tok.truncation or {}
        if (tr.get("max_length") != int(max_len) or tr.get("direction", "right") != "right" or tok.padding is not None
                or tr.get("strategy", "longest_first") not in ("longest_first", "only_first")):
            raise ValueError("the tokenizer must truncate to max_len (on the right) and not pad (texts the native path declines go through it)")
        vocab = tok.get_vocab()
        size = max(vocab.values()) + 1
        entries = [""] * size
        for t, i in vocab.items():
            if "\n" in t or "\r" in t:
                raise ValueError("vocabulary entry with a line break")
            entries[i] = t
        blob = "\n".join(entries).encode("utf-8")
        self._lib = _capi.load()
        h = ctypes.c_void_p()
        _capi.check(self._lib.mi_wordpiece_create(blob, len(blob), 1 if lower else 0, ctypes.byref(h)))
        self._h, self.tok, self.max_len, self.n_threads = h, tok, int(max_len), int(n_threads)
        self.unicode = None                                                      # None: not tried yet; True / False afterwards
        probe = ["", "Hello, World!  a-b_c", "x" * 101 + " " + "y" * 100, "The quick\tbrown\nfox's 12345.678 #tag ##ing [SEP] [MASK]",
                 "tab\x0bsep\x00nul\x7fdel", " ".join(list(vocab)[:: max(1, len(vocab) // 200)][:300])]
        probe = [p for p in probe if p.isascii()]
        flat, off = self.encode_csr(probe, _self_check=True)
        ref = [e.ids for e in tok.encode_batch(probe)]
        for i, r in enumerate(ref):
            if flat[off[i]:off[i + 1]].tolist() != r:
                self.close()
                raise ValueError(f"native WordPiece disagrees with tokenizers on probe {i}")

    def close(self):
        if getattr(self, "_h", None):
            self._lib.mi_wordpiece_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _unicode_tables(self):
        """(cls uint8 [N], map uint32 [N], pool bytes) for mi_wordpiece_set_unicode, read off the reference tokenizer."""
        import json
        import unicodedata
        N = self.UNICODE_CODEPOINTS
        nz, pt = self.tok.normalizer, self.tok.pre_tokenizer
        ncfg = json.loads(self.tok.to_str()).get("normalizer") or {}
        nfd = ncfg.get("strip_accents")
        nfd = bool(ncfg.get("lowercase", True)) if nfd is None else bool(nfd)       # (BertNormalizer decomposes only to strip accents)
        cls = [0] * N
        mp = [0xFFFFFFFF] * N
        pool = bytearray()
        outs = set()
        norm = nz.normalize_str
        for cp in range(N):
            if 0xD800 <= cp <= 0xDFFF:
                cls[cp] = 3
                continue
            ch = chr(cp)
            s = norm(ch)
            if s != ch:
                b = s.encode("utf-8")
                if len(b) > 255 or any(ord(c) >= N for c in s):
                    cls[cp] = 3                                                  # (maps to itself and declines)
                    continue
                mp[cp] = (len(pool) << 8) | len(b)
                pool += b
                outs.update(s)
            else:
                outs.add(ch)
        split = pt.pre_tokenize_str
        for c in outs:
            kind = {1: 0, 2: 1, 3: 2}.get(len(split("a" + c + "a")), 3)
            # a character of non-zero combining class that survives NFD could be put in canonical order with its
            # neighbours by it, which no per-character table can express: such texts stay with the library (and so does a
            # character this Python does not know but the library kept: its class is anyone's guess)
            if nfd and (unicodedata.combining(c) or unicodedata.category(c) == "Cn"):
                kind = 3
            cls[ord(c)] = max(cls[ord(c)], kind)
        cls, mp = np.array(cls, dtype=np.uint8), np.array(mp, dtype=np.uint32)
        return cls, mp, bytes(pool)

    def _enable_unicode(self) -> bool:
        """Once, at the first non-ASCII text: builds the tables, sets them, and checks a probe against the library."""
        if self.unicode is not None:
            return self.unicode
        self.unicode = False
        try:
            cls, mp, pool = self._unicode_tables()
            _capi.check(self._lib.mi_wordpiece_set_unicode(self._h, cls.size, cls.ctypes.data_as(ctypes.c_void_p),
                                                           mp.ctypes.data_as(ctypes.c_void_p), pool, len(pool)))
            self.unicode = True
            probe = ["caf\u00e9 na\u00efve \u00c5ngstr\u00f6m", "\u201cquoted\u201d \u2014 dash\u2026 \u00a7 1", "\u4e2d\u6587 text \ud55c\uad6d\uc5b4",
                     "\u0394\u03bf\u03ba\u03b9\u03bc\u03ae \u041f\u0440\u0438\u0432\u0435\u0442", "zero\u200bwidth nb\u00a0sp \ufffd \u0130stanbul", "x\u00e9" * 60 + " " + "\u00e9" * 101]
            flat, off = self.encode_csr(probe)
            ref = [e.ids for e in self.tok.encode_batch(probe)]
            if any(flat[off[i]:off[i + 1]].tolist() != r for i, r in enumerate(ref)):
                raise ValueError("native WordPiece disagrees with tokenizers on a Unicode probe")
        except Exception:
            self.unicode = False
            self._lib.mi_wordpiece_set_unicode(self._h, 0, None, None, None, 0)
        return self.unicode

    def pairs_supported(self) -> bool:
        """True when (a, b) pairs are framed [CLS] a [SEP] b [SEP] with segment ids 0 0 0 1 1 and cut "longest_first" with
        no stride: what mi_wordpiece_encode_pairs restates."""
        import json
        try:
            pp = json.loads(self.tok.to_str()).get("post_processor") or {}
            tr = self.tok.truncation or {}
            if tr.get("strategy", "longest_first") != "longest_first" or tr.get("stride", 0) != 0 or self.max_len < 3:
                return False
            if pp.get("type") == "BertProcessing":
                return True
            pair = [(list(x.keys())[0], list(x.values())[0].get("id"), list(x.values())[0].get("type_id")) for x in pp.get("pair", [])]
            return pair == [("SpecialToken", "[CLS]", 0), ("Sequence", "A", 0), ("SpecialToken", "[SEP]", 0),
                            ("Sequence", "B", 1), ("SpecialToken", "[SEP]", 1)]
        except Exception:
            return False

    def encode_pairs_csr(self, pairs):
        """(a, b) string pairs -> (flat int32 ids, flat int32 segment ids, int64 offsets [n + 1]): the cross-encoder's
        input, each pair as the tokenizer's own encode(a, b) gives it.  Pairs the native path declines go through it."""
        n = len(pairs)
        offsets = np.zeros(n + 1, dtype=np.int64)
        if n == 0:
            return np.zeros(0, dtype=np.int32), np.zeros(0, dtype=np.int32), offsets
        ok = np.ones(n, dtype=bool)
        if not all(a.isascii() and b.isascii() for a, b in pairs) and self.unicode is not False:
            self._enable_unicode()
        codec = "utf-8" if self.unicode else "ascii"
        ea, eb = [b""] * n, [b""] * n
        last_a, last_ea = None, b""
        for i, (a, b) in enumerate(pairs):
            try:
                if a is not last_a:
                    last_a, last_ea = a, a.encode(codec)
                ea[i], eb[i] = last_ea, b.encode(codec)
            except UnicodeEncodeError:                                           # non-ASCII without tables, or a lone surrogate
                ok[i] = False
                last_a = None
        offa = np.zeros(n + 1, dtype=np.int64)
        offb = np.zeros(n + 1, dtype=np.int64)
        np.cumsum(np.fromiter((len(x) for x in ea), dtype=np.int64, count=n), out=offa[1:])
        np.cumsum(np.fromiter((len(x) for x in eb), dtype=np.int64, count=n), out=offb[1:])
        rows = np.empty((n, self.max_len), dtype=np.int32)
        types = np.empty((n, self.max_len), dtype=np.int32)
        lens = np.zeros(n, dtype=np.int64)
        p = ctypes.c_void_p
        _capi.check(self._lib.mi_wordpiece_encode_pairs(self._h, b"".join(ea), offa.ctypes.data_as(p), b"".join(eb), offb.ctypes.data_as(p), n,
                                                        self.max_len, self.n_threads, rows.ctypes.data_as(p), types.ctypes.data_as(p),
                                                        lens.ctypes.data_as(p)))
        ok &= lens >= 0
        rest = np.flatnonzero(~ok)
        self.last_native_texts = int(n - rest.size)
        for i in rest:
            e = self.tok.encode(pairs[i][0], pairs[i][1])
            m = len(e.ids)
            rows[i, :m] = e.ids
            types[i, :m] = e.type_ids
            lens[i] = m
        np.cumsum(lens, out=offsets[1:])
        keep = np.arange(self.max_len)[None, :] < lens[:, None]
        return np.ascontiguousarray(rows[keep], dtype=np.int32), np.ascontiguousarray(types[keep], dtype=np.int32), offsets

    def encode_csr(self, texts: List[str], _self_check: bool = False):
        """texts -> (flat int32 ids, int64 offsets [n + 1]), each text as [CLS] pieces [SEP] cut to max_len."""
        n = len(texts)
        offsets = np.zeros(n + 1, dtype=np.int64)
        if n == 0:
            return np.zeros(0, dtype=np.int32), offsets
        ascii_mask = np.fromiter((t.isascii() for t in texts), dtype=bool, count=n)
        if not ascii_mask.all() and not _self_check and self.unicode is not False and self._enable_unicode():
            enc = [None] * n
            for i, t in enumerate(texts):
                try:
                    enc[i] = t.encode("ascii" if ascii_mask[i] else "utf-8")
                    ascii_mask[i] = True                                          # (from here on: "goes to the native path")
                except UnicodeEncodeError:                                       # a lone surrogate: the library's business
                    pass
        else:
            enc = None
        nat = np.flatnonzero(ascii_mask)
        rows = np.empty((n, self.max_len), dtype=np.int32)
        lens = np.zeros(n, dtype=np.int64)
        if nat.size:
            sub = texts if nat.size == n else [texts[i] for i in nat]
            toff = np.zeros(nat.size + 1, dtype=np.int64)
            if enc is None:
                blob = "".join(sub).encode("ascii")
                np.cumsum(np.fromiter((len(t) for t in sub), dtype=np.int64, count=nat.size), out=toff[1:])
            else:
                sub = [enc[i] for i in nat]
                blob = b"".join(sub)
                np.cumsum(np.fromiter((len(b) for b in sub), dtype=np.int64, count=nat.size), out=toff[1:])
            sub_rows = rows if nat.size == n else np.empty((nat.size, self.max_len), dtype=np.int32)
            sub_lens = lens if nat.size == n else np.zeros(nat.size, dtype=np.int64)
            _capi.check(self._lib.mi_wordpiece_encode_batch(self._h, blob, toff.ctypes.data_as(ctypes.c_void_p), nat.size, self.max_len,
                                                            self.n_threads, sub_rows.ctypes.data_as(ctypes.c_void_p),
                                                            sub_lens.ctypes.data_as(ctypes.c_void_p)))
            if nat.size != n:
                rows[nat] = sub_rows
                lens[nat] = sub_lens
            ascii_mask = ascii_mask & (lens >= 0)                                # (-1: declined, e.g. a spelled-out special token)
        rest = np.flatnonzero(~ascii_mask)
        self.last_native_texts = int(n - rest.size)                             # (the others went through `tokenizers`)
        if rest.size:
            fn = getattr(self.tok, "encode_batch_fast", None) or self.tok.encode_batch
            for i, e in zip(rest, fn([texts[i].replace("\n", " ") for i in rest])):
                ids = e.ids
                rows[i, :len(ids)] = ids
                lens[i] = len(ids)
        np.cumsum(lens, out=offsets[1:])
        flat = rows[np.arange(self.max_len)[None, :] < lens[:, None]]          # row-major: every text's ids in order
        return np.ascontiguousarray(flat, dtype=np.int32), offsets


class MI355Embeddings:
    """HuggingFaceEmbeddings-compatible: model_name (local dir), model_kwargs, encode_kwargs.

    Extensions (none changes a result): `devices=[0, ..., 7]` (or model_kwargs={"devices": [...]}) keeps one encoder per
    listed GPU in this one process and splits every bulk call over them -- pure data parallel, no collective, input
    order kept (BASELINE config 3 from the single process advanced_rag.py is); `slab_texts` is the number of texts
    tokenised per step of the host pipeline below."""

    def __init__(self, model_name: str, cache_folder: Optional[str] = None,
                 model_kwargs: Optional[Dict[str, Any]] = None, encode_kwargs: Optional[Dict[str, Any]] = None,
                 multi_process: bool = False, max_seq_length: int = 256, devices: Optional[List[int]] = None,
                 slab_texts: int = 8192, **_: Any):
        self.model_name = model_name
        self.model_kwargs = dict(model_kwargs or {})
        self.encode_kwargs = dict(encode_kwargs or {})
        dev = self.model_kwargs.get("device", 0)
        # the reference passes "cpu" (advanced_rag.py:106); this build has only the GPU path
        if isinstance(dev, str):
            dev = int(dev.split(":")[1]) if ":" in dev else 0
        devices = devices if devices is not None else self.model_kwargs.get("devices")
        self.devices = [int(x) for x in devices] if devices else [int(dev)]
        self.device = self.devices[0]
        self._batch_size_set = "batch_size" in self.encode_kwargs
        self.batch_size = int(self.encode_kwargs.get("batch_size", 32))
        normalize = bool(self.encode_kwargs.get("normalize_embeddings", True))
        cfg, blob = load_local_dir(model_name, normalize=normalize)
        self.max_seq_length = min(int(max_seq_length), int(cfg.max_position_embeddings))
        self.tokenizer = _load_tokenizer(model_name, self.max_seq_length)
        # the stock BERT pipeline on ASCII text runs natively (mi_wordpiece_*); anything else stays with `tokenizers`
        self.native_tokenizer = None
        if os.environ.get("MI_NATIVE_TOKENIZER", "1") != "0":
            try:
                self.native_tokenizer = NativeWordPiece(self.tokenizer, self.max_seq_length)
            except (ValueError, _capi.MI355Error):
                self.native_tokenizer = None
        self.encoders = [Encoder(cfg, blob, device=d) for d in self.devices]
        self.encoder = self.encoders[0]
        self.slab_texts = max(1, int(slab_texts))
        self.last_timing: Dict[str, float] = {}

    # ---- tokenisation; the length-sorted batching of sentence-transformers' encode loop lives in the library
    def _tokenize(self, texts: List[str]):
        # encode_batch_fast (tokenizers >= 0.20) skips the character-offset bookkeeping nobody reads here: 3x the rate
        fn = getattr(self.tokenizer, "encode_batch_fast", None) or self.tokenizer.encode_batch
        return [e.ids for e in fn(texts)]

    @staticmethod
    def _to_csr(toks) -> tuple:
        """Lists of token ids -> (flat int32 ids, int64 offsets [n + 1]).  One array conversion per sequence and one
        concatenate: a per-token Python generator here cost as much as the GPU's whole forward at bulk scale."""
        n = len(toks)
        offsets = np.zeros(n + 1, dtype=np.int64)
        if n == 0:
            return np.zeros(0, dtype=np.int32), offsets
        parts = [np.asarray(t, dtype=np.int32) for t in toks]
        np.cumsum(np.fromiter((p.size for p in parts), dtype=np.int64, count=n), out=offsets[1:])
        flat = np.concatenate(parts) if offsets[-1] else np.zeros(0, dtype=np.int32)
        return flat, offsets

    def _tokenize_csr(self, texts: List[str]) -> tuple:
        if self.native_tokenizer is not None:
            return self.native_tokenizer.encode_csr(texts)
        return self._to_csr(self._tokenize([t.replace("\n", " ") for t in texts]))

    def _encode_csr(self, flat_ids: np.ndarray, offsets: np.ndarray) -> np.ndarray:
        kw = dict(max_len=self.max_seq_length, max_batch=self.batch_size if self._batch_size_set else 0)
        n = offsets.size - 1
        G = len(self.encoders)
        if G == 1 or n < 2 * G:
            return self.encoder.encode_ragged(flat_ids, offsets, **kw)
        # one host thread per GPU (ctypes releases the GIL): contiguous ranges cut at equal token counts
        from concurrent.futures import ThreadPoolExecutor
        cuts = np.searchsorted(offsets, np.linspace(0, offsets[-1], G + 1)[1:-1]).astype(np.int64)
        bounds = np.concatenate([[0], np.clip(cuts, 0, n), [n]])
        out = np.empty((n, self.encoder.hidden_size), dtype=np.float32)

        def run(g):
            lo, hi = int(bounds[g]), int(bounds[g + 1])
            if hi > lo:
                out[lo:hi] = self.encoders[g].encode_ragged(flat_ids[offsets[lo]:offsets[hi]], offsets[lo:hi + 1] - offsets[lo], **kw)
        with ThreadPoolExecutor(max_workers=G) as ex:
            list(ex.map(run, range(G)))
        return out

    def embed_tokenized(self, flat_ids: np.ndarray, offsets: np.ndarray) -> np.ndarray:
        """Token ids in (CSR), embeddings out, input order: the step of embed_documents below the tokenizer.
        `encode_kwargs["batch_size"]` caps the rows per pass only when the caller set it; by default a pass is cut
        by the library's token budget, so a bulk call (FAISS.from_documents, advanced_rag.py:295) runs the fused
        bulk kernels."""
        return self._encode_csr(np.ascontiguousarray(flat_ids, dtype=np.int32), np.ascontiguousarray(offsets, dtype=np.int64))

    def embed_documents_array(self, texts: List[str]) -> np.ndarray:
        """Strings in, [n][hidden] fp32 out.  Bulk calls run as a two-stage host pipeline over slabs of `slab_texts`
        texts: a worker thread tokenises slab i + 1 (the Rust tokenizer releases the GIL and uses the host cores)
        while this thread has slab i on the GPU (mi_encoder_encode_ragged, GIL released).  Per-slab results land at
        their input positions, so order and bits equal the one-shot call's."""
        import time
        n = len(texts)
        if n == 0:
            return np.zeros((0, self.encoder.hidden_size), dtype=np.float32)
        t_all = time.perf_counter()
        S = self.slab_texts
        if n <= S:
            t0 = time.perf_counter()
            flat, offsets = self._tokenize_csr(texts)
            t1 = time.perf_counter()
            out = self._encode_csr(flat, offsets)
            t2 = time.perf_counter()
            self.last_timing = {"tokenize_s": t1 - t0, "encode_s": t2 - t1, "total_s": t2 - t_all, "slabs": 1, "tokens": int(offsets[-1])}
            return out
        from concurrent.futures import ThreadPoolExecutor
        out = np.empty((n, self.encoder.hidden_size), dtype=np.float32)
        tok_s = enc_s = 0.0
        tokens = 0

        # the first slab is short (the GPU has nothing to do until it is tokenised), the others `slab_texts` long
        cuts = [0, min(S, 512)]
        while cuts[-1] < n:
            cuts.append(min(n, cuts[-1] + S))

        def tok(j):
            t0 = time.perf_counter()
            r = self._tokenize_csr(texts[cuts[j]:cuts[j + 1]])
            return r, time.perf_counter() - t0
        with ThreadPoolExecutor(max_workers=1) as ex:
            nxt = ex.submit(tok, 0)
            for j in range(len(cuts) - 1):
                (flat, offsets), dt = nxt.result()
                tok_s += dt
                if j + 2 < len(cuts):
                    nxt = ex.submit(tok, j + 1)
                t0 = time.perf_counter()
                out[cuts[j]:cuts[j + 1]] = self._encode_csr(flat, offsets)
                enc_s += time.perf_counter() - t0
                tokens += int(offsets[-1])
        self.last_timing = {"tokenize_s": tok_s, "encode_s": enc_s, "total_s": time.perf_counter() - t_all,
                            "slabs": len(cuts) - 1, "tokens": tokens}
        return out

    def embed_documents(self, texts: List[str]) -> List[List[float]]:
        return self.embed_documents_array(list(texts)).tolist()

    def embed_query(self, text: str) -> List[float]:
        return self.embed_documents_array([text])[0].tolist()


# the name the reference imports (advanced_rag.py:17)
HuggingFaceEmbeddings = MI355Embeddings
