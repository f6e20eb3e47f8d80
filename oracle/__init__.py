"""ctypes/NumPy front end of the CPU oracle (oracle/*.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package (rag_amd/) never imports this module.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")

METRIC_IP, METRIC_L2 = 0, 1
DTYPE_F32, DTYPE_BF16 = 0, 1


def build(force: bool = False) -> str:
    """Compile liboracle.so with gcc (oracle/Makefile) if missing or stale."""
    srcs = [os.path.join(_HERE, f) for f in ("oracle_search.c", "oracle_encoder.c", "Makefile")]
    stale = force or not os.path.exists(_LIB_PATH) or any(
        os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in srcs)
    if stale:
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B"])
    return _LIB_PATH


class BertConfig(ctypes.Structure):
    _fields_ = [("vocab_size", ctypes.c_int32), ("hidden_size", ctypes.c_int32),
                ("num_hidden_layers", ctypes.c_int32), ("num_attention_heads", ctypes.c_int32),
                ("intermediate_size", ctypes.c_int32), ("max_position_embeddings", ctypes.c_int32),
                ("type_vocab_size", ctypes.c_int32), ("layer_norm_eps", ctypes.c_float),
                ("normalize", ctypes.c_int32)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = ctypes.CDLL(_LIB_PATH)
        vp, i64, i32, f32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_float
        L.oracle_search.argtypes = [i32, i32, i64, i32, vp, i64, vp, i32, i64, vp, vp, i32]
        L.oracle_search.restype = i32
        L.oracle_round_bf16.argtypes = [vp, vp, i64]
        L.oracle_round_bf16.restype = None
        L.oracle_score_pair.argtypes = [i32, i32, i32, vp, vp]
        L.oracle_score_pair.restype = f32
        L.oracle_topk_merge.argtypes = [i32, i64, i64, vp, vp, i32, vp, vp]
        L.oracle_topk_merge.restype = i32
        L.oracle_synth_rows.argtypes = [ctypes.c_uint64, i64, i64, i32, i32, vp]
        L.oracle_synth_rows.restype = None
        L.oracle_num_threads.restype = i32
        L.oracle_encoder_blob_elems.argtypes = [ctypes.POINTER(BertConfig)]
        L.oracle_encoder_blob_elems.restype = i64
        L.oracle_encode.argtypes = [ctypes.POINTER(BertConfig), vp, i64, i32, vp, vp, vp, i32]
        L.oracle_encode.restype = i32
        L.oracle_classify.argtypes = [ctypes.POINTER(BertConfig), vp, vp, i32, i64, i32, vp, vp, vp, vp, i32]
        L.oracle_classify.restype = i32
        L.oracle_encode_taps.argtypes = [ctypes.POINTER(BertConfig), vp, i32, vp, vp, vp, vp, vp]
        L.oracle_encode_taps.restype = i32
        _lib = L
    return _lib


def _p(a: np.ndarray):
    return a.ctypes.data_as(ctypes.c_void_p)


def num_threads() -> int:
    return int(lib().oracle_num_threads())


def round_bf16(x: np.ndarray) -> np.ndarray:
    """fp32 -> bf16 bit patterns (uint16), round-to-nearest-even."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    out = np.empty(x.shape, dtype=np.uint16)
    lib().oracle_round_bf16(_p(x), _p(out), x.size)
    return out


def bf16_to_f32(h: np.ndarray) -> np.ndarray:
    return (h.astype(np.uint32) << 16).view(np.float32)


def search(corpus: np.ndarray, queries: np.ndarray, k: int, metric: int = METRIC_IP,
           dtype: int = DTYPE_F32, id_base: int = 0, nthreads: int = 0):
    """Exact top-k with the canonical arithmetic.  corpus is fp32 [n][d]; for DTYPE_BF16 it is
    rounded to bf16 here (or pass uint16 bf16 bit patterns directly)."""
    if dtype == DTYPE_BF16:
        c = corpus if corpus.dtype == np.uint16 else round_bf16(corpus)
    else:
        c = np.ascontiguousarray(corpus, dtype=np.float32)
    c = np.ascontiguousarray(c)
    q = np.ascontiguousarray(queries, dtype=np.float32)
    n, d = c.shape if c.ndim == 2 else (0, q.shape[1])
    nq = q.shape[0]
    scores = np.empty((nq, k), dtype=np.float32)
    ids = np.empty((nq, k), dtype=np.int64)
    rc = lib().oracle_search(metric, dtype, n, d, _p(c), nq, _p(q), k, id_base, _p(scores), _p(ids), nthreads)
    if rc != 0:
        raise ValueError("oracle_search: bad arguments")
    return scores, ids


def score_pair(q: np.ndarray, x: np.ndarray, metric: int = METRIC_IP, dtype: int = DTYPE_F32) -> float:
    q = np.ascontiguousarray(q, dtype=np.float32)
    x = np.ascontiguousarray(x, dtype=np.float32)
    return float(lib().oracle_score_pair(metric, dtype, q.shape[0], _p(q), _p(x)))


def topk_merge(scores: np.ndarray, ids: np.ndarray, k: int, metric: int = METRIC_IP):
    s = np.ascontiguousarray(scores, dtype=np.float32)
    i = np.ascontiguousarray(ids, dtype=np.int64)
    nq, m = s.shape
    os_ = np.empty((nq, k), dtype=np.float32)
    oi = np.empty((nq, k), dtype=np.int64)
    if lib().oracle_topk_merge(metric, nq, m, _p(s), _p(i), k, _p(os_), _p(oi)) != 0:
        raise ValueError("oracle_topk_merge: bad arguments")
    return os_, oi


def synth_rows(seed: int, row0: int, n: int, d: int, normalize: bool = True) -> np.ndarray:
    out = np.empty((n, d), dtype=np.float32)
    lib().oracle_synth_rows(seed, row0, n, d, int(normalize), _p(out))
    return out


def blob_elems(cfg: BertConfig) -> int:
    return int(lib().oracle_encoder_blob_elems(ctypes.byref(cfg)))


def encode(cfg: BertConfig, blob: np.ndarray, ids: np.ndarray, mask: np.ndarray, nthreads: int = 0) -> np.ndarray:
    blob = np.ascontiguousarray(blob, dtype=np.float32)
    assert blob.size == blob_elems(cfg), "weights blob size does not match the config"
    ids = np.ascontiguousarray(ids, dtype=np.int32)
    mask = np.ascontiguousarray(mask, dtype=np.int32)
    B, L = ids.shape
    out = np.empty((B, cfg.hidden_size), dtype=np.float32)
    if lib().oracle_encode(ctypes.byref(cfg), _p(blob), B, L, _p(ids), _p(mask), _p(out), nthreads) != 0:
        raise ValueError("oracle_encode: bad arguments (token id out of range or L too long)")
    return out


def classify(cfg: BertConfig, blob: np.ndarray, head, ids: np.ndarray, types, mask: np.ndarray, nthreads: int = 0) -> np.ndarray:
    """Cross-encoder logits [B][num_labels]; head = (pooler_w [H][H], pooler_b [H], cls_w [nl][H], cls_b [nl])."""
    blob = np.ascontiguousarray(blob, dtype=np.float32)
    assert blob.size == blob_elems(cfg), "weights blob size does not match the config"
    pw, pb, cw, cb = (np.ascontiguousarray(a, dtype=np.float32) for a in head)
    nl = cw.shape[0]
    hflat = np.concatenate([pw.reshape(-1), pb.reshape(-1), cw.reshape(-1), cb.reshape(-1)])
    ids = np.ascontiguousarray(ids, dtype=np.int32)
    mask = np.ascontiguousarray(mask, dtype=np.int32)
    tp = None
    if types is not None:
        types = np.ascontiguousarray(types, dtype=np.int32)
        tp = _p(types)
    B, L = ids.shape
    out = np.empty((B, nl), dtype=np.float32)
    if lib().oracle_classify(ctypes.byref(cfg), _p(blob), _p(hflat), nl, B, L, _p(ids), tp, _p(mask), _p(out), nthreads) != 0:
        raise ValueError("oracle_classify: bad arguments")
    return out


def encode_taps(cfg: BertConfig, blob: np.ndarray, ids: np.ndarray, mask: np.ndarray):
    blob = np.ascontiguousarray(blob, dtype=np.float32)
    ids = np.ascontiguousarray(ids, dtype=np.int32)
    mask = np.ascontiguousarray(mask, dtype=np.int32)
    L = ids.shape[0]
    H, nl = cfg.hidden_size, cfg.num_hidden_layers
    out = np.empty((H,), dtype=np.float32)
    tap_emb = np.empty((L, H), dtype=np.float32)
    tap_layers = np.empty((nl, L, H), dtype=np.float32)
    if lib().oracle_encode_taps(ctypes.byref(cfg), _p(blob), L, _p(ids), _p(mask), _p(out), _p(tap_emb), _p(tap_layers)) != 0:
        raise ValueError("oracle_encode_taps: bad arguments")
    return out, tap_emb, tap_layers
