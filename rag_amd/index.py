"""FlatIndex — the exact (flat) index the vector store owns.

Mirrors the slice of faiss.IndexFlatL2 / IndexFlatIP that langchain-community's FAISS wrapper
uses behind advanced_rag.py:295-296: `d`, `ntotal`, `add(x)`, `search(x, k) -> (D, I)`,
`reconstruct_n`, `reset`.  All arithmetic runs in libmi355rag.so on an MI355X.
"""
from __future__ import annotations

import ctypes
from typing import Optional, Tuple

import numpy as np

from . import _capi
from ._capi import DTYPE_BF16, DTYPE_F32, METRIC_IP, METRIC_L2, MODE_AUTO, MODE_EXACT, MODE_FAST  # noqa: F401


def _np_ptr(a: np.ndarray):
    return a.ctypes.data_as(ctypes.c_void_p)


class FlatIndex:
    """Exact top-k over a row-major matrix resident in HBM.

    metric: "l2" (squared L2, ascending — faiss.IndexFlatL2, LangChain's default) or
            "ip" (inner product, descending — faiss.IndexFlatIP).
    dtype : "f32" or "bf16" storage.  A bf16 index rounds rows AND queries to bf16 (RNE).
    """

    def __init__(self, d: int, metric: str = "l2", dtype: str = "f32", device: int = 0):
        self._lib = _capi.load()
        self._ctx = _capi.get_ctx(device)
        self.d = int(d)
        self.metric = {"l2": METRIC_L2, "ip": METRIC_IP}[metric.lower()]
        self.dtype = {"f32": DTYPE_F32, "fp32": DTYPE_F32, "float32": DTYPE_F32,
                      "bf16": DTYPE_BF16, "bfloat16": DTYPE_BF16}[dtype.lower()]
        self.device = int(device)
        h = ctypes.c_void_p()
        _capi.check(self._lib.mi_index_create(self._ctx, self.d, self.metric, self.dtype, ctypes.byref(h)))
        self._h = h

    def __del__(self):
        h = getattr(self, "_h", None)
        if h is not None and h.value:
            try:
                self._lib.mi_index_destroy(h)
            except Exception:
                pass
            self._h = None

    # ---- faiss-like surface
    @property
    def ntotal(self) -> int:
        return int(self._lib.mi_index_ntotal(self._h))

    def reserve(self, n_rows: int) -> None:
        _capi.check(self._lib.mi_index_reserve(self._h, int(n_rows)))

    def add(self, x) -> None:
        x = np.ascontiguousarray(x, dtype=np.float32)
        if x.ndim != 2 or x.shape[1] != self.d:
            raise ValueError(f"add: expected [n][{self.d}] array, got {x.shape}")
        _capi.check(self._lib.mi_index_add(self._h, x.shape[0], _np_ptr(x)))

    def add_torch(self, x) -> None:
        """Zero-copy add from a CUDA(ROCm) torch tensor, fp32 or bf16, [n][d] contiguous."""
        import torch
        if not x.is_cuda or x.dim() != 2 or x.shape[1] != self.d or not x.is_contiguous():
            raise ValueError("add_torch: need a contiguous [n][d] tensor on the GPU")
        src = {torch.float32: DTYPE_F32, torch.bfloat16: DTYPE_BF16}[x.dtype]
        stream = _capi.stream_arg(x.device)
        _capi.check(self._lib.mi_index_add_dev(self._h, x.shape[0], ctypes.c_void_p(x.data_ptr()), src,
                                               stream))
        torch.cuda.current_stream(x.device).synchronize()

    def add_synthetic(self, n: int, seed: int = 0, normalize: bool = True) -> None:
        _capi.check(self._lib.mi_index_add_synthetic(self._h, int(n), int(seed), int(normalize)))

    def search(self, q, k: int) -> Tuple[np.ndarray, np.ndarray]:
        q = np.ascontiguousarray(q, dtype=np.float32)
        if q.ndim != 2 or q.shape[1] != self.d:
            raise ValueError(f"search: expected [nq][{self.d}] array, got {q.shape}")
        nq = q.shape[0]
        D = np.empty((nq, k), dtype=np.float32)
        I = np.empty((nq, k), dtype=np.int64)
        _capi.check(self._lib.mi_index_search(self._h, nq, _np_ptr(q), int(k), _np_ptr(D), _np_ptr(I)))
        return D, I

    def search_torch(self, q, k: int):
        """Device-resident search: q is a contiguous fp32 CUDA tensor [nq][d]; returns CUDA
        tensors (scores fp32 [nq][k], ids int64 [nq][k])."""
        import torch
        if not q.is_cuda or q.dtype != torch.float32 or q.dim() != 2 or q.shape[1] != self.d or not q.is_contiguous():
            raise ValueError("search_torch: need a contiguous fp32 [nq][d] tensor on the GPU")
        nq = q.shape[0]
        D = torch.empty((nq, k), dtype=torch.float32, device=q.device)
        I = torch.empty((nq, k), dtype=torch.int64, device=q.device)
        stream = _capi.stream_arg(q.device)
        _capi.check(self._lib.mi_index_search_dev(self._h, nq, ctypes.c_void_p(q.data_ptr()), int(k),
                                                  ctypes.c_void_p(D.data_ptr()), ctypes.c_void_p(I.data_ptr()),
                                                  stream))
        return D, I

    def search_packed(self, q, k: int, block) -> None:
        """Device-resident search whose result lands in a packed block (uint8 CUDA tensor of
        mi_topk_packed_layout(nq, k) bytes: [nq][k] fp32 scores | [nq][k] int64 ids) -- the message of the
        sharded search's single all-gather (rag_amd.distributed)."""
        import torch
        if not q.is_cuda or q.dtype != torch.float32 or q.dim() != 2 or q.shape[1] != self.d or not q.is_contiguous():
            raise ValueError("search_packed: need a contiguous fp32 [nq][d] tensor on the GPU")
        nq = q.shape[0]
        off, nbytes = ctypes.c_size_t(), ctypes.c_size_t()
        _capi.check(self._lib.mi_topk_packed_layout(nq, int(k), ctypes.byref(off), ctypes.byref(nbytes)))
        if block.dtype != torch.uint8 or not block.is_contiguous() or block.numel() < nbytes.value or block.data_ptr() % 8:
            raise ValueError(f"search_packed: block must be a contiguous, 8-byte aligned uint8 tensor of >= {nbytes.value} bytes")
        stream = _capi.stream_arg(q.device)
        _capi.check(self._lib.mi_index_search_dev(self._h, nq, ctypes.c_void_p(q.data_ptr()), int(k),
                                                  ctypes.c_void_p(block.data_ptr()), ctypes.c_void_p(block.data_ptr() + off.value),
                                                  stream))

    def reconstruct_n(self, row0: int, n: int) -> np.ndarray:
        out = np.empty((n, self.d), dtype=np.float32)
        _capi.check(self._lib.mi_index_get_rows(self._h, int(row0), int(n), _np_ptr(out)))
        return out

    def reconstruct_raw(self, row0: int, n: int) -> np.ndarray:
        """Rows in the storage type: float32 [n][d], or uint16 bf16 bit patterns [n][d] for a bf16 index."""
        out = np.empty((n, self.d), dtype=np.float32 if self.dtype == DTYPE_F32 else np.uint16)
        _capi.check(self._lib.mi_index_get_rows_raw(self._h, int(row0), int(n), _np_ptr(out)))
        return out

    def add_raw(self, x: np.ndarray) -> None:
        """Append host rows given as float32 or as uint16 bf16 bit patterns (a shard file's contents)."""
        if x.dtype == np.uint16:
            src = DTYPE_BF16
        elif x.dtype == np.float32:
            src = DTYPE_F32
        else:
            raise ValueError("add_raw: float32 or uint16 (bf16 bit patterns) rows")
        x = np.ascontiguousarray(x)
        if x.ndim != 2 or x.shape[1] != self.d:
            raise ValueError(f"add_raw: expected [n][{self.d}] array, got {x.shape}")
        _capi.check(self._lib.mi_index_add_raw(self._h, x.shape[0], _np_ptr(x), src))

    def reset(self) -> None:
        _capi.check(self._lib.mi_index_reset(self._h))

    # ---- extensions
    def set_id_base(self, base: int) -> None:
        _capi.check(self._lib.mi_index_set_id_base(self._h, int(base)))

    def set_mode(self, mode: int) -> None:
        _capi.check(self._lib.mi_index_set_mode(self._h, int(mode)))

    def stats(self) -> dict:
        out = (ctypes.c_int64 * 4)()
        _capi.check(self._lib.mi_index_stats(self._h, out))
        return {"queries": out[0], "fast": out[1], "exact": out[2], "rows": out[3]}

    def last_scan_time(self) -> dict:
        ms, n, kind = ctypes.c_float(), ctypes.c_int(), ctypes.c_int()
        _capi.check(self._lib.mi_index_last_scan_time(self._h, ctypes.byref(ms), ctypes.byref(n), ctypes.byref(kind)))
        return {"ms": ms.value, "launches": n.value, "kind": "mfma_bf16" if kind.value == 1 else "exact_f32"}


class MultiFlatIndex:
    """ONE exact flat index over several GPUs, driven from this one process (mi_multi_index_*): the store behind
    `FAISS.from_documents(..., devices=[0, ..., 7])`.  Same surface as FlatIndex; ids are global row numbers in insertion
    order; the answer equals the single-GPU answer in ids and score bits.  `devices` may repeat a device (one shard
    per entry)."""

    def __init__(self, d: int, metric: str = "l2", dtype: str = "f32", devices=(0,), block_rows: Optional[int] = None):
        self._lib = _capi.load()
        self.d = int(d)
        self.metric = {"l2": METRIC_L2, "ip": METRIC_IP}[metric.lower()]
        self.dtype = {"f32": DTYPE_F32, "fp32": DTYPE_F32, "float32": DTYPE_F32,
                      "bf16": DTYPE_BF16, "bfloat16": DTYPE_BF16}[dtype.lower()]
        self.devices = [int(x) for x in devices]
        if not self.devices:
            raise ValueError("MultiFlatIndex: devices must name at least one GPU")
        self.device = self.devices[0]
        arr = (ctypes.c_int * len(self.devices))(*self.devices)
        h = ctypes.c_void_p()
        _capi.check(self._lib.mi_multi_index_create(arr, len(self.devices), self.d, self.metric, self.dtype, ctypes.byref(h)))
        self._h = h
        if block_rows is not None:
            _capi.check(self._lib.mi_multi_index_set_block_rows(self._h, int(block_rows)))

    def __del__(self):
        h = getattr(self, "_h", None)
        if h is not None and h.value:
            try:
                self._lib.mi_multi_index_destroy(h)
            except Exception:
                pass
            self._h = None

    @property
    def ntotal(self) -> int:
        return int(self._lib.mi_multi_index_ntotal(self._h))

    def shard_ntotal(self, shard: int) -> int:
        return int(self._lib.mi_multi_index_shard_ntotal(self._h, int(shard)))

    def reserve(self, n_rows: int) -> None:
        _capi.check(self._lib.mi_multi_index_reserve(self._h, int(n_rows)))

    def add(self, x) -> None:
        x = np.ascontiguousarray(x, dtype=np.float32)
        if x.ndim != 2 or x.shape[1] != self.d:
            raise ValueError(f"add: expected [n][{self.d}] array, got {x.shape}")
        _capi.check(self._lib.mi_multi_index_add(self._h, x.shape[0], _np_ptr(x)))

    def add_raw(self, x: np.ndarray) -> None:
        if x.dtype == np.uint16:
            src = DTYPE_BF16
        elif x.dtype == np.float32:
            src = DTYPE_F32
        else:
            raise ValueError("add_raw: float32 or uint16 (bf16 bit patterns) rows")
        x = np.ascontiguousarray(x)
        if x.ndim != 2 or x.shape[1] != self.d:
            raise ValueError(f"add_raw: expected [n][{self.d}] array, got {x.shape}")
        _capi.check(self._lib.mi_multi_index_add_raw(self._h, x.shape[0], _np_ptr(x), src))

    def add_synthetic(self, n: int, seed: int = 0, normalize: bool = True) -> None:
        _capi.check(self._lib.mi_multi_index_add_synthetic(self._h, int(n), int(seed), int(normalize)))

    def search(self, q, k: int) -> Tuple[np.ndarray, np.ndarray]:
        q = np.ascontiguousarray(q, dtype=np.float32)
        if q.ndim != 2 or q.shape[1] != self.d:
            raise ValueError(f"search: expected [nq][{self.d}] array, got {q.shape}")
        nq = q.shape[0]
        D = np.empty((nq, k), dtype=np.float32)
        I = np.empty((nq, k), dtype=np.int64)
        _capi.check(self._lib.mi_multi_index_search(self._h, nq, _np_ptr(q), int(k), _np_ptr(D), _np_ptr(I)))
        return D, I

    def search_torch(self, q, k: int):
        """q: contiguous fp32 CUDA tensor [nq][d] on the FIRST listed device; results on that device."""
        import torch
        if (not q.is_cuda or q.dtype != torch.float32 or q.dim() != 2 or q.shape[1] != self.d or not q.is_contiguous()
                or q.device.index != self.devices[0]):
            raise ValueError(f"search_torch: need a contiguous fp32 [nq][d] tensor on cuda:{self.devices[0]}")
        nq = q.shape[0]
        D = torch.empty((nq, k), dtype=torch.float32, device=q.device)
        I = torch.empty((nq, k), dtype=torch.int64, device=q.device)
        _capi.check(self._lib.mi_multi_index_search_dev(self._h, nq, ctypes.c_void_p(q.data_ptr()), int(k),
                                                        ctypes.c_void_p(D.data_ptr()), ctypes.c_void_p(I.data_ptr()),
                                                        _capi.stream_arg(q.device)))
        return D, I

    def reconstruct_n(self, row0: int, n: int) -> np.ndarray:
        out = np.empty((n, self.d), dtype=np.float32)
        _capi.check(self._lib.mi_multi_index_get_rows(self._h, int(row0), int(n), _np_ptr(out), 0))
        return out

    def reconstruct_raw(self, row0: int, n: int) -> np.ndarray:
        out = np.empty((n, self.d), dtype=np.float32 if self.dtype == DTYPE_F32 else np.uint16)
        _capi.check(self._lib.mi_multi_index_get_rows(self._h, int(row0), int(n), _np_ptr(out), 1))
        return out

    def reset(self) -> None:
        _capi.check(self._lib.mi_multi_index_reset(self._h))

    def set_mode(self, mode: int) -> None:
        _capi.check(self._lib.mi_multi_index_set_mode(self._h, int(mode)))

    def stats(self) -> dict:
        out = (ctypes.c_int64 * 4)()
        _capi.check(self._lib.mi_multi_index_stats(self._h, out))
        return {"queries": out[0], "fast": out[1], "exact": out[2], "rows": out[3]}


def topk_merge_torch(scores, ids, k: int, metric: str = "ip", device: Optional[int] = None):
    """Merge best-first lists: scores/ids CUDA tensors [nq][m]; returns ([nq][k], [nq][k])."""
    import torch
    lib = _capi.load()
    dev = scores.device.index if device is None else device
    ctx = _capi.get_ctx(dev or 0)
    nq, m = scores.shape
    scores = scores.contiguous().float()
    ids = ids.contiguous().long()
    D = torch.empty((nq, k), dtype=torch.float32, device=scores.device)
    I = torch.empty((nq, k), dtype=torch.int64, device=scores.device)
    stream = _capi.stream_arg(scores.device)
    met = {"l2": METRIC_L2, "ip": METRIC_IP}[metric.lower()]
    _capi.check(lib.mi_topk_merge_dev(ctx, met, nq, 1, m, ctypes.c_void_p(scores.data_ptr()),
                                      ctypes.c_void_p(ids.data_ptr()), int(k), ctypes.c_void_p(D.data_ptr()),
                                      ctypes.c_void_p(I.data_ptr()), stream))
    return D, I


def topk_merge_packed_torch(packed, nq: int, nlists: int, k_in: int, k: int, metric: str = "ip", device: Optional[int] = None):
    """Merge `nlists` packed blocks (rank-major, as all_gather_into_tensor leaves them; see FlatIndex.search_packed)
    into ([nq][k] scores, [nq][k] ids), reading the blocks in place."""
    import torch
    lib = _capi.load()
    dev = packed.device.index if device is None else device
    ctx = _capi.get_ctx(dev or 0)
    if packed.dtype != torch.uint8 or not packed.is_contiguous():
        raise ValueError("topk_merge_packed_torch: need a contiguous uint8 tensor")
    D = torch.empty((nq, k), dtype=torch.float32, device=packed.device)
    I = torch.empty((nq, k), dtype=torch.int64, device=packed.device)
    stream = _capi.stream_arg(packed.device)
    met = {"l2": METRIC_L2, "ip": METRIC_IP}[metric.lower()]
    _capi.check(lib.mi_topk_merge_packed_dev(ctx, met, nq, int(nlists), int(k_in), ctypes.c_void_p(packed.data_ptr()), int(k),
                                             ctypes.c_void_p(D.data_ptr()), ctypes.c_void_p(I.data_ptr()), stream))
    return D, I
