"""ctypes binding of libmi355rag.so (include/mi355rag.h).

The product path has no CPU fallback: if the HIP library is missing, or no gfx950 device
is visible, the calls below raise.  ctypes releases the GIL around every call, so the
library may be entered from LangChain RunnableParallel workers and Gradio handler threads
(advanced_rag.py:302-305, 545-555); the C side serialises per handle.
"""
from __future__ import annotations

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# MI355RAG_LIB: another build of the same library (A/B timing of two builds on one device)
LIB_PATH = os.environ.get("MI355RAG_LIB") or os.path.join(_HERE, "lib", "libmi355rag.so")

MI_OK = 0
METRIC_IP, METRIC_L2 = 0, 1
DTYPE_F32, DTYPE_BF16 = 0, 1
MODE_AUTO, MODE_EXACT, MODE_FAST = 0, 1, 2
MAX_K = 2048

_ERR_NAMES = {-1: "MI_ERR_INVALID_ARG", -2: "MI_ERR_NO_DEVICE", -3: "MI_ERR_HIP", -4: "MI_ERR_OOM",
              -5: "MI_ERR_UNSUPPORTED", -6: "MI_ERR_INTERNAL"}


class MI355Error(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"{_ERR_NAMES.get(code, code)}: {msg}")
        self.code = code


class BertConfig(ctypes.Structure):
    """mi_bert_config"""
    _fields_ = [("vocab_size", ctypes.c_int32), ("hidden_size", ctypes.c_int32),
                ("num_hidden_layers", ctypes.c_int32), ("num_attention_heads", ctypes.c_int32),
                ("intermediate_size", ctypes.c_int32), ("max_position_embeddings", ctypes.c_int32),
                ("type_vocab_size", ctypes.c_int32), ("layer_norm_eps", ctypes.c_float),
                ("normalize", ctypes.c_int32)]


# every symbol include/mi355rag.h declares: name -> (restype, argtypes)
_vp, _i64, _i32, _u64 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_uint64
_pp = ctypes.POINTER(ctypes.c_void_p)
_cfgp = ctypes.POINTER(BertConfig)
SYMBOLS = {
    "mi_init": (_i32, [_i32, _pp]),
    "mi_ctx_destroy": (None, [_vp]),
    "mi_ctx_synchronize": (_i32, [_vp]),
    "mi_last_error": (ctypes.c_char_p, []),
    "mi_version": (ctypes.c_char_p, []),
    "mi_index_create": (_i32, [_vp, _i32, _i32, _i32, _pp]),
    "mi_index_destroy": (None, [_vp]),
    "mi_index_reserve": (_i32, [_vp, _i64]),
    "mi_index_add": (_i32, [_vp, _i64, _vp]),
    "mi_index_add_dev": (_i32, [_vp, _i64, _vp, _i32, _vp]),
    "mi_index_add_synthetic": (_i32, [_vp, _i64, _u64, _i32]),
    "mi_index_ntotal": (_i64, [_vp]),
    "mi_index_dim": (_i32, [_vp]),
    "mi_index_set_id_base": (_i32, [_vp, _i64]),
    "mi_index_set_mode": (_i32, [_vp, _i32]),
    "mi_index_reset": (_i32, [_vp]),
    "mi_index_get_rows": (_i32, [_vp, _i64, _i64, _vp]),
    "mi_index_get_rows_raw": (_i32, [_vp, _i64, _i64, _vp]),
    "mi_index_add_raw": (_i32, [_vp, _i64, _vp, _i32]),
    "mi_index_dtype": (_i32, [_vp]),
    "mi_index_search": (_i32, [_vp, _i64, _vp, _i32, _vp, _vp]),
    "mi_index_search_dev": (_i32, [_vp, _i64, _vp, _i32, _vp, _vp, _vp]),
    "mi_index_stats": (_i32, [_vp, ctypes.POINTER(ctypes.c_int64)]),
    "mi_index_last_scan_time": (_i32, [_vp, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]),
    "mi_topk_merge_dev": (_i32, [_vp, _i32, _i64, _i32, _i32, _vp, _vp, _i32, _vp, _vp, _vp]),
    "mi_topk_packed_layout": (_i32, [_i64, _i32, ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_size_t)]),
    "mi_topk_merge_packed_dev": (_i32, [_vp, _i32, _i64, _i32, _i32, _vp, _i32, _vp, _vp, _vp]),
    "mi_comm_unique_id": (_i32, [_vp]),
    "mi_comm_create": (_i32, [_vp, _vp, _i32, _i32, _pp]),
    "mi_comm_destroy": (None, [_vp]),
    "mi_comm_size": (_i32, [_vp]),
    "mi_comm_rank": (_i32, [_vp]),
    "mi_comm_all_gather_dev": (_i32, [_vp, _vp, _vp, ctypes.c_size_t, _vp]),
    "mi_index_search_sharded_dev": (_i32, [_vp, _vp, _i64, _vp, _i32, _vp, _vp, _vp]),
    "mi_multi_index_create": (_i32, [ctypes.POINTER(ctypes.c_int), _i32, _i32, _i32, _i32, _pp]),
    "mi_multi_index_destroy": (None, [_vp]),
    "mi_multi_index_set_block_rows": (_i32, [_vp, _i64]),
    "mi_multi_index_devices": (_i32, [_vp]),
    "mi_multi_index_ntotal": (_i64, [_vp]),
    "mi_multi_index_shard_ntotal": (_i64, [_vp, _i32]),
    "mi_multi_index_set_mode": (_i32, [_vp, _i32]),
    "mi_multi_index_reserve": (_i32, [_vp, _i64]),
    "mi_multi_index_add": (_i32, [_vp, _i64, _vp]),
    "mi_multi_index_add_raw": (_i32, [_vp, _i64, _vp, _i32]),
    "mi_multi_index_add_synthetic": (_i32, [_vp, _i64, _u64, _i32]),
    "mi_multi_index_reset": (_i32, [_vp]),
    "mi_multi_index_get_rows": (_i32, [_vp, _i64, _i64, _vp, _i32]),
    "mi_multi_index_search": (_i32, [_vp, _i64, _vp, _i32, _vp, _vp]),
    "mi_multi_index_search_dev": (_i32, [_vp, _i64, _vp, _i32, _vp, _vp, _vp]),
    "mi_multi_index_stats": (_i32, [_vp, ctypes.POINTER(ctypes.c_int64)]),
    "mi_encoder_blob_elems": (_i64, [_cfgp]),
    "mi_encoder_create": (_i32, [_vp, _cfgp, _vp, ctypes.c_size_t, _pp]),
    "mi_encoder_destroy": (None, [_vp]),
    "mi_encoder_encode": (_i32, [_vp, _i64, _i32, _vp, _vp, _vp]),
    "mi_encoder_encode_dev": (_i32, [_vp, _i64, _i32, _vp, _vp, _vp, _vp]),
    "mi_encoder_encode_dev_noalloc": (_i32, [_vp, _i64, _i32, _vp, _vp, _vp, _vp]),
    "mi_encoder_encode_ragged": (_i32, [_vp, _i64, _vp, _vp, _i32, _i32, _vp]),
    "mi_encoder_set_head": (_i32, [_vp, _vp, _vp, _vp, _vp, _i32]),
    "mi_encoder_classify": (_i32, [_vp, _i64, _i32, _vp, _vp, _vp, _vp]),
    "mi_encoder_classify_dev": (_i32, [_vp, _i64, _i32, _vp, _vp, _vp, _vp, _vp]),
    "mi_encoder_classify_ragged": (_i32, [_vp, _i64, _vp, _vp, _vp, _i32, _vp]),
    "mi_encoder_stats": (_i32, [_vp, ctypes.POINTER(ctypes.c_int64)]),
    "mi_encoder_last_time": (_i32, [_vp, ctypes.POINTER(ctypes.c_float)]),
    "mi_wordpiece_create": (_i32, [ctypes.c_char_p, _i64, _i32, _pp]),
    "mi_wordpiece_set_unicode": (_i32, [_vp, _i64, _vp, _vp, ctypes.c_char_p, _i64]),
    "mi_wordpiece_destroy": (None, [_vp]),
    "mi_wordpiece_encode_batch": (_i32, [_vp, ctypes.c_char_p, _vp, _i64, _i32, _i32, _vp, _vp]),
    "mi_wordpiece_encode_pairs": (_i32, [_vp, ctypes.c_char_p, _vp, ctypes.c_char_p, _vp, _i64, _i32, _i32, _vp, _vp, _vp]),
}

_lib = None


def load():
    """Loads the shared library (no GPU needed for this step) and types every entry point."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `python -m rag_amd._build` "
                "(or __graft_entry__.build()).  rag_amd has no CPU fallback.")
        # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.so.7, and a
        # process that loads /opt/rocm's copy first and torch's second ends up with two
        # runtimes, the second of which sees no GPU.  Importing torch first makes the dynamic
        # loader satisfy our NEEDED libamdhip64.so.7 with the copy torch already mapped.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(lib, name)          # AttributeError if the .so does not export it
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def check(rc: int) -> None:
    if rc != MI_OK:
        raise MI355Error(rc, load().mi_last_error().decode("utf-8", "replace"))


def stream_arg(device=None) -> ctypes.c_void_p:
    """The `stream` argument of a *_dev entry point for PyTorch's current stream on `device`.  PyTorch's default
    stream has handle 0, which the C-ABI reads as "the ctx's own stream" (non-blocking: it would not wait for the
    tensors' producers, nor would their consumers wait for it): pass MI_STREAM_LEGACY for it."""
    import torch
    h = torch.cuda.current_stream(device).cuda_stream
    return ctypes.c_void_p(h if h else 1)


_ctx_cache: dict = {}


def get_ctx(device: int = 0):
    """One mi_ctx per (process, device), created on first use."""
    key = (os.getpid(), int(device))
    if key not in _ctx_cache:
        h = ctypes.c_void_p()
        check(load().mi_init(int(device), ctypes.byref(h)))
        _ctx_cache[key] = h
    return _ctx_cache[key]
