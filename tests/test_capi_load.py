"""CPU: libmi355rag.so loads without a GPU, exports every symbol include/mi355rag.h declares,
and the product path fails loudly (no CPU fallback) when no gfx950 device is present."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "mi355rag.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mi_[a-z0-9_]+)\s*\(", src)))


def test_header_and_binding_list_the_same_symbols():
    from rag_amd import _capi
    assert _declared_symbols() == sorted(_capi.SYMBOLS)


def test_library_exports_every_declared_symbol():
    from rag_amd import _capi
    lib = _capi.load()
    for name in _declared_symbols():
        assert hasattr(lib, name), name
    assert b"gfx950" in lib.mi_version()


def test_no_torch_types_in_the_abi():
    src = open(os.path.join(ROOT, "include", "mi355rag.h")).read()
    code = re.sub(r"/\*.*?\*/", "", src, flags=re.S)           # declarations only, comments stripped
    assert "torch" not in code and "at::" not in code and "c10::" not in code
    assert set(re.findall(r"#include\s*<([^>]+)>", code)) == {"stddef.h", "stdint.h"}


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.mark.skipif(_has_gpu(), reason="checks the no-GPU failure mode")
def test_fails_loudly_without_a_gpu():
    from rag_amd import _capi
    lib = _capi.load()
    h = ctypes.c_void_p()
    rc = lib.mi_init(0, ctypes.byref(h))
    assert rc == -2 and not h.value                       # MI_ERR_NO_DEVICE
    assert b"no CPU fallback" in lib.mi_last_error() or b"gfx950" in lib.mi_last_error()
    from rag_amd.index import FlatIndex
    with pytest.raises(_capi.MI355Error):
        FlatIndex(8)


def test_encoder_blob_size_matches_oracle(oracle_mod):
    from rag_amd import _capi
    lib = _capi.load()
    cfg = _capi.BertConfig(30522, 384, 6, 12, 1536, 512, 2, 1e-12, 1)
    ocfg = oracle_mod.BertConfig(30522, 384, 6, 12, 1536, 512, 2, 1e-12, 1)
    n = lib.mi_encoder_blob_elems(ctypes.byref(cfg))
    assert n == oracle_mod.blob_elems(ocfg) and n > 22_000_000


def test_product_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "rag_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(import|from)\s+oracle\b", txt, flags=re.M), f
                assert "liboracle" not in txt, f
