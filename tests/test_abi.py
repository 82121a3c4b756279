"""The C-ABI library loads (no GPU needed for that) and exports every symbol include/frad_hip.h declares;
the Python binding declares exactly the same set.  No compute calls here."""
import os
import re

import pytest

from frad_python_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "frad_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(frad_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert declared_symbols() == sorted(_lib.SYMBOLS)


def test_library_exports_every_declared_symbol():
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    lib = _lib.FradLib(_lib.LIB_PATH)                      # resolves every symbol or raises
    assert lib.dll.frad_abi_version() == 1
    assert lib.payload_bytes(2048, 2, 32) == 16384 and lib.payload_bytes(5, 1, 12) == 8
    assert lib.dll.frad_strerror(-3).decode().startswith("HIP runtime error")
    assert lib.has_fast_path(2048, 2) and not lib.has_fast_path(896, 2)


def test_product_path_fails_loudly_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from frad_python_amd import core
    from frad_python_amd.bridge import HipBridge
    with pytest.raises(RuntimeError):
        core.analogue_batch(0, torch.zeros(4096, dtype=torch.int16), "s16le", 1, 2048, 1, 32)
    with pytest.raises(RuntimeError):
        HipBridge()
    with pytest.raises(RuntimeError):
        _lib.FradLib("/nonexistent/libfrad_hip.so")
