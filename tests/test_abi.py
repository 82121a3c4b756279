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


def test_baseline_path_kernels_keep_their_registers():
    """Code-object metadata of the built library (tools/resources.py, no GPU needed): the kernels on the BASELINE
    configurations' paths must not use scratch memory -- a spill in a wave-autonomous kernel costs 10 % at once (round 3: clip
    addressing had put 52 B into the headline encode kernel).  Two known debts are pinned at their recorded size instead."""
    import sys
    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip("libfrad_hip.so not built")
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import resources
    rows = {r["demangled"]: int(r.get("private_segment_fixed_size", 0)) for r in resources.kernels(_lib.LIB_PATH)}
    assert len(rows) > 100
    zero = ["k_p0_fwd_wave<1, 2, 32, false>", "k_p0_inv_wave<2, 32, false>",          # cfg 2 full frames
            "k_p0_fwd_wave<1, 2, 32, true>", "k_p0_inv_wave<2, 32, true>",            # cfg 3 full frames (clip batch, in place)
            "k_p0_fwd_unit<double, PlanA9, 1, 2>", "k_p0_inv_unit<PlanA9, 32, 2>",   # cfg 2's 1024-sample tail
            "k_p0_fwd_half32<11, 4, 32>", "k_p0_inv_grp2<11, 2, 32, 2, false>",      # cfg 4
            "k_p1_fwd_wave<1, 2>", "k_p1_inv_wave<2>", "k_p1_ola<0>", "k_gol_decode_wave", "k_gol_encode"]   # cfg 5 (K7 since the end of round 3)
    for name in zero:
        hit = [k for k in rows if k.endswith(name) or name in k]
        assert hit, name
        for k in hit:
            assert rows[k] == 0, f"{k}: {rows[k]} B of scratch per lane"
    # recorded debts: cfg 3's 896-sample tails -- the mixed-radix kernels call the shared stage-in / pack helpers
    # out of line: 112 B of call frame, no spilled registers
    debts = {"k_p0_fwd_mixed<1>": 112, "k_p0_inv_mixed<0>": 112}
    for name, cap in debts.items():
        hit = [k for k in rows if name in k]
        assert hit, name
        assert max(rows[k] for k in hit) <= cap, (name, [rows[k] for k in hit])
