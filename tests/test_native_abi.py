"""CPU-side checks of the C-ABI library: it loads, and exports every symbol the header declares."""
import ctypes as C
import os
import re

import pytest

from conftest import ROOT


def _built():
    from audiocodecs_amd import _native

    if not os.path.exists(_native.lib_path):
        import __graft_entry__

        __graft_entry__.build()
    return _native


def test_library_exports_every_declared_symbol():
    native = _built()
    header = open(os.path.join(ROOT, "include", "audiocodecs_amd.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(ac_[a-z_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    L = C.CDLL(native.lib_path)
    for sym in sorted(declared):
        assert hasattr(L, sym), f"{sym} declared in include/audiocodecs_amd.h but not exported"
    assert declared == set(native.EXPORTS), declared ^ set(native.EXPORTS)
    assert native.lib().ac_version() >= 100


def test_struct_layout_matches_header():
    native = _built()
    assert C.sizeof(native.AcConfig) == 4 * (5 + 8 + 8)
    assert C.sizeof(native.AcKernelStat) == 96 + 4 + 4 + 8 + 8
    assert C.sizeof(native.AcMimiConfig) == 4 * (5 + 8 + 4 + 4 + 7 + 2)
    assert C.sizeof(native.AcDacConfig) == 4 * (5 + 8 + 8 + 4 + 4 + 1)
    assert C.sizeof(native.AcWavtokConfig) == 4 * (5 + 8 + 14)


def test_create_rejects_bad_config_without_gpu():
    native = _built()
    L = native.lib()
    cfg = native.AcConfig()
    h = C.c_void_p()
    assert L.ac_create(C.byref(cfg), C.byref(h)) == -1  # struct_size == 0 -> AC_EINVAL
    assert L.ac_create(None, C.byref(h)) == -1
    assert L.ac_mimi_create(C.byref(native.AcMimiConfig()), C.byref(h)) == -1
    assert L.ac_dac_create(C.byref(native.AcDacConfig()), C.byref(h)) == -1
    assert L.ac_wavtok_create(C.byref(native.AcWavtokConfig()), C.byref(h)) == -1
    assert L.ac_set_precision(None, 0) == -1
    assert L.ac_last_error(None) == b"null handle"


def test_product_library_carries_no_developer_switch():
    """the product library neither reads the fault-injection / timing-mode environment words nor accepts their keys
    (the refusal's message is in it); the developer library built beside it does.  The GPU side of this is
    tests/test_fault_injection_gpu.py."""
    import os

    native = _built()
    native.lib()
    data = open(native.lib_path, "rb").read()
    assert b"AC_RB6_DBG" not in data and b"AC_LSTM_DBG" not in data
    assert b"developer-build switch" in data
    dev = os.path.join(os.path.dirname(native.lib_path), "libaudiocodecs_amd_dev.so")
    if os.path.exists(dev) and "AUDIOCODECS_AMD_LIB" not in os.environ:
        assert b"AC_LSTM_DBG" in open(dev, "rb").read()
