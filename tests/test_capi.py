"""The C-ABI library loads and exports every symbol include/mjrl.h declares (no compute without a GPU)."""
import ctypes
import os
import re

import pytest

from mjrl_amd import _capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "mjrl.h")).read()
    return sorted(set(re.findall(r"\b(mjrl_[a-z_]+)\s*\(", text)))


def test_header_and_binding_list_the_same_symbols():
    assert declared_symbols() == sorted(_capi.SYMBOLS)


def test_library_exports_every_declared_symbol():
    if not os.path.exists(_capi.LIB_PATH):
        import __graft_entry__ as entry
        entry.build()
    lib = ctypes.CDLL(_capi.LIB_PATH)
    for name in declared_symbols():
        assert hasattr(lib, name), name
    assert b"gfx950" in _capi.load().mjrl_version()


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(_capi, "_lib", None)
    monkeypatch.setattr(_capi, "LIB_PATH", "/nonexistent/libmjrl_hip.so")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _capi.load()
