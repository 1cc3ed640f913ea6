"""CPU: the C-ABI library loads and exports every symbol include/vnqa_hip.h declares."""
import os
import re

from videonavqa_amd import _lib as L

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _declared():
    text = open(os.path.join(ROOT, "include", "vnqa_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(vnqa_[a-z0-9_]+)\s*\(", text)))


def test_library_loads_and_exports_every_declared_symbol():
    import ctypes
    from videonavqa_amd.build import build
    build(verbose=False, variant="all")
    lib = L.lib()
    declared = _declared()
    assert len(declared) >= 8
    for name in declared:
        assert hasattr(lib, name), name
    # both storage builds (fp16: the default precision's; bf16: BASELINE.json's dtype) export the same ABI
    for path in (L.LIB_PATH, L.LIB_PATH_F16):
        other = ctypes.CDLL(path)
        for name in declared:
            assert hasattr(other, name), (path, name)
    # the Python binding covers the whole header
    assert set(declared) == set(L.exported_symbols()), set(declared) ^ set(L.exported_symbols())
    assert lib.vnqa_version() >= 100


def test_missing_library_fails_loudly(monkeypatch):
    import pytest
    monkeypatch.setattr(L, "_lib", None)
    monkeypatch.setattr(L, "LIB_PATH", "/nonexistent/libvnqa_hip.so")
    monkeypatch.setattr(L, "LIB_PATH_F16", "/nonexistent/libvnqa_hip_f16.so")
    with pytest.raises(L.VnqaError):
        L.lib()
