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
    from videonavqa_amd.build import build
    build(verbose=False)
    lib = L.lib()
    declared = _declared()
    assert len(declared) >= 8
    for name in declared:
        assert hasattr(lib, name), name
    # the Python binding covers the whole header
    assert set(declared) == set(L.exported_symbols()), set(declared) ^ set(L.exported_symbols())
    assert lib.vnqa_version() >= 100


def test_missing_library_fails_loudly(monkeypatch):
    import pytest
    monkeypatch.setattr(L, "_lib", None)
    monkeypatch.setattr(L, "LIB_PATH", "/nonexistent/libvnqa_hip.so")
    with pytest.raises(L.VnqaError):
        L.lib()
