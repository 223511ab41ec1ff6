"""The C-ABI library loads and exports exactly what include/drn.h declares (no compute: runs without a GPU)."""
import os
import re

from conftest import ROOT


def _declared():
    text = open(os.path.join(ROOT, "include", "drn.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(drn_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree(pkg):
    assert _declared() == sorted(pkg.native.SIGNATURES.keys())


def test_library_exports_every_symbol(pkg):
    import __graft_entry__ as ge
    ge.build()
    lib = pkg.native.load_library()
    for name in _declared():
        assert hasattr(lib, name), name
    assert lib.drn_abi_version() == 1
    assert b"unsupported" in lib.drn_error_string(-1)


def test_argument_counts_match_header(pkg):
    text = open(os.path.join(ROOT, "include", "drn.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    for name, argtypes in pkg.native.SIGNATURES.items():
        m = re.search(r"\b" + name + r"\s*\(([^;]*?)\)\s*;", text, flags=re.S)
        assert m, name
        args = m.group(1).strip()
        n = 0 if args in ("", "void") else len(args.split(","))
        assert n == len(argtypes), (name, n, len(argtypes))


def test_missing_library_fails_loudly(pkg, monkeypatch):
    import pytest
    monkeypatch.setattr(pkg.native, "_LIB", None)
    monkeypatch.setattr(pkg.native, "_LIB_NAME", "libdrn_missing.so")
    with pytest.raises(RuntimeError, match="no CPU / eager fallback"):
        pkg.native.load_library()
