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


def test_struct_mirrors_match_the_library(pkg):
    """native.DitForwardArgs / DitSub are field-for-field mirrors of drn_dit_forward_args / drn_dit_sub: same size as compiled,
    and the attention plan / workspace helpers are host-only (callable without a GPU)."""
    import ctypes
    N = pkg.native
    lib = N.load_library()
    assert lib.drn_dit_forward_args_bytes() == ctypes.sizeof(N.DitForwardArgs)
    assert lib.drn_dit_sub_bytes() == ctypes.sizeof(N.DitSub)
    assert N.attention_plan(1, 32, 18432, 18432) == [(0, 18432, 1)]
    assert lib.drn_dit_forward_attn_workspace_bytes(1, 32, 18432) == 0
    # cfg 1: split-K partials of the widest few-token product (MLP-down, 16 slices of [256, 4096] fp32) fit the workspace
    assert lib.drn_dit_forward_gemm_workspace_bytes(1, 256, 4096, 16384, 128, 192) >= 16 * 256 * 4096 * 4
    a = N.DitForwardArgs()
    a.struct_bytes = ctypes.sizeof(N.DitForwardArgs) - 8
    assert lib.drn_dit_forward(ctypes.byref(a), None) == -1          # DRN_EINVAL before anything is launched
