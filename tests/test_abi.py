"""The C-ABI library must load without a GPU and export exactly what include/lycoris_amd.h declares."""
import ctypes
import os
import re

import pytest

from conftest import ROOT

HEADER = os.path.join(ROOT, "include", "lycoris_amd.h")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(lyc_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_something():
    names = declared_functions()
    assert "lyc_lokr_linear_fwd" in names and "lyc_abi_version" in names and len(names) >= 14


def test_library_loads_and_exports_every_declared_symbol():
    from lycoris_amd import _native
    lib = _native.load()
    assert lib.lyc_abi_version() == _native.ABI_VERSION
    raw = ctypes.CDLL(_native.lib_path())
    for name in declared_functions():
        assert hasattr(raw, name), f"{name} is declared in the header but not exported"


def test_python_binding_table_matches_header():
    from lycoris_amd import _native
    bound = set(_native.SIGNATURES) | set(_native.VALUE_SIGNATURES) | {"lyc_abi_version", "lyc_last_error"}
    assert bound == set(declared_functions())
    src = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    for name, argtypes in _native.SIGNATURES.items():
        m = re.search(r"\b" + name + r"\s*\(([^;]*?)\)\s*;", src, flags=re.S)
        assert m, name
        nargs = len([a for a in m.group(1).split(",") if a.strip()])
        assert nargs == len(argtypes), f"{name}: header has {nargs} parameters, binding has {len(argtypes)}"


def test_argument_errors_are_reported_not_crashed():
    """Argument checks run before any launch, so they work on the CPU-only build box."""
    from lycoris_amd import _native as N
    with pytest.raises(RuntimeError, match="128x128"):
        N.call("lyc_lokr_linear_fwd", None, None, None, None, 4, 200, 8, 4, 4, 1.0, N.LYC_BF16, None)
    with pytest.raises(RuntimeError, match="null pointer"):
        N.call("lyc_locon_linear_fwd", None, None, None, None, None, 4, 8, 8, 2, 1.0, N.LYC_BF16, None)
    with pytest.raises(RuntimeError, match="bad dims"):
        N.call("lyc_chan_scale", None, None, None, None, 4, 0, 1, 1.0, 1.0, N.LYC_F32, None)
    assert N.load().lyc_loha_workspace_bytes(1280, 1280, N.LYC_BF16) == 4 * 1280 * 1280 * 2
